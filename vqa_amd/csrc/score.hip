// Token scoring: head-averaged single-query attention weights without projecting the keys.
//
// For one query per sample the key projection folds into the query:
//   (Wq_h q + bq_h) . (Wk_h x_n + bk_h) / sqrt(dh) = x_n . u_h + const_h,
//   u_h = Wk_h^T (Wq_h q + bq_h) / sqrt(dh),
// and const_h cancels in the softmax over n.  So scoring is one streaming pass over the tokens
// with H dot products of length D per token: bandwidth-bound (16 flop per bf16 element).
//
// head_scores_kernel: a wave owns a token at a time.  The 64 lanes each hold 8 consecutive
// columns of the row (one 16-byte load for bf16, two for f32: fully coalesced 1-2 KiB per
// wave-instruction) and the matching 8 x H slice of u in registers.  The H partial sums per
// lane are combined with a halving exchange (4+2+1 shuffles for H = 8) followed by a 3-step
// butterfly, instead of H full 6-step reductions.
#include <stdlib.h>

#include <atomic>
#include <mutex>

#include "common.h"

namespace {

constexpr int SC_THREADS = 256;
constexpr int SC_WAVES = SC_THREADS / VQAF_WAVE;
constexpr int SC_UNROLL = 4;

template <int HP>
struct Log2 { static constexpr int v = 1 + Log2<HP / 2>::v; };
template <>
struct Log2<1> { static constexpr int v = 0; };

// Reduces part[0..HP) over the 64 lanes.  On return every lane holds in part[0] the total of
// head (lane >> (6 - log2 HP)).
template <int HP>
__device__ __forceinline__ void heads_reduce(float (&part)[HP], int lane) {
  constexpr int LG = Log2<HP>::v;
  int cnt = HP;
#pragma unroll
  for (int s = 0; s < LG; ++s) {
    const int off = 32 >> s;
    const bool hi = (lane & off) != 0;
    const int half = cnt >> 1;
#pragma unroll
    for (int i = 0; i < HP / 2; ++i) {
      if (i < half) {
        const float keep = hi ? part[i + half] : part[i];
        const float send = hi ? part[i] : part[i + half];
        part[i] = keep + __shfl_xor(send, off, 64);
      }
    }
    cnt = half;
  }
#pragma unroll
  for (int off = (32 >> LG); off > 0; off >>= 1) part[0] += __shfl_xor(part[0], off, 64);
}

// CH = number of 512-column chunks (D <= 512*CH).  Lane l covers columns [512c + 8l, +8).
template <bool BF16, int HP, int CH>
__global__ void __launch_bounds__(SC_THREADS)
head_scores_kernel(const void* __restrict__ seq, const float* __restrict__ u, int N, int D, int H,
                   int tokb, float* __restrict__ s) {
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * tokb;
  const int t1 = min(N, t0 + tokb);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;

  float ur[CH][HP][8];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int col = 512 * c + 8 * lane;
#pragma unroll
    for (int h = 0; h < HP; ++h) {
      if (h < H && col < D) {
        load8<false>(u, ((size_t)b * H + h) * D + col, ur[c][h]);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) ur[c][h][i] = 0.f;
      }
    }
  }
  const size_t seq_b = (size_t)b * N * D;
  constexpr int LG = Log2<HP>::v;
  for (int t = t0 + wave; t < t1; t += SC_WAVES * SC_UNROLL) {
    float x[SC_UNROLL][CH][8];
#pragma unroll
    for (int uu = 0; uu < SC_UNROLL; ++uu) {
      const int tt = t + uu * SC_WAVES;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int col = 512 * c + 8 * lane;
        if (tt < t1 && col < D) {
          load8<BF16>(seq, seq_b + (size_t)tt * D + col, x[uu][c]);
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) x[uu][c][i] = 0.f;
        }
      }
    }
#pragma unroll
    for (int uu = 0; uu < SC_UNROLL; ++uu) {
      const int tt = t + uu * SC_WAVES;
      float part[HP];
#pragma unroll
      for (int h = 0; h < HP; ++h) {
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
          for (int i = 0; i < 8; ++i) a = fmaf(x[uu][c][i], ur[c][h][i], a);
        part[h] = a;
      }
      heads_reduce<HP>(part, lane);
      const int h = lane >> (6 - LG);
      if (tt < t1 && (lane & ((64 >> LG) - 1)) == 0 && h < H)
        s[((size_t)b * N + tt) * H + h] = part[0];
    }
  }
}

// bf16 tokens on the matrix cores.  One wave scores 16 tokens per tile with
// v_mfma_f32_16x16x32_bf16: A = the token tile (16 tokens x 32 columns per step, each lane one
// 16-byte load straight from HBM: rows 64 B apart in k, so two consecutive steps use one 128-B
// line), B = [bf16(u) | remainder] (32 columns x 16), so output column h holds x.u_hi and column
// h+8 holds x.u_lo; their sum is the f32-accurate score.  No cross-lane reduction is needed.
// The B operand is built once per workgroup in LDS straight from u (f32): row h = bf16(u_h), row
// h + 8 = bf16(u_h - bf16(u_h)) (rows padded by 16 B: conflict-free ds_read_b128), shared by the
// four waves; token tiles are prefetched one tile ahead, the first one before the LDS staging.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
// Up to four token sets of different length scored by one launch (blockIdx.z picks the set).
constexpr int SC_MAX_MODS = 4;
struct ScoreMods {
  const bf16_t* seq[SC_MAX_MODS];
  const float* u[SC_MAX_MODS];
  float* s[SC_MAX_MODS];
  int N[SC_MAX_MODS];
};
template <int STEPS>
__global__ void __launch_bounds__(SC_THREADS)
head_scores_mfma_kernel(ScoreMods mods, int H, int tokb) {
  constexpr int D = 32 * STEPS;
  constexpr int PITCH = D + 8;
  __shared__ __attribute__((aligned(16))) bf16_t ub[16 * PITCH];
  const bf16_t* __restrict__ seq = mods.seq[blockIdx.z];
  const float* __restrict__ u = mods.u[blockIdx.z];
  float* __restrict__ s = mods.s[blockIdx.z];
  const int N = mods.N[blockIdx.z];
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * tokb;
  if (t0 >= N) return;  // a shorter set needs fewer token blocks
  const int t1 = min(N, t0 + tokb);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int col = lane & 15, kg = lane >> 4;
  const bf16_t* seq_b = seq + (size_t)b * N * D + 8 * kg;

  bf16x8_t cur[STEPS], nxt[STEPS];
  auto load_tile = [&](bf16x8_t (&dst)[STEPS], int tile) {
    const int row = min(tile + col, N - 1);  // rows past the end re-read the last token; never stored
    const bf16_t* xr = seq_b + (size_t)row * D;
#pragma unroll
    for (int st = 0; st < STEPS; ++st) dst[st] = *reinterpret_cast<const bf16x8_t*>(xr + 32 * st);
  };
  int tile = t0 + 16 * wave;
  load_tile(cur, tile);
  {
    const float* ug = u + (size_t)b * H * D;
    for (int c = threadIdx.x; c < 8 * D / 4; c += SC_THREADS) {
      const int h = c / (D / 4), k4 = c % (D / 4);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (h < H) v = *reinterpret_cast<const f32x4*>(ug + (size_t)h * D + 4 * k4);
      const bf16_t h0 = f32_to_bf16(v.x), h1 = f32_to_bf16(v.y), h2 = f32_to_bf16(v.z), h3 = f32_to_bf16(v.w);
      const u32x2 hi = {(unsigned int)h0 | ((unsigned int)h1 << 16), (unsigned int)h2 | ((unsigned int)h3 << 16)};
      const u32x2 lo = {pack_bf16x2(v.x - bf16_to_f32(h0), v.y - bf16_to_f32(h1)),
                        pack_bf16x2(v.z - bf16_to_f32(h2), v.w - bf16_to_f32(h3))};
      *reinterpret_cast<u32x2*>(&ub[h * PITCH + 4 * k4]) = hi;
      *reinterpret_cast<u32x2*>(&ub[(h + 8) * PITCH + 4 * k4]) = lo;
    }
  }
  __syncthreads();
  const bf16_t* ubl = &ub[col * PITCH + 8 * kg];
  for (; tile < t1; tile += 16 * SC_WAVES) {
    const bool more = tile + 16 * SC_WAVES < t1;
    load_tile(nxt, tile + 16 * SC_WAVES);  // unconditional (row clamped): keeps the waits counted
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
      const bf16x8_t bf = *reinterpret_cast<const bf16x8_t*>(ubl + 32 * st);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[st], bf, acc, 0, 0, 0);
    }
    // C/D layout: column = lane & 15, row = 4 * (lane >> 4) + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = acc[r] + __shfl_xor(acc[r], 8, 64);
      const int tok = tile + 4 * kg + r;
      if (col < H && tok < t1) s[((size_t)b * N + tok) * H + col] = v;
    }
    if (more) {
#pragma unroll
      for (int st = 0; st < STEPS; ++st) cur[st] = nxt[st];
    }
  }
}

// f32 tokens on the f32 matrix cores (v_mfma_f32_16x16x4_f32: exact f32 products, f32 sums).  Same
// tiling as the bf16 kernel: a wave scores 16 tokens per tile, lane (token = lane & 15, kg = lane >> 4)
// loads 16 bytes = columns 16 j + 4 kg .. + 3 of its token straight from HBM and feeds them to four
// MFMAs, one column each; the B lane (head = lane & 15, kg) reads the same four columns of u from
// LDS, so both operands agree on which k a step multiplies and no transposition is needed.  The tile
// is loaded in sets of eight 16-byte loads that alternate between two register sets: the MFMAs of
// one set run while the loads of the next (of this or of the wave's next tile) are in flight, at a
// register count that lets four waves share a SIMD.  Heads beyond H multiply zeros.
template <int STEPS>  // D = 16 * STEPS
__global__ void __launch_bounds__(SC_THREADS)
head_scores_mfma_f32_kernel(ScoreMods mods, int H, int tokb) {
  constexpr int D = 16 * STEPS;
  constexpr int HS = STEPS >= 16 ? 8 : STEPS / 2;  // 16-byte loads per register set
  constexpr int Q = STEPS / HS;                    // sets per tile (even)
  constexpr int PITCH = D + 4;
  __shared__ __attribute__((aligned(16))) float ub[8 * PITCH];
  const float* __restrict__ seq = reinterpret_cast<const float*>(mods.seq[blockIdx.z]);
  const float* __restrict__ u = mods.u[blockIdx.z];
  float* __restrict__ s = mods.s[blockIdx.z];
  const int N = mods.N[blockIdx.z];
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * tokb;
  if (t0 >= N) return;
  const int t1 = min(N, t0 + tokb);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int col = lane & 15, kg = lane >> 4;
  const float* seq_b = seq + (size_t)b * N * D + 4 * kg;

  f32x4 ha[HS], hb[HS];
  auto load_set = [&](f32x4 (&dst)[HS], int tile, int q) {
    const int row = min(tile + col, N - 1);  // rows past the end re-read the last token; never stored
    const float* xr = seq_b + (size_t)row * D + 16 * HS * q;
#pragma unroll
    for (int st = 0; st < HS; ++st) dst[st] = *reinterpret_cast<const f32x4*>(xr + 16 * st);
  };
  int tile = t0 + 16 * wave;
  load_set(ha, tile, 0);
  load_set(hb, tile, 1);
  {
    const float* ug = u + (size_t)b * H * D;
    for (int c = threadIdx.x; c < H * (D / 4); c += SC_THREADS) {
      const int h = c / (D / 4), k4 = c % (D / 4);
      *reinterpret_cast<f32x4*>(&ub[h * PITCH + 4 * k4]) = *reinterpret_cast<const f32x4*>(ug + (size_t)h * D + 4 * k4);
    }
  }
  __syncthreads();
  const bool head = col < H;
  const float* ubl = &ub[(head ? col : 0) * PITCH + 4 * kg];
  f32x4 acc;
  auto multiply = [&](const f32x4 (&x)[HS], int q) {
#pragma unroll
    for (int st = 0; st < HS; ++st) {
      f32x4 bf = *reinterpret_cast<const f32x4*>(ubl + 16 * (HS * q + st));
      if (!head) bf = f32x4{0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x[st].x, bf.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x[st].y, bf.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x[st].z, bf.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x[st].w, bf.w, acc, 0, 0, 0);
    }
  };
  for (; tile < t1; tile += 16 * SC_WAVES) {
    const int next = tile + 16 * SC_WAVES;
    acc = f32x4{0.f, 0.f, 0.f, 0.f};
    // u is read from LDS again for every tile: hoisted out of this loop it would take 8 * STEPS
    // registers and halve the waves per SIMD
    asm volatile("" ::: "memory");
#pragma unroll
    for (int q = 0; q < Q; q += 2) {
      // every load is unconditional (rows clamped), so the waits stay counted
      multiply(ha, q);
      if (q + 2 < Q) load_set(ha, tile, q + 2); else load_set(ha, next, 0);
      multiply(hb, q + 1);
      if (q + 3 < Q) load_set(hb, tile, q + 3); else load_set(hb, next, 1);
    }
    // C/D layout: column = lane & 15, row = 4 * (lane >> 4) + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int tok = tile + 4 * kg + r;
      if (head && tok < t1) s[((size_t)b * N + tok) * H + col] = acc[r];
    }
  }
}

// w[b,n] = (1/H) sum_h exp(s[b,n,h] - max_h) / sum_n exp(s[b,n,h] - max_h).
// One workgroup per sample; s of one sample (N*H*4 bytes) is L2-resident, so the three
// passes re-read it from cache.
template <int HP>
__global__ void __launch_bounds__(SC_THREADS)
attn_weights_kernel(const float* __restrict__ s, int N, int H, float* __restrict__ w) {
  __shared__ float red[SC_WAVES][HP];
  __shared__ float stat[2][HP];
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float* sb = s + (size_t)b * N * H;
  float m[HP];
#pragma unroll
  for (int h = 0; h < HP; ++h) m[h] = -INFINITY;
  for (int n = tid; n < N; n += SC_THREADS)
#pragma unroll
    for (int h = 0; h < HP; ++h)
      if (h < H) m[h] = fmaxf(m[h], sb[(size_t)n * H + h]);
#pragma unroll
  for (int h = 0; h < HP; ++h) {
    const float v = wave_max(m[h]);
    if (lane == 0) red[wave][h] = v;
  }
  __syncthreads();
  if (tid < HP) {
    float v = red[0][tid];
    for (int i = 1; i < SC_WAVES; ++i) v = fmaxf(v, red[i][tid]);
    stat[0][tid] = v;
  }
  __syncthreads();
#pragma unroll
  for (int h = 0; h < HP; ++h) m[h] = stat[0][h];
  float z[HP];
#pragma unroll
  for (int h = 0; h < HP; ++h) z[h] = 0.f;
  for (int n = tid; n < N; n += SC_THREADS)
#pragma unroll
    for (int h = 0; h < HP; ++h)
      if (h < H) z[h] += expf(sb[(size_t)n * H + h] - m[h]);
#pragma unroll
  for (int h = 0; h < HP; ++h) {
    const float v = wave_sum(z[h]);
    if (lane == 0) red[wave][h] = v;
  }
  __syncthreads();
  if (tid < HP) {
    float v = red[0][tid];
    for (int i = 1; i < SC_WAVES; ++i) v += red[i][tid];
    stat[1][tid] = v;
  }
  __syncthreads();
#pragma unroll
  for (int h = 0; h < HP; ++h) z[h] = stat[1][h];
  for (int n = tid; n < N; n += SC_THREADS) {
    float a = 0.f;
#pragma unroll
    for (int h = 0; h < HP; ++h)
      if (h < H) a += expf(sb[(size_t)n * H + h] - m[h]) / z[h];
    w[(size_t)b * N + n] = a / (float)H;
  }
}

inline int pad_heads(int H) {
  int p = 1;
  while (p < H) p <<= 1;
  return p;
}
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" int vqaf_fold_query(const void* q, const void* in_proj_weight, const void* in_proj_bias,
                               int B, int D, int H, int dtype, int w_dtype, float* qproj_ws,
                               float* u, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(q && in_proj_weight && in_proj_bias && qproj_ws && u, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_dtype_ok(dtype) && vqaf_dtype_ok(w_dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(B > 0 && D > 0 && H > 0 && D % H == 0, VQAF_E_SHAPE);
  // qproj (B,D) = q Wq^T + bq;  u[b,h,:] = (1/sqrt(dh)) * qproj[b, h*dh:(h+1)*dh] @ Wk[h*dh:(h+1)*dh, :]
  const vqaf_fold_desc d = {q, in_proj_weight, in_proj_bias, qproj_ws, u};
  return vqaf_fold_query_multi(&d, 1, B, D, H, dtype, w_dtype, stream);
}

static int launch_scores_mfma(const ScoreMods& mods, int n, int B, int Nmax, int D, int H, hipStream_t st);

static bool mfma_steps_ok(int D) {
  return D == 64 || D == 128 || D == 256 || D == 512 || D == 768 || D == 1024;
}
static bool mfma_f32_ok(int D, int H) {  // two half tiles of D/32 16-byte registers per lane
  return H <= 8 && (D == 64 || D == 128 || D == 256 || D == 512);
}
static int launch_scores_mfma_f32(const ScoreMods& mods, int n, int B, int Nmax, int D, int H, hipStream_t st);

extern "C" int vqaf_head_scores(const void* seq, const float* u, int B, int N, int D, int H,
                                int dtype, float* s, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(seq && u && s, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_dtype_ok(dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(B > 0 && N > 0 && D > 0 && H > 0 && B <= 65535, VQAF_E_SHAPE);
  VQAF_REQUIRE(D % 8 == 0 && vqaf_aligned16(seq) && vqaf_aligned16(u), VQAF_E_ALIGN);
  if (dtype == VQAF_BF16 && H <= 8 && mfma_steps_ok(D)) {
    ScoreMods mods = {};
    mods.seq[0] = reinterpret_cast<const bf16_t*>(seq);
    mods.u[0] = u;
    mods.s[0] = s;
    mods.N[0] = N;
    return launch_scores_mfma(mods, 1, B, N, D, H, (hipStream_t)stream);
  }
  if (dtype == VQAF_F32 && mfma_f32_ok(D, H)) {
    ScoreMods mods = {};
    mods.seq[0] = reinterpret_cast<const bf16_t*>(seq);  // the f32 kernel reads it as float
    mods.u[0] = u;
    mods.s[0] = s;
    mods.N[0] = N;
    return launch_scores_mfma_f32(mods, 1, B, N, D, H, (hipStream_t)stream);
  }
  const int HP = pad_heads(H);
  const int CH = (D + 511) / 512;
  VQAF_REQUIRE(HP * CH <= 16 && CH <= 2, VQAF_E_SHAPE);  // register budget of the u slice
  int tokb = 64;
  if ((long)B * ((N + 63) / 64) < 512) tokb = 32;
  if ((long)B * ((N + 31) / 32) < 512) tokb = 16;
  const dim3 grid((N + tokb - 1) / tokb, B), block(SC_THREADS);
  hipStream_t st = (hipStream_t)stream;
  const bool bf = dtype == VQAF_BF16;
#define VQAF_HS(HPv, CHv)                                                                       \
  do {                                                                                          \
    if (bf) hipLaunchKernelGGL((head_scores_kernel<true, HPv, CHv>), grid, block, 0, st, seq, u, \
                               N, D, H, tokb, s);                                               \
    else hipLaunchKernelGGL((head_scores_kernel<false, HPv, CHv>), grid, block, 0, st, seq, u,   \
                            N, D, H, tokb, s);                                                  \
  } while (0)
  if (CH == 1) {
    switch (HP) {
      case 1: VQAF_HS(1, 1); break;
      case 2: VQAF_HS(2, 1); break;
      case 4: VQAF_HS(4, 1); break;
      case 8: VQAF_HS(8, 1); break;
      default: VQAF_HS(16, 1); break;
    }
  } else {
    switch (HP) {
      case 1: VQAF_HS(1, 2); break;
      case 2: VQAF_HS(2, 2); break;
      case 4: VQAF_HS(4, 2); break;
      default: VQAF_HS(8, 2); break;
    }
  }
#undef VQAF_HS
  VQAF_LAUNCH_CHECK();
}

static int launch_scores_mfma(const ScoreMods& mods, int n, int B, int Nmax, int D, int H, hipStream_t st) {
  static const int forced_tokb = [] {
    const char* e = getenv("VQAF_HS_TOKB");  // tuning knob: tokens per workgroup (multiple of 64)
    return e ? atoi(e) : 0;
  }();
  int tokb = 64;
  if ((long)B * ((Nmax + 127) / 128) >= 512) tokb = 128;
  if ((long)B * ((Nmax + 255) / 256) >= 1024) tokb = 256;
  // Once blocks of up to 320 tokens still give every CU a workgroup, a sample is cut into EQUAL blocks of that
  // size class: a wave then streams five tiles behind one staging of u, and no block is a sliver (N = 577 at
  // batch 128: 2 x 320 takes 18.3 us where 5 x 128 -- the last one 65 tokens -- takes 23.9 us).
  const int blocks320 = (Nmax + 319) / 320;
  if ((long)B * blocks320 >= 256) tokb = ((Nmax + blocks320 - 1) / blocks320 + 63) / 64 * 64;
  if (forced_tokb >= 64 && forced_tokb % 64 == 0) tokb = forced_tokb;
  const dim3 grid((Nmax + tokb - 1) / tokb, B, n), block(SC_THREADS);
#define VQAF_HSM(S) hipLaunchKernelGGL(head_scores_mfma_kernel<S>, grid, block, 0, st, mods, H, tokb)
  switch (D / 32) {
    case 2: VQAF_HSM(2); break;
    case 4: VQAF_HSM(4); break;
    case 8: VQAF_HSM(8); break;
    case 16: VQAF_HSM(16); break;
    case 24: VQAF_HSM(24); break;
    default: VQAF_HSM(32); break;
  }
#undef VQAF_HSM
  VQAF_LAUNCH_CHECK();
}

static int launch_scores_mfma_f32(const ScoreMods& mods, int n, int B, int Nmax, int D, int H, hipStream_t st) {
  int tokb = 64;
  if ((long)B * ((Nmax + 127) / 128) >= 512) tokb = 128;
  if ((long)B * ((Nmax + 255) / 256) >= 1024) tokb = 256;
  const dim3 grid((Nmax + tokb - 1) / tokb, B, n), block(SC_THREADS);
#define VQAF_HSF(S) hipLaunchKernelGGL(head_scores_mfma_f32_kernel<S>, grid, block, 0, st, mods, H, tokb)
  switch (D / 16) {
    case 4: VQAF_HSF(4); break;
    case 8: VQAF_HSF(8); break;
    case 16: VQAF_HSF(16); break;
    default: VQAF_HSF(32); break;
  }
#undef VQAF_HSF
  VQAF_LAUNCH_CHECK();
}

extern "C" int vqaf_head_scores_multi(const vqaf_scores_desc* descs_host, int n, int B, int D, int H,
                                      int dtype, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(descs_host != nullptr, VQAF_E_NULL);
  VQAF_REQUIRE(n > 0 && n <= SC_MAX_MODS && B > 0 && B <= 65535 && D > 0 && H > 0, VQAF_E_SHAPE);
  VQAF_REQUIRE(vqaf_dtype_ok(dtype), VQAF_E_DTYPE);
  const bool f32_mfma = dtype == VQAF_F32 && mfma_f32_ok(D, H);
  if (!(dtype == VQAF_BF16 && H <= 8 && mfma_steps_ok(D)) && !f32_mfma) {
    for (int i = 0; i < n; ++i) {  // other shapes: one launch per set
      const int rc = vqaf_head_scores(descs_host[i].seq, descs_host[i].u, B, descs_host[i].N, D, H, dtype,
                                      descs_host[i].s, stream);
      if (rc != VQAF_OK) return rc;
    }
    return VQAF_OK;
  }
  ScoreMods mods = {};
  int nmax = 0;
  for (int i = 0; i < n; ++i) {
    const vqaf_scores_desc& d = descs_host[i];
    VQAF_REQUIRE(d.seq && d.u && d.s, VQAF_E_NULL);
    VQAF_REQUIRE(d.N > 0, VQAF_E_SHAPE);
    VQAF_REQUIRE(vqaf_aligned16(d.seq) && vqaf_aligned16(d.u), VQAF_E_ALIGN);
    mods.seq[i] = reinterpret_cast<const bf16_t*>(d.seq);
    mods.u[i] = d.u;
    mods.s[i] = d.s;
    mods.N[i] = d.N;
    nmax = d.N > nmax ? d.N : nmax;
  }
  if (f32_mfma) return launch_scores_mfma_f32(mods, n, B, nmax, D, H, (hipStream_t)stream);
  return launch_scores_mfma(mods, n, B, nmax, D, H, (hipStream_t)stream);
}

extern "C" int vqaf_attn_weights(const float* s, int B, int N, int H, float* w,
                                 vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(s && w, VQAF_E_NULL);
  VQAF_REQUIRE(B > 0 && N > 0 && H > 0 && H <= 16, VQAF_E_SHAPE);
  const int HP = pad_heads(H);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(B), block(SC_THREADS);
  switch (HP) {
    case 1: hipLaunchKernelGGL(attn_weights_kernel<1>, grid, block, 0, st, s, N, H, w); break;
    case 2: hipLaunchKernelGGL(attn_weights_kernel<2>, grid, block, 0, st, s, N, H, w); break;
    case 4: hipLaunchKernelGGL(attn_weights_kernel<4>, grid, block, 0, st, s, N, H, w); break;
    case 8: hipLaunchKernelGGL(attn_weights_kernel<8>, grid, block, 0, st, s, N, H, w); break;
    default: hipLaunchKernelGGL(attn_weights_kernel<16>, grid, block, 0, st, s, N, H, w); break;
  }
  VQAF_LAUNCH_CHECK();
}

extern "C" size_t vqaf_score_workspace_bytes(int B, int N, int D, int H) {
  if (B <= 0 || N <= 0 || D <= 0 || H <= 0) return 0;
  return align256((size_t)B * D * 4) + align256((size_t)B * H * D * 4) +
         align256((size_t)B * N * H * 4);
}

extern "C" int vqaf_score(const void* q, const void* seq, const void* in_proj_weight,
                          const void* in_proj_bias, int B, int N, int D, int H, int dtype,
                          int w_dtype, void* workspace, size_t workspace_bytes, float* w,
                          vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(workspace != nullptr && w != nullptr, VQAF_E_NULL);
  VQAF_REQUIRE(B > 0 && N > 0 && D > 0 && H > 0, VQAF_E_SHAPE);
  VQAF_REQUIRE(workspace_bytes >= vqaf_score_workspace_bytes(B, N, D, H), VQAF_E_SHAPE);
  VQAF_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255u) == 0, VQAF_E_ALIGN);
  unsigned char* ws = reinterpret_cast<unsigned char*>(workspace);
  float* qproj = reinterpret_cast<float*>(ws);
  float* u = reinterpret_cast<float*>(ws + align256((size_t)B * D * 4));
  float* s = reinterpret_cast<float*>(ws + align256((size_t)B * D * 4) + align256((size_t)B * H * D * 4));
  int rc = vqaf_fold_query(q, in_proj_weight, in_proj_bias, B, D, H, dtype, w_dtype, qproj, u, stream);
  if (rc != VQAF_OK) return rc;
  rc = vqaf_head_scores(seq, u, B, N, D, H, dtype, s, stream);
  if (rc != VQAF_OK) return rc;
  return vqaf_attn_weights(s, B, N, H, w, stream);
}

// ---------------------------------------------------------------------------------------
// vqaf_score without a caller workspace and without any allocation: the three intermediate
// buffers (qproj, u, s) live in a file-scope device array that every GPU holds from the moment
// the code object is loaded until the process ends.
//
// `score_scratch` names the array in DEVICE code only; in host code the same identifier is a
// host-side stand-in whose address means nothing to a kernel.  The device address is asked for
// once per device (with that device current) and kept; host code below uses only that pointer.
// ---------------------------------------------------------------------------------------
static __device__ __align__(256) unsigned char score_scratch[VQAF_SCORE_SCRATCH_BYTES];

namespace {
constexpr int SCRATCH_MAX_DEVICES = 64;
std::atomic<unsigned char*> g_scratch_ptr[SCRATCH_MAX_DEVICES];
std::mutex g_scratch_mu;

// Device address of score_scratch on the CURRENT device (each GPU has its own copy at its own
// address).  The first call on a device resolves the symbol; later calls are one atomic load.
int scratch_on_current_device(unsigned char** out) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SCRATCH_MAX_DEVICES) return VQAF_E_LAUNCH;
  unsigned char* p = g_scratch_ptr[dev].load(std::memory_order_acquire);
  if (p == nullptr) {
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    p = g_scratch_ptr[dev].load(std::memory_order_relaxed);
    if (p == nullptr) {
      void* addr = nullptr;
      if (hipGetSymbolAddress(&addr, HIP_SYMBOL(score_scratch)) != hipSuccess || addr == nullptr) {
        (void)hipGetLastError();
        return VQAF_E_LAUNCH;
      }
      p = reinterpret_cast<unsigned char*>(addr);
      g_scratch_ptr[dev].store(p, std::memory_order_release);
    }
  }
  *out = p;
  return VQAF_OK;
}
}  // namespace

extern "C" int vqaf_score_noalloc(const void* q, const void* seq, const void* in_proj_weight,
                                  const void* in_proj_bias, int B, int N, int D, int H, int dtype,
                                  int w_dtype, float* w, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(w != nullptr, VQAF_E_NULL);
  VQAF_REQUIRE(B > 0 && N > 0 && D > 0 && H > 0, VQAF_E_SHAPE);
  // sizes are checked on the host before anything touches the device
  VQAF_REQUIRE(vqaf_score_workspace_bytes(B, N, D, H) <= (size_t)VQAF_SCORE_SCRATCH_BYTES, VQAF_E_SCRATCH);
  unsigned char* ws = nullptr;
  const int rc = scratch_on_current_device(&ws);
  if (rc != VQAF_OK) return rc;
  // every byte of qproj, u and s that a kernel reads is written earlier in this same call, so
  // nothing left in the array by the previous call needs clearing
  return vqaf_score(q, seq, in_proj_weight, in_proj_bias, B, N, D, H, dtype, w_dtype, ws,
                    (size_t)VQAF_SCORE_SCRATCH_BYTES, w, stream);
}
