// Pruning branch of one modality in ONE launch: token scoring -> softmax / head mean -> ranking ->
// adversarial index swap -> pooled gather + mix token.  One workgroup owns one sample.
//
// The separate kernels read the sample's tokens from HBM twice (scoring pass, pooling pass) with
// the selection kernel between them, by which time the tokens have left every cache.  Here the two
// passes of a sample are a few microseconds apart inside one workgroup, so the second pass is
// served on-die (Infinity Cache; the XCD's L2 for short token sets) and HBM sees each token once.
//
//   pass 1   s[n,h] = x_n . u_h on bf16 MFMA 16x16x32 (tokens straight from HBM as the A operand,
//            [bf16(u) | bf16(u - bf16(u))] as the B operand from LDS), scores stay in LDS;
//   select   select_core (select.h): the same code the selection kernels run -> keep list in LDS,
//            written out as int64 indices (the backward and the callers need them);
//   pass 2   rows again in memory order, a lane owns 8 columns of a row, a wave-sized lane group a
//            whole row: kept rows (times their multiplicity) into the pooled sum, dropped rows
//            into the mix-token sum; noise is added per keep slot; fixed summation order.
#include <stdlib.h>

#include "select.h"

namespace {

constexpr int FU_THREADS = 256;
constexpr int FU_WAVES = FU_THREADS / VQAF_WAVE;
constexpr int FU_MAX_MODS = 4;
constexpr int FU_RED_BYTES = 2 * FU_THREADS * 8 * 4;  // two sums x 256 lanes x 8 columns, f32

struct FusedMod {
  const bf16_t* seq;          // (B,N,D)
  const float* u;             // (B,H,D) folded queries
  const int64_t* pos;         // (B,k_adv) injected keep-list positions (adv_mode 1)
  const int64_t* src;         // (B,k_adv) injected bottom-list positions
  const float* noise;         // (B,K,D) injected noise or NULL
  const uint64_t* noise_rng;  // {seed, counter} of the in-kernel noise
  int64_t* idx_out;           // (B,K)
  float* w_out;               // (B,N) or NULL
  unsigned long long salt;
  float alpha, noise_scale;
  int noise_kind, N, K, NP, adv_mode, k_adv, NPK, NPR, col0, noise_ctr_off;
};
struct FusedMods {
  FusedMod m[FU_MAX_MODS];
};

__host__ __device__ inline size_t fu_align16(size_t x) { return (x + 15) & ~(size_t)15; }

// LDS carve, identical on host (size) and device (pointers).
struct FusedCarve {
  size_t ub, keys, sl, rank, s_pos, wl, mult, stat, misc, total;
};
__host__ __device__ inline FusedCarve fused_carve(int D, int H, int N, int NP, int k_adv) {
  FusedCarve c;
  size_t o = 0;
  c.ub = o;
  const size_t ub_bytes = (size_t)16 * (D + 8) * 2;
  o += fu_align16(ub_bytes > (size_t)FU_RED_BYTES ? ub_bytes : (size_t)FU_RED_BYTES);
  c.keys = o;  o += fu_align16((size_t)NP * 8);
  c.sl = o;    o += fu_align16((size_t)N * H * 4);
  c.rank = o;  o += fu_align16((size_t)N * 4);
  c.s_pos = o; o += fu_align16((size_t)(k_adv > 0 ? 2 * k_adv : 1) * 4);
  c.wl = o;    o += fu_align16((size_t)N * 4);
  c.mult = o;  o += fu_align16((size_t)N * 4);
  c.stat = o;  o += fu_align16((size_t)2 * SEL_MAX_H * 4);
  c.misc = o;  o += 16;
  c.total = o;
  return c;
}

typedef __bf16 fu_bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void unpack_bf16x8(const u32x4 r, float (&v)[8]) {
  v[0] = bf16_lo(r.x); v[1] = bf16_hi(r.x); v[2] = bf16_lo(r.y); v[3] = bf16_hi(r.y);
  v[4] = bf16_lo(r.z); v[5] = bf16_hi(r.z); v[6] = bf16_lo(r.w); v[7] = bf16_hi(r.w);
}

template <int STEPS>
__global__ void __launch_bounds__(FU_THREADS, (STEPS <= 16 ? 4 : 2))
fused_prune_pool_kernel(FusedMods mods, int n_mod, int H, uint64_t* sel_rng,
                        void* __restrict__ out, int ld_out, int out_bf16, int rng_advance) {
  // Modalities along y: every long (vision) sample is dispatched before the short (text) ones,
  // which then fill the tail of the launch.  Measured alternatives, both slower: alternating the
  // modalities in dispatch order (+22 us at B = 1024: the last long samples start late), and one
  // workgroup taking both modalities of its sample in turn (register spills).
  const int b = blockIdx.x;
  const FusedMod& M = mods.m[blockIdx.y];
  constexpr int D = 32 * STEPS;
  constexpr int PITCH = D + 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char fu_lds[];
  const int N = M.N, K = M.K, NP = M.NP, k_adv = M.k_adv;
  const FusedCarve cv = fused_carve(D, H, N, NP, k_adv);
  bf16_t* ub = reinterpret_cast<bf16_t*>(fu_lds + cv.ub);
  float* red = reinterpret_cast<float*>(fu_lds + cv.ub);  // re-used after pass 1
  SelectLds L;
  L.keys = reinterpret_cast<unsigned long long*>(fu_lds + cv.keys);
  L.sl = reinterpret_cast<float*>(fu_lds + cv.sl);
  L.rank = reinterpret_cast<int*>(fu_lds + cv.rank);
  L.s_pos = reinterpret_cast<int*>(fu_lds + cv.s_pos);
  L.wl = reinterpret_cast<float*>(fu_lds + cv.wl);
  L.stat = reinterpret_cast<float*>(fu_lds + cv.stat);
  int* mult = reinterpret_cast<int*>(fu_lds + cv.mult);
  int* misc = reinterpret_cast<int*>(fu_lds + cv.misc);

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const bf16_t* __restrict__ seq_b = M.seq + (size_t)b * N * D;

  // ------------------------------- pass 1: head scores -----------------------------------
  // A wave scores 16 tokens per tile.  The tile's D columns are loaded in two halves that live in
  // two register sets: while the MFMAs of one half run, the loads of the other half (of this or
  // of the wave's next tile) are in flight, at half the registers of whole-tile double buffering
  // (which is what lets four workgroups share a CU).
  {
    constexpr int HS = (STEPS + 1) / 2;  // steps per half (the second half may be shorter)
    const int col = lane & 15, kg = lane >> 4;
    const bf16_t* xk = seq_b + 8 * kg;
    fu_bf16x8 ha[HS], hb[HS];
    auto load_half = [&](fu_bf16x8 (&dst)[HS], int tile, int first) {
      const int row = min(tile + col, N - 1);  // rows past the end re-read the last token; never stored
      const bf16_t* xr = xk + (size_t)row * D;
#pragma unroll
      for (int st = 0; st < HS; ++st)
        if (first + st < STEPS) dst[st] = *reinterpret_cast<const fu_bf16x8*>(xr + 32 * (first + st));
    };
    int tile = 16 * wave;
    load_half(ha, tile, 0);  // in flight while the operand below is staged
    load_half(hb, tile, HS);
    const float* ug = M.u + (size_t)b * H * D;
    for (int c = tid; c < 8 * D / 4; c += FU_THREADS) {
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
    if (tid < 4) misc[tid] = 0;
    __syncthreads();
    const bf16_t* ubl = &ub[col * PITCH + 8 * kg];
    for (; tile < N; tile += 16 * FU_WAVES) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int st = 0; st < HS; ++st) {
        const fu_bf16x8 bf = *reinterpret_cast<const fu_bf16x8*>(ubl + 32 * st);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha[st], bf, acc, 0, 0, 0);
      }
      load_half(ha, tile + 16 * FU_WAVES, 0);  // unconditional (row clamped): the waits stay counted
#pragma unroll
      for (int st = 0; st < HS; ++st) {
        if (HS + st < STEPS) {
          const fu_bf16x8 bf = *reinterpret_cast<const fu_bf16x8*>(ubl + 32 * (HS + st));
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hb[st], bf, acc, 0, 0, 0);
        }
      }
      load_half(hb, tile + 16 * FU_WAVES, HS);
      // C/D layout: column = lane & 15, row = 4 * (lane >> 4) + r; column h + 8 holds the low part
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = acc[r] + __shfl_xor(acc[r], 8, 64);
        const int tok = tile + 4 * kg + r;
        if (col < H && tok < N) L.sl[tok * H + col] = v;
      }
    }
  }
  __syncthreads();

  // ------------------------------- selection ----------------------------------------------
  unsigned long long seed = 0, counter = 0;
  if (M.adv_mode == 2) {
    seed = sel_rng[0] + M.salt;
    counter = sel_rng[1];
  }
  select_core(L, H, N, K, NP, M.adv_mode, k_adv, M.NPK, M.NPR,
              M.pos ? M.pos + (size_t)b * k_adv : nullptr, M.src ? M.src + (size_t)b * k_adv : nullptr,
              seed, counter, (unsigned long long)b, M.w_out ? M.w_out + (size_t)b * N : nullptr,
              nullptr, nullptr, nullptr);
  for (int n = tid; n < N; n += FU_THREADS) mult[n] = 0;
  __syncthreads();
  for (int j = tid; j < K; j += FU_THREADS) {
    atomicAdd(&mult[L.rank[j]], 1);  // (the keep list goes out after pass 2: a barrier waits for stores)
  }
  __syncthreads();
  {
    int local = 0;
    for (int n = tid; n < N; n += FU_THREADS) local += (mult[n] == 0) ? 1 : 0;
    if (local) atomicAdd(&misc[0], local);
  }
  __syncthreads();
  const int R = misc[0];

  // ------------------------------- pass 2: pooled gather + mix token ----------------------
  constexpr int LPR = 4 * STEPS;           // lanes per row (8 columns each)
  constexpr int ROWS = FU_THREADS / LPR;   // rows per step
  constexpr int U = 8;                     // rows in flight per lane
  const int g = tid / LPR, li = tid % LPR, d0 = li * 8;
  const bool with_mix = M.alpha > 0.f;
  const int Kout = K + (with_mix ? 1 : 0);
  float acck[8], accd[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acck[i] = accd[i] = 0.f;
  {
    const bf16_t* row0 = seq_b + d0;
    u32x4 cur[U], nxt[U];
    auto load_batch = [&](u32x4 (&dst)[U], int n0) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int n = min(n0 + u * ROWS, N - 1);
        dst[u] = *reinterpret_cast<const u32x4*>(row0 + (size_t)n * D);
      }
    };
    load_batch(cur, g);
    for (int n0 = g; n0 < N; n0 += U * ROWS) {
      const bool more = n0 + U * ROWS < N;
      load_batch(nxt, n0 + U * ROWS);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int n = n0 + u * ROWS;
        if (n < N) {
          const int m = mult[n];
          const float mk = (float)m, md = (m == 0) ? 1.f : 0.f;
          float v[8];
          unpack_bf16x8(cur[u], v);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            acck[i] = fmaf(mk, v[i], acck[i]);
            accd[i] = fmaf(md, v[i], accd[i]);
          }
        }
      }
      if (more) {
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
      }
    }
  }
  for (int j = tid; j < K; j += FU_THREADS) M.idx_out[(size_t)b * K + j] = (int64_t)L.rank[j];
  // noise: one draw per keep slot and element, addressed exactly as the unfused kernels address it
  if (M.noise != nullptr) {
    for (int j = g; j < K; j += ROWS) {
      const f32x4* p = reinterpret_cast<const f32x4*>(M.noise + ((size_t)b * K + j) * D + d0);
      const f32x4 a = p[0], c = p[1];
      acck[0] += a.x; acck[1] += a.y; acck[2] += a.z; acck[3] += a.w;
      acck[4] += c.x; acck[5] += c.y; acck[6] += c.z; acck[7] += c.w;
    }
  } else if (M.noise_kind != VQAF_NOISE_NONE) {
    const unsigned long long nseed = M.noise_rng[0];
    const unsigned long long ncounter = M.noise_rng[1] + (unsigned long long)M.noise_ctr_off;
    for (int j = g; j < K; j += ROWS)
      vqaf_token_noise8(acck, nullptr, M.noise_kind, M.noise_scale, nseed, ncounter, ((size_t)b * K + j) * D + d0);
  }
  // fixed-order fold of the ROWS lane groups; `red` overlays the pass-1 operand (all waves are
  // past pass 1: several barriers ago)
  float* redk = red;
  float* redd = red + FU_THREADS * 8;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    redk[g * D + d0 + i] = acck[i];
    redd[g * D + d0 + i] = accd[i];
  }
  __syncthreads();
  const float cnt = (float)R + 1e-10f;
  for (int c = tid; c < D; c += FU_THREADS) {
    float sk = 0.f, sd = 0.f;
    for (int gg = 0; gg < ROWS; ++gg) {
      sk += redk[gg * D + c];
      sd += redd[gg * D + c];
    }
    float tot = sk;
    if (with_mix) tot += M.alpha * (sd / cnt);  // remaining_sum / (count + 1e-10), then * alpha
    const float r = tot / (float)Kout;
    const size_t o = (size_t)b * ld_out + M.col0 + c;
    if (out_bf16) reinterpret_cast<bf16_t*>(out)[o] = f32_to_bf16(r);
    else reinterpret_cast<float*>(out)[o] = r;
  }
  // every draw of this workgroup (selection above, noise in pass 2) has read the state by now
  if (rng_advance != 0)
    vqaf_rng_bump_by_last_workgroup(sel_rng, (unsigned long long)rng_advance, gridDim.x * gridDim.y);
}

}  // namespace

extern "C" int vqaf_fused_prune_pool_multi(const vqaf_fused_mod* mods_host, int n_mod, int B, int D, int H,
                                           int dtype, uint64_t* sel_rng_state, int rng_advance, void* out,
                                           int ld_out, int out_dtype, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(mods_host != nullptr && out != nullptr, VQAF_E_NULL);
  VQAF_REQUIRE(rng_advance >= 0 && (rng_advance == 0 || sel_rng_state != nullptr), VQAF_E_NULL);
  VQAF_REQUIRE(dtype == VQAF_BF16 && vqaf_dtype_ok(out_dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(n_mod > 0 && n_mod <= FU_MAX_MODS && B > 0 && H > 0 && H <= 8, VQAF_E_SHAPE);
  VQAF_REQUIRE(D == 64 || D == 128 || D == 256 || D == 512 || D == 1024, VQAF_E_SHAPE);
  FusedMods mods = {};
  size_t smem = 0;
  for (int i = 0; i < n_mod; ++i) {
    const vqaf_fused_mod& h = mods_host[i];
    VQAF_REQUIRE(h.seq && h.u && h.idx_out, VQAF_E_NULL);
    VQAF_REQUIRE(h.N > 0 && h.K > 0 && h.K <= h.N && h.N <= TK_MAX_N, VQAF_E_SHAPE);
    VQAF_REQUIRE(h.col0 >= 0 && ld_out >= h.col0 + D, VQAF_E_SHAPE);
    VQAF_REQUIRE(vqaf_aligned16(h.seq) && vqaf_aligned16(h.u) && (h.noise == nullptr || vqaf_aligned16(h.noise)),
                 VQAF_E_ALIGN);
    VQAF_REQUIRE(h.noise_kind >= VQAF_NOISE_NONE && h.noise_kind <= VQAF_NOISE_UNIFORM, VQAF_E_SHAPE);
    VQAF_REQUIRE(h.noise != nullptr || h.noise_kind == VQAF_NOISE_NONE || h.noise_rng_state != nullptr, VQAF_E_NULL);
    VQAF_REQUIRE(h.adv_mode >= 0 && h.adv_mode <= 2, VQAF_E_SHAPE);
    const int R = h.N - h.K;
    int k_adv = h.k_adv;
    if (h.adv_mode != 0) {
      VQAF_REQUIRE(k_adv > 0 && k_adv <= h.K && k_adv <= R, VQAF_E_SHAPE);
      if (h.adv_mode == 1) VQAF_REQUIRE(h.pos != nullptr && h.src != nullptr, VQAF_E_NULL);
      if (h.adv_mode == 2) VQAF_REQUIRE(sel_rng_state != nullptr, VQAF_E_NULL);
    } else {
      k_adv = 0;
    }
    const int NP = sel_next_pow2(h.N);
    mods.m[i] = FusedMod{reinterpret_cast<const bf16_t*>(h.seq), h.u, h.pos, h.src, h.noise, h.noise_rng_state,
                         h.idx_out, h.w_out, (unsigned long long)h.salt, h.alpha, h.noise_scale, h.noise_kind,
                         h.N, h.K, NP, h.adv_mode, k_adv, sel_next_pow2(h.K), sel_next_pow2(R > 0 ? R : 1), h.col0,
                         h.noise_counter_offset};
    const size_t need = fused_carve(D, H, h.N, NP, k_adv).total;
    smem = need > smem ? need : smem;
  }
  VQAF_REQUIRE(smem <= 64 * 1024, VQAF_E_SHAPE);
  const dim3 grid(B, n_mod), block(FU_THREADS);
  hipStream_t st = (hipStream_t)stream;
  const int obf = out_dtype == VQAF_BF16 ? 1 : 0;
#define VQAF_FU(S) \
  hipLaunchKernelGGL(fused_prune_pool_kernel<S>, grid, block, smem, st, mods, n_mod, H, sel_rng_state, out, ld_out, obf, rng_advance)
  switch (D / 32) {
    case 2: VQAF_FU(2); break;
    case 4: VQAF_FU(4); break;
    case 8: VQAF_FU(8); break;
    case 16: VQAF_FU(16); break;
    default: VQAF_FU(32); break;
  }
#undef VQAF_FU
  VQAF_LAUNCH_CHECK();
}
