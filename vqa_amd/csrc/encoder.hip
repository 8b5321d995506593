// Transformer-encoder blocks for the two encoders that feed the fusion (ViT-B/16-shaped vision
// encoder, BERT-base text encoder; the reference wraps Hugging Face models for both:
// model/vision_encoder.py:83-127, model/text_encoder.py:14-48).
//
//   * multi-head self-attention, forward and backward, bf16 on v_mfma_f32_32x32x16_bf16 with an
//     online softmax (scores never leave registers), head size 64;
//   * LayerNorm with the residual add folded in, forward and backward;
//   * exact (erf) GELU, forward and backward.
// The Linear layers around them are plain GEMMs and stay with the BLAS library.
//
// Attention layout.  MFMA 32x32 accumulators give a lane ONE column and 16 rows, so every product
// is arranged to put the softmax row (a query) on the column index: S^T = K Q^T, O^T = V^T P^T.
// Row maxima and sums are then per-lane scalars plus one exchange between lane l and l+32, the
// rescaling of O^T is a per-lane multiply, and the accumulator of S^T is, register for register,
// the B operand of the second product (its 16 rows are the contraction index; their order inside
// a 16-row step is a fixed permutation that the A operand is read with as well).  A operands
// that need the contraction index down a column of a row-major tile (V^T, K^T, Q^T, dO^T) come
// from LDS through the transposing read ds_read_b64_tr_b16.
#include <type_traits>

#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int AT_DH = 64;        // head size
constexpr int AT_OWN = 128;      // rows (queries, or keys in the dK/dV kernel) owned by a workgroup: 4 waves x 32
constexpr int AT_BLK = 64;       // streamed rows per stage
constexpr int AT_PR = 72;        // LDS pitch (elements) of tiles read by rows: 16-B reads of 16 lanes hit 64 distinct banks
constexpr int AT_PT = 96;        // LDS pitch of tiles read by the transposing read: its four rows fall on disjoint bank groups
constexpr float AT_LOG2E = 1.4426950408889634f;
constexpr float AT_MIN_BIAS = -1.0e30f;  // additive key bias is clamped here so that a fully masked row stays finite

struct AttnArgs {
  const bf16_t *q, *k, *v;   // (B, N, ld_qkv) each, head h in columns [64h, 64h+64)
  const bf16_t *dout;        // (B, N, ld_o)
  bf16_t *out;               // forward output (B, N, ld_o)
  bf16_t *dq, *dk, *dv;      // (B, N, ld_d)
  const float* key_bias;     // (B, N) additive, or null
  float* lse;                // (B, H, N): log2 of the softmax denominator, in the scaled base-2 domain
  float* dvec;               // (B, H, N): sum_m p[n,m] dp[n,m], the row term of the softmax backward
  long ld_qkv, ld_o, ld_d;
  int N, H;
  float scale;               // 1/sqrt(dh)
  // dropout on the attention probabilities: element (b,h,n,m) is kept iff word (m%4) of the Philox call
  // (seed, counter, ((b*H+h)*N + n) * ceil(N/4) + m/4) is >= drop_thr; kept values are scaled by drop_inv
  const uint64_t* drop_state;  // {seed, counter} in device memory, or null
  unsigned int drop_thr;
  float drop_inv;
};

// keep-scale factors of four consecutive keys 4*group .. 4*group+3 of one query row
__device__ __forceinline__ void drop_scale4(unsigned long long seed, unsigned long long ctr, unsigned long long row,
                                            int groups, int group, unsigned int thr, float inv, float (&mk)[4]) {
  const Philox4 r = philox4x32_10(seed, row * (unsigned long long)groups + (unsigned long long)group, ctr);
  mk[0] = r.x >= thr ? inv : 0.f;
  mk[1] = r.y >= thr ? inv : 0.f;
  mk[2] = r.z >= thr ? inv : 0.f;
  mk[3] = r.w >= thr ? inv : 0.f;
}

__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 as_frag(u32x4 v) {
  union { u32x4 u; bf16x8 b; } c; c.u = v; return c.b;
}
__device__ __forceinline__ float xchg32(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  // one of the two results is the own value, the other the partner's
  const unsigned int own = __float_as_uint(v);
  return __uint_as_float(r[0] == own ? r[1] : r[0]);
}
// A operand whose contraction index runs down the rows of a row-major LDS tile (pitch AT_PT):
// lane (i = lane%32, half = lane/32) receives column col0+i, rows row0 + 4*half + {0..3} and
// row0 + 8 + 4*half + {0..3}: the row order of a 32x32 accumulator within a 16-row step.
__device__ __forceinline__ bf16x8 tr_rows(const bf16_t* tile, int row0, int colbase, int lane) {
  const int i = lane & 15;
  const bf16_t* p0 = tile + (row0 + 4 * (lane >> 5) + (i >> 2)) * AT_PT + colbase + 16 * ((lane >> 4) & 1) + 4 * (i & 3);
  typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)p0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p0 + 8 * AT_PT));
  union { s16x8 s; bf16x8 b; } u;
  u.s = s16x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  return u.b;
}
__device__ __forceinline__ bf16x8 pack8(const f32x16& v, int t) {
  return bf16x8{(__bf16)v[8 * t + 0], (__bf16)v[8 * t + 1], (__bf16)v[8 * t + 2], (__bf16)v[8 * t + 3],
                (__bf16)v[8 * t + 4], (__bf16)v[8 * t + 5], (__bf16)v[8 * t + 6], (__bf16)v[8 * t + 7]};
}
// accumulator register r of a lane in half hh belongs to row 8*(r/4) + 4*hh + r%4
__device__ __forceinline__ int acc_row(int r, int hh) { return 8 * (r >> 2) + 4 * hh + (r & 3); }

// Streams a 64-row block of a (N, ld) matrix (64 columns of one head) through registers.
struct Stage {
  u32x4 a[2];
  __device__ __forceinline__ void issue(const bf16_t* base, long ld, int row0, int N, int tid) {
    const int row = min(row0 + (tid >> 2), N - 1);
    const bf16_t* p = base + (size_t)row * ld + (tid & 3) * 16;
    a[0] = *reinterpret_cast<const u32x4*>(p);
    a[1] = *reinterpret_cast<const u32x4*>(p + 8);
  }
  __device__ __forceinline__ void commit(bf16_t* tile, int pitch, int tid) const {
    bf16_t* p = tile + (tid >> 2) * pitch + (tid & 3) * 16;
    *reinterpret_cast<u32x4*>(p) = a[0];
    *reinterpret_cast<u32x4*>(p + 8) = a[1];
  }
};

// Transposed 64 x 32 accumulator pair (rows = head columns, lane column = the owned row) -> memory.
__device__ __forceinline__ void store_owned(bf16_t* row_ptr, const f32x16 (&acc)[2], float mul, int hh) {
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const u32x2 w{pack_bf16x2(acc[d][4 * g] * mul, acc[d][4 * g + 1] * mul),
                    pack_bf16x2(acc[d][4 * g + 2] * mul, acc[d][4 * g + 3] * mul)};
      *reinterpret_cast<u32x2*>(row_ptr + 32 * d + 8 * g + 4 * hh) = w;
    }
}

// ----------------------------------------------------------------------------------------------
// Forward (MODE 0) and the dQ half of the backward (MODE 1): a wave owns 32 queries, K and V stream
// through LDS in blocks of 64 keys.
template <int MODE, bool DROP>
__global__ __launch_bounds__(256, 3) void attn_q_kernel(const AttnArgs a) {
  __shared__ __align__(16) bf16_t Kr[AT_BLK * AT_PR];                       // K by rows (S^T = K Q^T)
  __shared__ __align__(16) bf16_t Xt[AT_BLK * AT_PT];                       // forward: V, backward: K, transposed reads
  __shared__ __align__(16) bf16_t Vr[MODE == 1 ? AT_BLK * AT_PR : 8];       // backward: V by rows (dP^T = V dO^T)
  __shared__ __align__(16) float Bias[AT_BLK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hh = lane >> 5;
  const int N = a.N, b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * AT_OWN + wave * 32;
  const bool active = q0 < N;  // wave-uniform
  const size_t bo_qkv = (size_t)b * N * a.ld_qkv + (size_t)h * AT_DH;
  const size_t bo_o = (size_t)b * N * a.ld_o + (size_t)h * AT_DH;
  const bf16_t* kbase = a.k + bo_qkv;
  const bf16_t* vbase = a.v + bo_qkv;
  const float c2 = a.scale * AT_LOG2E;
  const int qrow = min(q0 + j, N - 1);

  bf16x8 qf[4], dof[4];
  float l2 = 0.f, dsum = 0.f;
  {
    const bf16_t* p = a.q + bo_qkv + (size_t)qrow * a.ld_qkv + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = as_frag(*reinterpret_cast<const u32x4*>(p + 16 * ks));
    if constexpr (MODE == 1) {
      const bf16_t* pd = a.dout + bo_o + (size_t)qrow * a.ld_o + 8 * hh;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) dof[ks] = as_frag(*reinterpret_cast<const u32x4*>(pd + 16 * ks));
      l2 = a.lse[((size_t)b * a.H + h) * N + qrow];
    }
  }

  f32x16 acc[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
  float m = -INFINITY, lsum = 0.f;

  Stage sk, sv;
  float breg = 0.f;
  auto issue = [&](int kb) {
    sk.issue(kbase, a.ld_qkv, kb * AT_BLK, N, tid);
    sv.issue(vbase, a.ld_qkv, kb * AT_BLK, N, tid);
    if (tid < AT_BLK) {
      const int key = kb * AT_BLK + tid;
      breg = 0.f;
      if (a.key_bias != nullptr) breg = fmaxf(a.key_bias[(size_t)b * N + min(key, N - 1)], AT_MIN_BIAS) * AT_LOG2E;
      if (key >= N) breg = -INFINITY;
    }
  };
  auto commit = [&]() {
    sk.commit(Kr, AT_PR, tid);
    if constexpr (MODE == 0) {
      sv.commit(Xt, AT_PT, tid);
    } else {
      sk.commit(Xt, AT_PT, tid);
      sv.commit(Vr, AT_PR, tid);
    }
    if (tid < AT_BLK) Bias[tid] = breg;
  };

  const int nblk = (N + AT_BLK - 1) / AT_BLK;
  // one pass over the keys: body(sub-block) runs on every 32-key sub-block that holds a key
  auto stream_keys = [&](auto&& body) {
    issue(0);
    commit();
    __syncthreads();
    for (int kb = 0; kb < nblk; ++kb) {
      if (kb + 1 < nblk) issue(kb + 1);
      if (active) body(kb);
      __syncthreads();
      if (kb + 1 < nblk) {
        commit();
        __syncthreads();
      }
    }
  };
  // body(sub-block, first key) on every 32-key sub-block of block kb that holds a key
  auto each_sub = [&](int kb, auto&& body) {
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
      if (kb * AT_BLK + sb * 32 >= N) break;
      body(sb, kb * AT_BLK + sb * 32);
    }
  };
  // scaled, biased scores of a sub-block in the base-2 domain (rows = keys, lane column = the query)
  auto scores = [&](int sb) -> f32x16 {
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 kf = as_frag(*reinterpret_cast<const u32x4*>(&Kr[(sb * 32 + j) * AT_PR + 16 * ks + 8 * hh]));
      s = mfma16(kf, qf[ks], s);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(&Bias[sb * 32 + 8 * g + 4 * hh]);
#pragma unroll
      for (int e = 0; e < 4; ++e) s[4 * g + e] = s[4 * g + e] * c2 + bv[e];
    }
    return s;
  };
  unsigned long long dseed = 0, dctr = 0;
  if constexpr (DROP) { dseed = a.drop_state[0]; dctr = a.drop_state[1]; }
  const unsigned long long drow = ((unsigned long long)b * a.H + h) * N + (unsigned long long)(q0 + j);
  const int dgroups = (N + 3) >> 2;
  // w[r] *= keep-scale of (query of this lane, key of register r)
  auto apply_drop = [&](int key0, f32x16& w) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float mk[4];
      drop_scale4(dseed, dctr, drow, dgroups, (key0 >> 2) + 2 * g + hh, a.drop_thr, a.drop_inv, mk);
#pragma unroll
      for (int e = 0; e < 4; ++e) w[4 * g + e] *= mk[e];
    }
  };
  auto second_product = [&](int sb, const f32x16& w) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const bf16x8 pb = pack8(w, t);
#pragma unroll
      for (int d = 0; d < 2; ++d) acc[d] = mfma16(tr_rows(Xt, sb * 32 + 16 * t, 32 * d, lane), pb, acc[d]);
    }
  };

  if constexpr (MODE == 0) {
    // The online-softmax step covers NS sub-blocks at once: their score products are independent MFMA
    // chains, and the running maximum, the rescaling of the accumulators and the lane exchange are paid
    // once per 64 keys.
    auto step = [&](auto ns_tag, int kb) {
      constexpr int NS = decltype(ns_tag)::value;
      f32x16 s[NS];
#pragma unroll
      for (int i = 0; i < NS; ++i) s[i] = scores(i);
      float mx = s[0][0];
#pragma unroll
      for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[i][r]);
      mx = fmaxf(mx, xchg32(mx));
      const float mn = fmaxf(m, mx);
      const float alpha = __builtin_amdgcn_exp2f(m - mn);
      m = mn;
      float ps = 0.f;
#pragma unroll
      for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s[i][r] = __builtin_amdgcn_exp2f(s[i][r] - mn);
          ps += s[i][r];
        }
      lsum = lsum * alpha + ps;
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[d][r] *= alpha;
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        if constexpr (DROP) apply_drop(kb * AT_BLK + 32 * i, s[i]);  // after the row sum: the softmax is normalised over every key
        second_product(i, s[i]);
      }
    };
    stream_keys([&](int kb) {
      if (kb * AT_BLK + 32 < N) step(std::integral_constant<int, 2>{}, kb);
      else step(std::integral_constant<int, 1>{}, kb);
    });
  } else {
    auto dprobs = [&](int sb) -> f32x16 {
      f32x16 dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) dp[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 vf = as_frag(*reinterpret_cast<const u32x4*>(&Vr[(sb * 32 + j) * AT_PR + 16 * ks + 8 * hh]));
        dp = mfma16(vf, dof[ks], dp);
      }
      return dp;
    };
    // Pass 1: the softmax-backward row term sum_m p[n,m] dp[n,m], from the very probabilities and dp of
    // pass 2.  (rowsum(dO * O) equals it in exact arithmetic, but O is stored rounded to bf16 and the
    // difference does not cancel in sum_m ds[n,m] k_m: it shows up as an error of dq proportional to |dO||O|.)
    float dacc = 0.f;
    stream_keys([&](int kb) {
      each_sub(kb, [&](int sb, int key0) {
        const f32x16 s = scores(sb);
        f32x16 dp = dprobs(sb);
        if constexpr (DROP) apply_drop(key0, dp);  // d(prob) = mask/(1-p) * d(dropped prob)
#pragma unroll
        for (int r = 0; r < 16; ++r) dacc += __builtin_amdgcn_exp2f(s[r] - l2) * dp[r];
      });
    });
    dsum = dacc + xchg32(dacc);
    if (active && hh == 0 && q0 + j < N) a.dvec[((size_t)b * a.H + h) * N + q0 + j] = dsum;
    stream_keys([&](int kb) {
      each_sub(kb, [&](int sb, int key0) {
        f32x16 s = scores(sb);
        f32x16 dp = dprobs(sb);
        if constexpr (DROP) apply_drop(key0, dp);
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(s[r] - l2) * (dp[r] - dsum);
        second_product(sb, s);
      });
    });
  }
  if (!active || q0 + j >= N) return;
  if constexpr (MODE == 0) {
    lsum += xchg32(lsum);
    store_owned(a.out + bo_o + (size_t)(q0 + j) * a.ld_o, acc, 1.f / lsum, hh);
    if (hh == 0) a.lse[((size_t)b * a.H + h) * N + q0 + j] = m + __builtin_amdgcn_logf(lsum);
  } else {
    store_owned(a.dq + (size_t)b * N * a.ld_d + (size_t)h * AT_DH + (size_t)(q0 + j) * a.ld_d, acc, a.scale, hh);
  }
}

// ----------------------------------------------------------------------------------------------
// dK / dV: a wave owns 32 keys, Q and dO stream through LDS in blocks of 64 queries.
template <bool DROP>
__global__ __launch_bounds__(256, DROP ? 1 : 2) void attn_kv_kernel(const AttnArgs a) {
  __shared__ __align__(16) bf16_t Qr[AT_BLK * AT_PR];
  __shared__ __align__(16) bf16_t Qt[AT_BLK * AT_PT];
  __shared__ __align__(16) bf16_t Dr[AT_BLK * AT_PR];
  __shared__ __align__(16) bf16_t Dt[AT_BLK * AT_PT];
  __shared__ __align__(16) float L2s[AT_BLK];
  __shared__ __align__(16) float Ds[AT_BLK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hh = lane >> 5;
  const int N = a.N, b = blockIdx.z, h = blockIdx.y;
  const int k0 = blockIdx.x * AT_OWN + wave * 32;
  const bool active = k0 < N;
  const size_t bo_qkv = (size_t)b * N * a.ld_qkv + (size_t)h * AT_DH;
  const size_t bo_o = (size_t)b * N * a.ld_o + (size_t)h * AT_DH;
  const bf16_t* qbase = a.q + bo_qkv;
  const bf16_t* dbase = a.dout + bo_o;
  const float* lbase = a.lse + ((size_t)b * a.H + h) * N;
  const float* vbase = a.dvec + ((size_t)b * a.H + h) * N;
  const float c2 = a.scale * AT_LOG2E;
  const int krow = min(k0 + j, N - 1);

  bf16x8 kf[4], vf[4];
  {
    const bf16_t* pk = a.k + bo_qkv + (size_t)krow * a.ld_qkv + 8 * hh;
    const bf16_t* pv = a.v + bo_qkv + (size_t)krow * a.ld_qkv + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kf[ks] = as_frag(*reinterpret_cast<const u32x4*>(pk + 16 * ks));
      vf[ks] = as_frag(*reinterpret_cast<const u32x4*>(pv + 16 * ks));
    }
  }
  float kbias = 0.f;
  if (a.key_bias != nullptr) kbias = fmaxf(a.key_bias[(size_t)b * N + krow], AT_MIN_BIAS) * AT_LOG2E;
  unsigned long long dseed = 0, dctr = 0;
  if constexpr (DROP) { dseed = a.drop_state[0]; dctr = a.drop_state[1]; }
  const unsigned long long rowbase = ((unsigned long long)b * a.H + h) * N;
  const int dgroups = (N + 3) >> 2;
  const int krow_true = k0 + j;

  f32x16 dk[2], dv[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) dk[d][r] = dv[d][r] = 0.f;

  Stage sq, sd;
  float lreg = 0.f, dreg = 0.f;
  auto issue = [&](int qb) {
    sq.issue(qbase, a.ld_qkv, qb * AT_BLK, N, tid);
    sd.issue(dbase, a.ld_o, qb * AT_BLK, N, tid);
    if (tid < AT_BLK) {
      const int q = qb * AT_BLK + tid;
      lreg = q < N ? lbase[q] : INFINITY;  // exp2(x - inf) = 0: rows past the end contribute nothing
      dreg = q < N ? vbase[q] : 0.f;
    }
  };
  auto commit = [&]() {
    sq.commit(Qr, AT_PR, tid);
    sq.commit(Qt, AT_PT, tid);
    sd.commit(Dr, AT_PR, tid);
    sd.commit(Dt, AT_PT, tid);
    if (tid < AT_BLK) { L2s[tid] = lreg; Ds[tid] = dreg; }
  };

  const int nblk = (N + AT_BLK - 1) / AT_BLK;
  issue(0);
  commit();
  __syncthreads();
  for (int qb = 0; qb < nblk; ++qb) {
    if (qb + 1 < nblk) issue(qb + 1);
    if (active) {
#pragma unroll
      for (int sb = 0; sb < 2; ++sb) {
        if (qb * AT_BLK + sb * 32 >= N) break;
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16x8 qa = as_frag(*reinterpret_cast<const u32x4*>(&Qr[(sb * 32 + j) * AT_PR + 16 * ks + 8 * hh]));
          const bf16x8 da = as_frag(*reinterpret_cast<const u32x4*>(&Dr[(sb * 32 + j) * AT_PR + 16 * ks + 8 * hh]));
          s = mfma16(qa, kf[ks], s);
          dp = mfma16(da, vf[ks], dp);
        }
        // s, dp: rows = queries of the sub-block, lane column = the owned key
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 lv = *reinterpret_cast<const f32x4*>(&L2s[sb * 32 + 8 * g + 4 * hh]);
          const f32x4 dvv = *reinterpret_cast<const f32x4*>(&Ds[sb * 32 + 8 * g + 4 * hh]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float p = __builtin_amdgcn_exp2f(s[4 * g + e] * c2 + kbias - lv[e]);
            float mk = 1.f;
            if constexpr (DROP) {  // the word of this lane's key in the call of the register's query row
              const unsigned long long row = rowbase + (unsigned long long)(qb * AT_BLK + sb * 32 + 8 * g + 4 * hh + e);
              const Philox4 r = philox4x32_10(dseed, row * (unsigned long long)dgroups + (unsigned long long)(krow_true >> 2), dctr);
              const unsigned int w4[4] = {r.x, r.y, r.z, r.w};
              mk = w4[krow_true & 3] >= a.drop_thr ? a.drop_inv : 0.f;
            }
            s[4 * g + e] = p * mk;                                  // dropped probability: dV = P~^T dO
            dp[4 * g + e] = p * (mk * dp[4 * g + e] - dvv[e]);
          }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const bf16x8 pb = pack8(s, t), db = pack8(dp, t);
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            dv[d] = mfma16(tr_rows(Dt, sb * 32 + 16 * t, 32 * d, lane), pb, dv[d]);
            dk[d] = mfma16(tr_rows(Qt, sb * 32 + 16 * t, 32 * d, lane), db, dk[d]);
          }
        }
      }
    }
    __syncthreads();
    if (qb + 1 < nblk) {
      commit();
      __syncthreads();
    }
  }
  if (!active || k0 + j >= N) return;
  const size_t off = (size_t)b * N * a.ld_d + (size_t)h * AT_DH + (size_t)(k0 + j) * a.ld_d;
  store_owned(a.dk + off, dk, a.scale, hh);
  store_owned(a.dv + off, dv, 1.f, hh);
}

// ----------------------------------------------------------------------------------------------
// LayerNorm over the last dimension with an optional residual add in front:
//   s = x (+ r);  y = (s - mean(s)) * rstd(s) * gamma + beta.
// One wave per row, the row stays in registers (D <= 512 * NCH).
template <bool BF16, bool PBF16, int NCH>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const void* x, const void* r, const void* gamma, const void* beta,
                                                            long rows, int D, float eps, void* s_out, void* y,
                                                            float* mean, float* rstd) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const size_t base = (size_t)row * D;
  float v[NCH][8];
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * 64 + lane) * 8;
    if (col < D) {
      load8<BF16>(x, base + col, v[c]);
      if (r != nullptr) {
        float t[8];
        load8<BF16>(r, base + col, t);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[c][e] += t[e];
        if constexpr (BF16) {
          // the sum is what the next layer sees: round it once, normalise the rounded value
#pragma unroll
          for (int e = 0; e < 8; ++e) v[c][e] = bf16_to_f32(f32_to_bf16(v[c][e]));
        }
      }
      if (s_out != nullptr) store8<BF16>(s_out, base + col, v[c]);
#pragma unroll
      for (int e = 0; e < 8; ++e) sum += v[c][e];
    }
  }
  const float mu = wave_sum(sum) / (float)D;
  float sq = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * 64 + lane) * 8;
    if (col < D) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[c][e] - mu; sq += d * d; }
    }
  }
  const float rs = 1.0f / sqrtf(wave_sum(sq) / (float)D + eps);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * 64 + lane) * 8;
    if (col < D) {
      float g[8], bt[8], o[8];
      load8<PBF16>(gamma, col, g);
      load8<PBF16>(beta, col, bt);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (v[c][e] - mu) * rs * g[e] + bt[e];
      store8<BF16>(y, base + col, o);
    }
  }
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

// dx = rstd * (dy*gamma - mean(dy*gamma) - xhat * mean(dy*gamma*xhat)) (+ ds_in);  dgamma = sum dy*xhat, dbeta = sum dy.
// A workgroup takes rows_per_wg consecutive rows (two rows per wave at a time) and writes its column sums
// to part[block][2][D]; layernorm_param_grad_kernel folds the blocks.  (Atomics onto the 2*D sums
// were tried first: every workgroup hits the same addresses and the kernel ran at a fifth of its
// memory-bound time.)
// rows per workgroup: two workgroups per CU on a 256-CU part take the whole matrix in one round
static inline int ln_rows_per_wg(long rows) {
  const long r = (rows + 511) / 512;
  return (int)(r < 8 ? 8 : r > 64 ? 64 : r);
}
template <bool BF16, bool PBF16, int NCH>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const void* dy, const void* s, const void* gamma,
                                                            const float* mean, const float* rstd, const void* ds_in,
                                                            long rows, int D, int rows_per_wg, void* dx, float* part) {
  extern __shared__ float red[];  // [3][2][NCH*512]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float ag[NCH][8], ab[NCH][8];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int e = 0; e < 8; ++e) ag[c][e] = ab[c][e] = 0.f;
  float g[NCH][8];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * 64 + lane) * 8;
    if (col < D) load8<PBF16>(gamma, col, g[c]);
  }
  // Two rows per wave are in flight (their loads are issued before either is reduced): with one, the
  // kernel waited out a full memory latency per row and ran at half its bandwidth.
  const long row_end = min(rows, ((long)blockIdx.x + 1) * rows_per_wg);
  for (long row0 = (long)blockIdx.x * rows_per_wg + wave; row0 < row_end; row0 += 8) {
    float dyv[2][NCH][8], xh[2][NCH][8], dsv[2][NCH][8];
    float mu[2], rs[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long row = min(row0 + 4 * u, rows - 1);  // clamped: the second row may not exist
      const size_t base = (size_t)row * D;
      mu[u] = mean[row];
      rs[u] = rstd[row];
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int col = (c * 64 + lane) * 8;
        if (col < D) {
          load8<BF16>(dy, base + col, dyv[u][c]);
          load8<BF16>(s, base + col, xh[u][c]);
          if (ds_in != nullptr) load8<BF16>(ds_in, base + col, dsv[u][c]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long row = row0 + 4 * u;
      if (row >= row_end) break;
      const size_t base = (size_t)row * D;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int col = (c * 64 + lane) * 8;
        if (col < D) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            xh[u][c][e] = (xh[u][c][e] - mu[u]) * rs[u];
            ag[c][e] += dyv[u][c][e] * xh[u][c][e];
            ab[c][e] += dyv[u][c][e];
            dyv[u][c][e] *= g[c][e];
            s1 += dyv[u][c][e];
            s2 += dyv[u][c][e] * xh[u][c][e];
          }
        }
      }
      s1 = wave_sum(s1) / (float)D;
      s2 = wave_sum(s2) / (float)D;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int col = (c * 64 + lane) * 8;
        if (col < D) {
          float o[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = rs[u] * (dyv[u][c][e] - s1 - xh[u][c][e] * s2);
          if (ds_in != nullptr) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += dsv[u][c][e];
          }
          store8<BF16>(dx, base + col, o);
        }
      }
    }
  }
  // fold the four waves, then one store per column
  const int DP = NCH * 512;
  if (wave > 0) {
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[((wave - 1) * 2 + 0) * DP + (c * 64 + lane) * 8 + e] = ag[c][e];
        red[((wave - 1) * 2 + 1) * DP + (c * 64 + lane) * 8 + e] = ab[c][e];
      }
  }
  __syncthreads();
  if (wave == 0) {
    float* pg = part + (size_t)blockIdx.x * 2 * D;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = (c * 64 + lane) * 8;
      if (col < D) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
          for (int w = 0; w < 3; ++w) {
            ag[c][e] += red[(w * 2 + 0) * DP + col + e];
            ab[c][e] += red[(w * 2 + 1) * DP + col + e];
          }
        }
        store8<false>(pg, col, ag[c]);
        store8<false>(pg + D, col, ab[c]);
      }
    }
  }
}

// dgamma, dbeta (D) <- part[nblk][2][D], in the parameters' dtype; fixed summation order.
// A workgroup folds 16 columns: thread (q, c) adds blocks q, q+16, ... of column c.
template <bool PBF16>
__global__ __launch_bounds__(256) void layernorm_param_grad_kernel(const float* part, int nblk, int D, void* dgamma,
                                                                   void* dbeta) {
  __shared__ float red[16][17];
  const int cl = threadIdx.x & 15, q = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;  // column of the (2, D) pair, flattened
  float acc = 0.f;
  if (c < 2 * D)
    for (int b = q; b < nblk; b += 16) acc += part[(size_t)b * 2 * D + c];
  red[q][cl] = acc;
  __syncthreads();
  if (q == 0 && c < 2 * D) {
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) v += red[i][cl];
    if (c < D) store1<PBF16>(dgamma, c, v);
    else store1<PBF16>(dbeta, c - D, v);
  }
}

// ----------------------------------------------------------------------------------------------
// GELU, exact form: y = x * Phi(x), dy/dx = Phi(x) + x * phi(x).
template <bool BF16, bool BWD>
__global__ __launch_bounds__(256) void gelu_kernel(const void* x, const void* dy, size_t n8, void* out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  float v[8], o[8];
  load8<BF16>(x, i * 8, v);
  if constexpr (BWD) {
    float d[8];
    load8<BF16>(dy, i * 8, d);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float cdf = 0.5f * (1.f + erff(v[e] * 0.70710678118654752f));
      const float pdf = 0.3989422804014327f * __expf(-0.5f * v[e] * v[e]);
      o[e] = d[e] * (cdf + v[e] * pdf);
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.5f * v[e] * (1.f + erff(v[e] * 0.70710678118654752f));
  }
  store8<BF16>(out, i * 8, o);
}

int attn_check(const void* q, const void* k, const void* v, long ld_qkv, long ld_o, int B, int H, int N, int dh) {
  VQAF_REQUIRE(q && k && v, VQAF_E_NULL);
  VQAF_REQUIRE(B > 0 && H > 0 && N > 0 && B <= 65535 && H <= 65535, VQAF_E_SHAPE);
  VQAF_REQUIRE(dh == AT_DH, VQAF_E_SHAPE);
  VQAF_REQUIRE(ld_qkv >= (long)H * dh && ld_o >= (long)H * dh, VQAF_E_SHAPE);
  VQAF_REQUIRE(ld_qkv % 8 == 0 && ld_o % 8 == 0, VQAF_E_ALIGN);
  VQAF_REQUIRE(vqaf_aligned16(q) && vqaf_aligned16(k) && vqaf_aligned16(v), VQAF_E_ALIGN);
  return VQAF_OK;
}

}  // namespace

namespace {
int attn_dropout(AttnArgs& a, float p, const uint64_t* state) {
  VQAF_REQUIRE(p >= 0.f && p < 1.f, VQAF_E_SHAPE);
  a.drop_state = nullptr;
  if (p > 0.f) {
    VQAF_REQUIRE(state != nullptr, VQAF_E_NULL);
    a.drop_state = state;
    a.drop_thr = (unsigned int)fmin((double)p * 4294967296.0, 4294967295.0);
    a.drop_inv = 1.0f / (1.0f - p);
  }
  return VQAF_OK;
}
}  // namespace

extern "C" int vqaf_attn_fwd(const void* q, const void* k, const void* v, long ld_qkv, const float* key_bias,
                             int B, int H, int N, int dh, float scale, float dropout_p,
                             const uint64_t* dropout_state, void* out, long ld_o, float* lse,
                             vqaf_stream_t stream) {
  VQAF_ENTRY();
  const int rc = attn_check(q, k, v, ld_qkv, ld_o, B, H, N, dh);
  if (rc != VQAF_OK) return rc;
  VQAF_REQUIRE(out && lse, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_aligned16(out), VQAF_E_ALIGN);
  AttnArgs a{};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v;
  a.out = (bf16_t*)out; a.key_bias = key_bias; a.lse = lse;
  a.ld_qkv = ld_qkv; a.ld_o = ld_o; a.N = N; a.H = H; a.scale = scale;
  const int rd = attn_dropout(a, dropout_p, dropout_state);
  if (rd != VQAF_OK) return rd;
  const dim3 grid((N + AT_OWN - 1) / AT_OWN, H, B);
  if (a.drop_state != nullptr) hipLaunchKernelGGL((attn_q_kernel<0, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((attn_q_kernel<0, false>), grid, dim3(256), 0, (hipStream_t)stream, a);
  VQAF_LAUNCH_CHECK();
}

extern "C" int vqaf_attn_bwd(const void* q, const void* k, const void* v, long ld_qkv, const float* key_bias,
                             const void* dout, long ld_o, const float* lse, int B, int H, int N,
                             int dh, float scale, float dropout_p, const uint64_t* dropout_state, void* dq,
                             void* dk, void* dv, long ld_d, float* dvec, vqaf_stream_t stream) {
  VQAF_ENTRY();
  const int rc = attn_check(q, k, v, ld_qkv, ld_o, B, H, N, dh);
  if (rc != VQAF_OK) return rc;
  VQAF_REQUIRE(dout && lse && dq && dk && dv && dvec, VQAF_E_NULL);
  VQAF_REQUIRE(ld_d >= (long)H * dh, VQAF_E_SHAPE);
  VQAF_REQUIRE(ld_d % 4 == 0, VQAF_E_ALIGN);
  VQAF_REQUIRE(vqaf_aligned16(dout), VQAF_E_ALIGN);
  VQAF_REQUIRE(((uintptr_t)dq & 7u) == 0 && ((uintptr_t)dk & 7u) == 0 && ((uintptr_t)dv & 7u) == 0, VQAF_E_ALIGN);
  AttnArgs a{};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v;
  a.dout = (const bf16_t*)dout;
  a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv;
  a.key_bias = key_bias; a.lse = const_cast<float*>(lse); a.dvec = dvec;
  a.ld_qkv = ld_qkv; a.ld_o = ld_o; a.ld_d = ld_d; a.N = N; a.H = H; a.scale = scale;
  const int rd = attn_dropout(a, dropout_p, dropout_state);
  if (rd != VQAF_OK) return rd;
  const dim3 grid((N + AT_OWN - 1) / AT_OWN, H, B);
  // dQ (and the softmax row terms for the next launch), then dK / dV
  if (a.drop_state != nullptr) {
    hipLaunchKernelGGL((attn_q_kernel<1, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(attn_kv_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
  } else {
    hipLaunchKernelGGL((attn_q_kernel<1, false>), grid, dim3(256), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(attn_kv_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
  }
  VQAF_LAUNCH_CHECK();
}

namespace {
template <bool BF16, bool PBF16>
void launch_ln_fwd(int nch, const void* x, const void* r, const void* gamma, const void* beta, long rows, int D,
                   float eps, void* s_out, void* y, float* mean, float* rstd, hipStream_t stream) {
  const dim3 grid((unsigned)((rows + 3) / 4));
#define LN_FWD(NCH) hipLaunchKernelGGL((layernorm_fwd_kernel<BF16, PBF16, NCH>), grid, dim3(256), 0, stream, x, r, gamma, beta, rows, D, eps, s_out, y, mean, rstd)
  if (nch == 1) LN_FWD(1); else if (nch == 2) LN_FWD(2); else LN_FWD(4);
#undef LN_FWD
}
template <bool BF16, bool PBF16>
void launch_ln_bwd(int nch, const void* dy, const void* s, const void* gamma, const float* mean, const float* rstd,
                   const void* ds_in, long rows, int D, void* dx, float* part, hipStream_t stream) {
  const int rpw = ln_rows_per_wg(rows);
  const dim3 grid((unsigned)((rows + rpw - 1) / rpw));
  const size_t lds = (size_t)3 * 2 * nch * 512 * sizeof(float);
#define LN_BWD(NCH) hipLaunchKernelGGL((layernorm_bwd_kernel<BF16, PBF16, NCH>), grid, dim3(256), lds, stream, dy, s, gamma, mean, rstd, ds_in, rows, D, rpw, dx, part)
  if (nch == 1) LN_BWD(1); else if (nch == 2) LN_BWD(2); else LN_BWD(4);
#undef LN_BWD
}
int ln_chunks(int D) { return D <= 512 ? 1 : D <= 1024 ? 2 : 4; }
}  // namespace

extern "C" int vqaf_layernorm_fwd(const void* x, const void* residual, const void* gamma, const void* beta,
                                  int param_dtype, long rows, int D, float eps, int dtype, void* sum_out, void* y,
                                  float* mean, float* rstd, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(x && gamma && beta && y && mean && rstd, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_dtype_ok(dtype) && vqaf_dtype_ok(param_dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(rows > 0 && D > 0 && D <= 2048 && D % 8 == 0 && (rows + 3) / 4 <= 0x7fffffffL, VQAF_E_SHAPE);
  VQAF_REQUIRE(vqaf_aligned16(x) && vqaf_aligned16(y) && vqaf_aligned16(gamma) && vqaf_aligned16(beta) &&
                   (!residual || vqaf_aligned16(residual)) && (!sum_out || vqaf_aligned16(sum_out)), VQAF_E_ALIGN);
  const int nch = ln_chunks(D);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VQAF_BF16) {
    if (param_dtype == VQAF_BF16) launch_ln_fwd<true, true>(nch, x, residual, gamma, beta, rows, D, eps, sum_out, y, mean, rstd, st);
    else launch_ln_fwd<true, false>(nch, x, residual, gamma, beta, rows, D, eps, sum_out, y, mean, rstd, st);
  } else {
    if (param_dtype == VQAF_BF16) launch_ln_fwd<false, true>(nch, x, residual, gamma, beta, rows, D, eps, sum_out, y, mean, rstd, st);
    else launch_ln_fwd<false, false>(nch, x, residual, gamma, beta, rows, D, eps, sum_out, y, mean, rstd, st);
  }
  VQAF_LAUNCH_CHECK();
}

extern "C" size_t vqaf_layernorm_bwd_workspace_bytes(long rows, int D) {
  if (rows <= 0 || D <= 0) return 0;
  const int rpw = ln_rows_per_wg(rows);
  return (size_t)((rows + rpw - 1) / rpw) * 2 * (size_t)D * sizeof(float);
}

extern "C" int vqaf_layernorm_bwd(const void* dy, const void* s, const void* gamma, int param_dtype,
                                  const float* mean, const float* rstd, const void* dsum_in, long rows, int D,
                                  int dtype, void* dx, void* dgamma, void* dbeta, void* workspace,
                                  size_t workspace_bytes, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(dy && s && gamma && mean && rstd && dx && dgamma && dbeta && workspace, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_dtype_ok(dtype) && vqaf_dtype_ok(param_dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(rows > 0 && D > 0 && D <= 2048 && D % 8 == 0, VQAF_E_SHAPE);
  VQAF_REQUIRE(workspace_bytes >= vqaf_layernorm_bwd_workspace_bytes(rows, D), VQAF_E_SHAPE);
  VQAF_REQUIRE(vqaf_aligned16(dy) && vqaf_aligned16(s) && vqaf_aligned16(dx) && vqaf_aligned16(gamma) &&
                   vqaf_aligned16(workspace) && (!dsum_in || vqaf_aligned16(dsum_in)), VQAF_E_ALIGN);
  const int nch = ln_chunks(D);
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)workspace;
  if (dtype == VQAF_BF16) {
    if (param_dtype == VQAF_BF16) launch_ln_bwd<true, true>(nch, dy, s, gamma, mean, rstd, dsum_in, rows, D, dx, part, st);
    else launch_ln_bwd<true, false>(nch, dy, s, gamma, mean, rstd, dsum_in, rows, D, dx, part, st);
  } else {
    if (param_dtype == VQAF_BF16) launch_ln_bwd<false, true>(nch, dy, s, gamma, mean, rstd, dsum_in, rows, D, dx, part, st);
    else launch_ln_bwd<false, false>(nch, dy, s, gamma, mean, rstd, dsum_in, rows, D, dx, part, st);
  }
  const int rpw = ln_rows_per_wg(rows);
  const int nblk = (int)((rows + rpw - 1) / rpw);
  const dim3 grid((2 * D + 15) / 16);
  if (param_dtype == VQAF_BF16) hipLaunchKernelGGL(layernorm_param_grad_kernel<true>, grid, dim3(256), 0, st, part, nblk, D, dgamma, dbeta);
  else hipLaunchKernelGGL(layernorm_param_grad_kernel<false>, grid, dim3(256), 0, st, part, nblk, D, dgamma, dbeta);
  VQAF_LAUNCH_CHECK();
}

extern "C" int vqaf_gelu_fwd(const void* x, size_t n, int dtype, void* y, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(x && y, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_dtype_ok(dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(n > 0 && n % 8 == 0 && n / 8 / 256 < 0x7fffffffUL, VQAF_E_SHAPE);
  VQAF_REQUIRE(vqaf_aligned16(x) && vqaf_aligned16(y), VQAF_E_ALIGN);
  const size_t n8 = n / 8;
  const dim3 grid((unsigned)((n8 + 255) / 256));
  if (dtype == VQAF_BF16) hipLaunchKernelGGL((gelu_kernel<true, false>), grid, dim3(256), 0, (hipStream_t)stream, x, nullptr, n8, y);
  else hipLaunchKernelGGL((gelu_kernel<false, false>), grid, dim3(256), 0, (hipStream_t)stream, x, nullptr, n8, y);
  VQAF_LAUNCH_CHECK();
}

extern "C" int vqaf_gelu_bwd(const void* x, const void* dy, size_t n, int dtype, void* dx, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(x && dy && dx, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_dtype_ok(dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(n > 0 && n % 8 == 0 && n / 8 / 256 < 0x7fffffffUL, VQAF_E_SHAPE);
  VQAF_REQUIRE(vqaf_aligned16(x) && vqaf_aligned16(dy) && vqaf_aligned16(dx), VQAF_E_ALIGN);
  const size_t n8 = n / 8;
  const dim3 grid((unsigned)((n8 + 255) / 256));
  if (dtype == VQAF_BF16) hipLaunchKernelGGL((gelu_kernel<true, true>), grid, dim3(256), 0, (hipStream_t)stream, x, dy, n8, dx);
  else hipLaunchKernelGGL((gelu_kernel<false, true>), grid, dim3(256), 0, (hipStream_t)stream, x, dy, n8, dx);
  VQAF_LAUNCH_CHECK();
}
