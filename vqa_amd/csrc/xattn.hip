// Token-side kernels of dense single-query cross-attention (the non-pruning branch).
//
// With one query per sample the key AND value projections fold out of the token loop:
//   s[n,h] = x_n . u_h                      (u_h = Wk_h^T (Wq_h q + bq_h) / sqrt(dh), score.hip)
//   p[h,n] = softmax_n s[n,h]
//   z_h    = sum_n p[h,n] x_n               (then attn_h = Wv_h z_h + bv_h, since sum_n p = 1)
// so the forward is two streaming passes over the tokens with H dot products / H axpys per token,
// and the backward is the same two shapes again plus one rank-2H update that writes grad_seq:
//   dp[n,h] = x_n . t_h  (t_h = Wv_h^T d_attn_h),  ds = p * (dp - sum_n p dp),
//   y_h = sum_n ds[h,n] x_n  (= d u_h),  grad_seq[n,:] = sum_h ds[h,n] u_h + p[h,n] t_h.
// Everything here is bandwidth-bound; the (B,H,D)-sized projections around it are small GEMMs.
#include <algorithm>
#include "common.h"

namespace {

constexpr int XA_THREADS = 256;
constexpr int XA_LPR = 16;                   // lanes per row slice (8 columns each)
constexpr int XA_ROWS = XA_THREADS / XA_LPR;  // rows in flight per step
constexpr int XA_SLW = 8 * XA_LPR;            // columns per workgroup

// p[b,h,n] = softmax over n of s[b,n,h].  One workgroup per sample.
template <int HP>
__global__ void __launch_bounds__(XA_THREADS)
softmax_heads_kernel(const float* __restrict__ s, int N, int H, float* __restrict__ p) {
  __shared__ float red[XA_THREADS / 64][HP];
  __shared__ float stat[2][HP];
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float* sb = s + (size_t)b * N * H;
  float m[HP], z[HP];
#pragma unroll
  for (int h = 0; h < HP; ++h) { m[h] = -INFINITY; z[h] = 0.f; }
  for (int n = tid; n < N; n += XA_THREADS)
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
    for (int i = 1; i < XA_THREADS / 64; ++i) v = fmaxf(v, red[i][tid]);
    stat[0][tid] = v;
  }
  __syncthreads();
#pragma unroll
  for (int h = 0; h < HP; ++h) m[h] = stat[0][h];
  for (int n = tid; n < N; n += XA_THREADS)
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
    for (int i = 1; i < XA_THREADS / 64; ++i) v += red[i][tid];
    stat[1][tid] = v;
  }
  __syncthreads();
  for (int n = tid; n < N; n += XA_THREADS)
#pragma unroll
    for (int h = 0; h < HP; ++h)
      if (h < H) p[((size_t)b * H + h) * N + n] = expf(sb[(size_t)n * H + h] - m[h]) / stat[1][h];
}

// ds[b,h,n] = p[b,h,n] * (dp[b,n,h] - sum_m p[b,h,m] dp[b,m,h]).  One workgroup per sample.
template <int HP>
__global__ void __launch_bounds__(XA_THREADS)
softmax_heads_bwd_kernel(const float* __restrict__ p, const float* __restrict__ dp, int N, int H,
                         float* __restrict__ ds) {
  __shared__ float red[XA_THREADS / 64][HP];
  __shared__ float stat[HP];
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  float acc[HP];
#pragma unroll
  for (int h = 0; h < HP; ++h) acc[h] = 0.f;
  for (int n = tid; n < N; n += XA_THREADS)
#pragma unroll
    for (int h = 0; h < HP; ++h)
      if (h < H) acc[h] = fmaf(p[((size_t)b * H + h) * N + n], dp[((size_t)b * N + n) * H + h], acc[h]);
#pragma unroll
  for (int h = 0; h < HP; ++h) {
    const float v = wave_sum(acc[h]);
    if (lane == 0) red[wave][h] = v;
  }
  __syncthreads();
  if (tid < HP) {
    float v = red[0][tid];
    for (int i = 1; i < XA_THREADS / 64; ++i) v += red[i][tid];
    stat[tid] = v;
  }
  __syncthreads();
  for (int n = tid; n < N; n += XA_THREADS)
#pragma unroll
    for (int h = 0; h < HP; ++h)
      if (h < H) {
        const size_t i = ((size_t)b * H + h) * N + n;
        ds[i] = p[i] * (dp[((size_t)b * N + n) * H + h] - stat[h]);
      }
}

// z[b,h,:] = sum_n c[b,h,n] x[b,n,:].  Workgroup = (sample, 128-column slice); every token row is
// read once; the sample's coefficients sit in LDS token-major ([n][HP]: a lane fetches the HP
// weights of its row with 16-byte reads); four rows per lane are in flight; fixed-order reduction,
// no atomics.
template <bool BF16, int HP>
__global__ void __launch_bounds__(XA_THREADS)
weighted_token_sum_kernel(const void* __restrict__ seq, const float* __restrict__ c, int N, int D,
                          int H, float* __restrict__ z) {
  constexpr int U = 4;
  extern __shared__ __attribute__((aligned(16))) float xsm[];
  float* cs = xsm;                       // [N][HP]
  float* red = xsm + (size_t)HP * N;     // [XA_ROWS][XA_SLW]
  const int tid = threadIdx.x, b = blockIdx.y, g = tid / XA_LPR, li = tid % XA_LPR;
  const int d0 = blockIdx.x * XA_SLW + li * 8;
  for (int i = tid; i < HP * N; i += XA_THREADS) {
    const int h = i / N, n = i % N;
    cs[n * HP + h] = h < H ? c[((size_t)b * H + h) * N + n] : 0.f;
  }
  __syncthreads();
  float acc[HP][8];
#pragma unroll
  for (int h = 0; h < HP; ++h)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[h][i] = 0.f;
  if (d0 < D) {
    for (int n0 = g; n0 < N; n0 += U * XA_ROWS) {
      float x[U][8];
#pragma unroll
      for (int u = 0; u < U; ++u)  // rows past the end re-read the last one; their weight is zeroed below
        load8<BF16>(seq, ((size_t)b * N + min(n0 + u * XA_ROWS, N - 1)) * D + d0, x[u]);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int n = n0 + u * XA_ROWS;
        const bool in = n < N;
        const float* cw = cs + (size_t)min(n, N - 1) * HP;
        float w[HP];
        if constexpr (HP >= 4) {
#pragma unroll
          for (int h = 0; h < HP; h += 4) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(cw + h);
            w[h] = q.x; w[h + 1] = q.y; w[h + 2] = q.z; w[h + 3] = q.w;
          }
        } else {
#pragma unroll
          for (int h = 0; h < HP; ++h) w[h] = cw[h];
        }
#pragma unroll
        for (int h = 0; h < HP; ++h) {
          const float wh = in ? w[h] : 0.f;
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[h][i] = fmaf(wh, x[u][i], acc[h][i]);
        }
      }
    }
  }
  for (int h = 0; h < H; ++h) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float v = 0.f;
#pragma unroll
      for (int hh = 0; hh < HP; ++hh) v = (hh == h) ? acc[hh][i] : v;
      red[tid * 8 + i] = v;
    }
    __syncthreads();
    if (tid < XA_SLW) {
      float v = 0.f;
      for (int gg = 0; gg < XA_ROWS; ++gg) v += red[(gg * XA_LPR + tid / 8) * 8 + (tid & 7)];
      const int d = blockIdx.x * XA_SLW + tid;
      if (d < D) z[((size_t)b * H + h) * D + d] = v;
    }
  }
}

// out[b,n,:] = sum_h ca[b,h,n] va[b,h,:] + cb[b,h,n] vb[b,h,:].  Same geometry; pure write.  The
// 2*HP coefficients of a row sit together in LDS ([n][2*HP]) and are fetched with 16-byte reads;
// two rows per lane are computed side by side.
template <bool BF16, int HP>
__global__ void __launch_bounds__(XA_THREADS)
token_outer_kernel(const float* __restrict__ ca, const float* __restrict__ va,
                   const float* __restrict__ cb, const float* __restrict__ vb, int N, int D, int H,
                   void* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float xsm[];
  float* cs = xsm;  // [N][2*HP]: a_0..a_{HP-1}, b_0..b_{HP-1}
  const int tid = threadIdx.x, b = blockIdx.y, g = tid / XA_LPR, li = tid % XA_LPR;
  const int d0 = blockIdx.x * XA_SLW + li * 8;
  for (int i = tid; i < HP * N; i += XA_THREADS) {
    const int h = i / N, n = i % N;
    cs[n * 2 * HP + h] = h < H ? ca[((size_t)b * H + h) * N + n] : 0.f;
    cs[n * 2 * HP + HP + h] = h < H ? cb[((size_t)b * H + h) * N + n] : 0.f;
  }
  __syncthreads();
  if (d0 >= D) return;
  float ra[HP][8], rb[HP][8];
#pragma unroll
  for (int h = 0; h < HP; ++h) {
    if (h < H) {
      load8<false>(va, ((size_t)b * H + h) * D + d0, ra[h]);
      load8<false>(vb, ((size_t)b * H + h) * D + d0, rb[h]);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) ra[h][i] = rb[h][i] = 0.f;
    }
  }
  auto row = [&](int n, float (&o)[8]) {
    const float* cw = cs + (size_t)n * 2 * HP;
    float w[2 * HP];
    if constexpr (HP >= 2) {
#pragma unroll
      for (int h = 0; h < 2 * HP; h += 4) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(cw + h);
        w[h] = q.x; w[h + 1] = q.y; w[h + 2] = q.z; w[h + 3] = q.w;
      }
    } else {
      w[0] = cw[0]; w[1] = cw[1];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = 0.f;
#pragma unroll
    for (int h = 0; h < HP; ++h)
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = fmaf(w[h], ra[h][i], fmaf(w[HP + h], rb[h][i], o[i]));
  };
  int n = g;
  for (; n + XA_ROWS < N; n += 2 * XA_ROWS) {
    float o0[8], o1[8];
    row(n, o0);
    row(n + XA_ROWS, o1);
    store8<BF16>(out, ((size_t)b * N + n) * D + d0, o0);
    store8<BF16>(out, ((size_t)b * N + n + XA_ROWS) * D + d0, o1);
  }
  if (n < N) {
    float o0[8];
    row(n, o0);
    store8<BF16>(out, ((size_t)b * N + n) * D + d0, o0);
  }
}

// ---- Both attention branches of a dense co-attention in the same launches ------------------------
// The kernels below take up to XA_MAX_MODS problems of one batch size and width (different token counts
// and buffers; blockIdx.z picks the problem) and fuse the per-sample softmax steps into the token passes:
//   softmax_wsum_multi_kernel   s -> p (written once, by the first column slice) and z = sum_n p x
//   token_bwd_multi_kernel      (p, dp) -> ds in LDS, du = sum_n ds x and grad_seq = ds u + p t, one pass
// The arithmetic and the order of every sum are those of softmax_heads(_bwd)_kernel,
// weighted_token_sum_kernel and token_outer_kernel, so the results are bit-identical to the separate launches.
#ifdef XA_STAMPS  // developer build (make stamps): s_memrealtime stamps of token_bwd_multi_kernel's phases per workgroup
__device__ unsigned long long xa_stamp_buf[8192 * 8];
#define XA_STAMP(i)                                                                                        \
  do {                                                                                                     \
    const unsigned int wg_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);               \
    if (threadIdx.x == 0 && wg_ < 8192) xa_stamp_buf[wg_ * 8 + (i)] = __builtin_amdgcn_s_memrealtime();     \
  } while (0)
#else
#define XA_STAMP(i) do {} while (0)
#endif
constexpr int XA_MAX_MODS = 4;
struct XaFwdMod {
  const void* seq;  // (B,N,D)
  const float* s;   // (B,N,H) raw scores
  float* p;         // (B,H,N) out
  float* z;         // (B,H,D) out
  int N;
};
struct XaFwdMods { XaFwdMod m[XA_MAX_MODS]; };
struct XaBwdMod {
  const void* seq;  // (B,N,D)
  const float* p;   // (B,H,N)
  const float* dp;  // (B,N,H)
  const float* u;   // (B,H,D)
  const float* t;   // (B,H,D)
  float* du;        // (B,H,D) out or NULL
  void* gseq;       // (B,N,D) out or NULL
  int N;
};
struct XaBwdMods { XaBwdMod m[XA_MAX_MODS]; };

// HP consecutive floats of an LDS row (16-byte aligned when HP >= 4) as 16-byte accesses: eight lanes that own
// consecutive rows of pitch HP + 4 or 2 * HP + 4 floats then touch disjoint banks, where 4-byte accesses to the
// same column of consecutive rows collide four ways.
template <int HP>
__device__ __forceinline__ void xa_row_store(float* row, const float (&v)[HP]) {
  if constexpr (HP >= 4) {
#pragma unroll
    for (int h = 0; h < HP; h += 4) *reinterpret_cast<f32x4*>(row + h) = f32x4{v[h], v[h + 1], v[h + 2], v[h + 3]};
  } else {
#pragma unroll
    for (int h = 0; h < HP; ++h) row[h] = v[h];
  }
}
template <int HP>
__device__ __forceinline__ void xa_row_load(const float* row, float (&v)[HP]) {
  if constexpr (HP >= 4) {
#pragma unroll
    for (int h = 0; h < HP; h += 4) {
      const f32x4 q = *reinterpret_cast<const f32x4*>(row + h);
      v[h] = q.x; v[h + 1] = q.y; v[h + 2] = q.z; v[h + 3] = q.w;
    }
  } else {
#pragma unroll
    for (int h = 0; h < HP; ++h) v[h] = row[h];
  }
}

// Fixed-order reduction over the workgroup of HP per-thread values: wave, then waves in index order.
template <int HP, bool MAX>
__device__ __forceinline__ void xa_block_reduce(float (&v)[HP], float (*red)[HP], float* stat) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int h = 0; h < HP; ++h) {
    const float r = MAX ? wave_max(v[h]) : wave_sum(v[h]);
    if (lane == 0) red[wave][h] = r;
  }
  __syncthreads();
  if (tid < HP) {
    float r = red[0][tid];
    for (int i = 1; i < XA_THREADS / 64; ++i) r = MAX ? fmaxf(r, red[i][tid]) : r + red[i][tid];
    stat[tid] = r;
  }
  __syncthreads();
}

typedef float xa_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ xa_f32x2 xa_fma2(float w, xa_f32x2 x, xa_f32x2 a) {
  // two FMAs in one packed instruction (v_pk_fma_f32): each element is the same correctly rounded fmaf
  return __builtin_elementwise_fma(xa_f32x2{w, w}, x, a);
}
// A token row piece (8 columns) as it arrives from memory: bf16 rows stay packed until they are multiplied.
template <bool BF16>
struct XaRaw {
  u32x4 r[BF16 ? 1 : 2];
};
template <bool BF16>
__device__ __forceinline__ XaRaw<BF16> xa_load_raw(const void* __restrict__ seq, size_t e) {
  XaRaw<BF16> x;
  if constexpr (BF16) {
    x.r[0] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(seq) + e);
  } else {
    x.r[0] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const float*>(seq) + e);
    x.r[1] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const float*>(seq) + e + 4);
  }
  return x;
}
template <bool BF16>
__device__ __forceinline__ void xa_unpack(const XaRaw<BF16>& x, xa_f32x2 (&v)[4]) {
  if constexpr (BF16) {
    const u32x4 r = x.r[0];
    v[0] = xa_f32x2{bf16_lo(r.x), bf16_hi(r.x)}; v[1] = xa_f32x2{bf16_lo(r.y), bf16_hi(r.y)};
    v[2] = xa_f32x2{bf16_lo(r.z), bf16_hi(r.z)}; v[3] = xa_f32x2{bf16_lo(r.w), bf16_hi(r.w)};
  } else {
    const u32x4 a = x.r[0], c = x.r[1];
    v[0] = xa_f32x2{__uint_as_float(a.x), __uint_as_float(a.y)}; v[1] = xa_f32x2{__uint_as_float(a.z), __uint_as_float(a.w)};
    v[2] = xa_f32x2{__uint_as_float(c.x), __uint_as_float(c.y)}; v[3] = xa_f32x2{__uint_as_float(c.z), __uint_as_float(c.w)};
  }
}

// The weighted-sum pass of weighted_token_sum_kernel over coefficients already token-major in LDS
// (cs[n * CP + h], h < HP) and its fixed-order fold into z[b,h,slice].  Rows are requested one batch ahead of
// the batch being multiplied (`first` holds the first batch, requested by the caller before its own prologue).
constexpr int XA_WS_U = 2;  // rows per lane and batch
template <bool BF16>
__device__ __forceinline__ void xa_ws_fetch(const void* __restrict__ seq, int b, int N, int D, int n0,
                                            XaRaw<BF16> (&dst)[XA_WS_U]) {
  const int li = threadIdx.x % XA_LPR;
  const int d0 = min((int)blockIdx.x * XA_SLW + li * 8, D - 8);  // (a slice past the width is never accumulated)
#pragma unroll
  for (int u = 0; u < XA_WS_U; ++u)  // rows past the end re-read the last one; their weight is zeroed
    dst[u] = xa_load_raw<BF16>(seq, ((size_t)b * N + min(n0 + u * XA_ROWS, N - 1)) * D + d0);
}
template <bool BF16, int HP, int CP>
__device__ __forceinline__ void xa_weighted_sum(const void* __restrict__ seq, const float* cs, float* red, int b,
                                                int N, int D, int H, float* __restrict__ z,
                                                XaRaw<BF16> (&cur)[XA_WS_U]) {
  constexpr int U = XA_WS_U;
  const int tid = threadIdx.x, g = tid / XA_LPR, li = tid % XA_LPR;
  const int d0 = blockIdx.x * XA_SLW + li * 8;
  xa_f32x2 acc[HP][4];
#pragma unroll
  for (int h = 0; h < HP; ++h)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[h][i] = xa_f32x2{0.f, 0.f};
  auto multiply = [&](const XaRaw<BF16> (&rows)[U], int n0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      xa_f32x2 x[4];
      xa_unpack<BF16>(rows[u], x);
      const int n = n0 + u * XA_ROWS;
      const bool in = n < N;
      const float* cw = cs + (size_t)min(n, N - 1) * CP;
      float w[HP];
      if constexpr (HP >= 4) {
#pragma unroll
        for (int h = 0; h < HP; h += 4) {
          const f32x4 q = *reinterpret_cast<const f32x4*>(cw + h);
          w[h] = q.x; w[h + 1] = q.y; w[h + 2] = q.z; w[h + 3] = q.w;
        }
      } else {
#pragma unroll
        for (int h = 0; h < HP; ++h) w[h] = cw[h];
      }
#pragma unroll
      for (int h = 0; h < HP; ++h) {
        const float wh = in ? w[h] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[h][i] = xa_fma2(wh, x[i], acc[h][i]);
      }
    }
  };
  if (d0 < D) {
    // two register sets take turns (no copies: a copy of a register that a load is still writing would wait
    // for the load): while one batch is multiplied the next one is on its way
    XaRaw<BF16> other[U];
    for (int n0 = g; n0 < N; n0 += 2 * U * XA_ROWS) {
      xa_ws_fetch<BF16>(seq, b, N, D, n0 + U * XA_ROWS, other);
      multiply(cur, n0);
      xa_ws_fetch<BF16>(seq, b, N, D, n0 + 2 * U * XA_ROWS, cur);
      multiply(other, n0 + U * XA_ROWS);  // rows past the end carry zero weight
    }
  }
  for (int h = 0; h < H; ++h) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float v = 0.f;
#pragma unroll
      for (int hh = 0; hh < HP; ++hh) v = (hh == h) ? acc[hh][i >> 1][i & 1] : v;
      red[tid * 8 + i] = v;
    }
    __syncthreads();
    if (tid < XA_SLW) {
      float v = 0.f;
      for (int gg = 0; gg < XA_ROWS; ++gg) v += red[(gg * XA_LPR + tid / 8) * 8 + (tid & 7)];
      const int d = blockIdx.x * XA_SLW + tid;
      if (d < D) z[((size_t)b * H + h) * D + d] = v;
    }
  }
}

template <bool BF16, int HP>
__global__ void __launch_bounds__(XA_THREADS)
softmax_wsum_multi_kernel(XaFwdMods mods, int D, int H, float* __restrict__ zero_ptr, unsigned int zero_count) {
  extern __shared__ __attribute__((aligned(16))) float xsm[];
  {  // a range cleared on the side (outputs that the launches behind this one build with atomic adds)
    const unsigned int nwg = gridDim.x * gridDim.y * gridDim.z;
    const unsigned int wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    for (unsigned int i = wg * XA_THREADS + threadIdx.x; i < zero_count; i += nwg * XA_THREADS) zero_ptr[i] = 0.f;
  }
  const XaFwdMod& M = mods.m[blockIdx.z];
  const int N = M.N, b = blockIdx.y, tid = threadIdx.x;
  // a token's coefficients sit together; the row pitch (HP + 4 floats) spreads consecutive tokens over the LDS
  // banks (with a pitch of HP floats, lanes that own consecutive tokens collide 8 ways on every access)
  constexpr int CP = HP + 4;
  float* cs = xsm;                                  // [N][CP]
  float* red = xsm;                                 // [XA_THREADS * 8]: the final fold, when cs has been consumed
  float* small = xsm + max(CP * N, XA_THREADS * 8);
  float (*wred)[HP] = reinterpret_cast<float (*)[HP]>(small);  // [waves][HP] of the block reductions
  float* stat = small + (XA_THREADS / 64) * HP;     // [2][HP]
  const float* sb = M.s + (size_t)b * N * H;
  XA_STAMP(0);
  XaRaw<BF16> first[XA_WS_U];  // the first rows travel while the softmax is computed
  xa_ws_fetch<BF16>(M.seq, b, N, D, tid / XA_LPR, first);
  float m[HP], zz[HP];
#pragma unroll
  for (int h = 0; h < HP; ++h) { m[h] = -INFINITY; zz[h] = 0.f; }
  for (int n0 = tid; n0 < N; n0 += 4 * XA_THREADS) {  // four tokens per thread and round, all loads requested first
    float sv[4][HP];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = min(n0 + r * XA_THREADS, N - 1);
#pragma unroll
      for (int h = 0; h < HP; ++h) sv[r][h] = sb[(size_t)n * H + min(h, H - 1)];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + r * XA_THREADS;
      if (n < N) {
        float v[HP];
#pragma unroll
        for (int h = 0; h < HP; ++h) {
          v[h] = h < H ? sv[r][h] : 0.f;
          if (h < H) m[h] = fmaxf(m[h], v[h]);
        }
        xa_row_store<HP>(cs + n * CP, v);
      }
    }
  }
  XA_STAMP(4);
  xa_block_reduce<HP, true>(m, wred, stat);
  XA_STAMP(5);
#pragma unroll
  for (int h = 0; h < HP; ++h) m[h] = stat[h];
  __syncthreads();  // stat is rewritten below
  for (int n = tid; n < N; n += XA_THREADS) {  // the exponentials are kept: the second pass only divides
    float v[HP];
    xa_row_load<HP>(cs + n * CP, v);
#pragma unroll
    for (int h = 0; h < HP; ++h) {
      v[h] = h < H ? expf(v[h] - m[h]) : 0.f;
      if (h < H) zz[h] += v[h];
    }
    xa_row_store<HP>(cs + n * CP, v);
  }
  xa_block_reduce<HP, false>(zz, wred, stat + HP);
  float* pb = M.p + (size_t)b * H * N;
  for (int n = tid; n < N; n += XA_THREADS) {
    float v[HP];
    xa_row_load<HP>(cs + n * CP, v);
#pragma unroll
    for (int h = 0; h < HP; ++h)
      if (h < H) {
        v[h] = v[h] / stat[HP + h];
        if (blockIdx.x == 0) pb[(size_t)h * N + n] = v[h];
      }
    xa_row_store<HP>(cs + n * CP, v);
  }
  __syncthreads();
  XA_STAMP(1);
  XA_STAMP(2);
  xa_weighted_sum<BF16, HP, CP>(M.seq, cs, red, b, N, D, H, M.z, first);
  XA_STAMP(3);
}

template <bool BF16, int HP, bool NEED_U, bool NEED_SEQ>
__global__ void __launch_bounds__(XA_THREADS, 3)  // three workgroups per CU: the short (text) problems run beside the long ones
token_bwd_multi_kernel(XaBwdMods mods, int D, int H) {
  extern __shared__ __attribute__((aligned(16))) float xsm[];
  const XaBwdMod& M = mods.m[blockIdx.z];
  const int N = M.N, b = blockIdx.y, tid = threadIdx.x, g = tid / XA_LPR, li = tid % XA_LPR;
  // row pitch 2 * HP + 4 floats: consecutive tokens fall on different LDS banks (a pitch of 2 * HP floats makes
  // the lanes of the staging loops collide 16 ways: 14 of this kernel's 16 us prologue at N = 577)
  constexpr int CP = 2 * HP + 4;
  float* cs = xsm;                                  // [N][CP]: ds_0..ds_{HP-1}, p_0..p_{HP-1}
  float* red = xsm;                                 // [XA_THREADS * 8]: the final fold, when cs has been consumed
  float* small = xsm + max(CP * N, XA_THREADS * 8);
  float (*wred)[HP] = reinterpret_cast<float (*)[HP]>(small);  // [waves][HP] of the block reduction
  float* stat = small + (XA_THREADS / 64) * HP;     // [HP]
  const float* pb = M.p + (size_t)b * H * N;
  const float* dpb = M.dp + (size_t)b * N * H;
  XA_STAMP(0);
  XaRaw<BF16> first[XA_WS_U];  // the first rows of the folded-query gradient's pass: requested up front when that
  // pass follows the prologue directly, else after the token gradient (their registers are needed there)
  if constexpr (NEED_U && !NEED_SEQ) xa_ws_fetch<BF16>(M.seq, b, N, D, g, first);
  float acc[HP];
#pragma unroll
  for (int h = 0; h < HP; ++h) acc[h] = 0.f;
  // four tokens per thread and round: every load of a round is requested before the first is used (indices past
  // the end are clamped, their values dropped), so a round costs one memory latency
  for (int n0 = tid; n0 < N; n0 += 4 * XA_THREADS) {
    float pv[4][HP], dv[4][HP];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = min(n0 + r * XA_THREADS, N - 1);
#pragma unroll
      for (int h = 0; h < HP; ++h) {
        const int hc = min(h, H - 1);
        pv[r][h] = pb[(size_t)hc * N + n];
        dv[r][h] = dpb[(size_t)n * H + hc];
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + r * XA_THREADS;
      if (n < N) {
        float p_[HP], d_[HP];
#pragma unroll
        for (int h = 0; h < HP; ++h) {
          p_[h] = h < H ? pv[r][h] : 0.f;
          d_[h] = h < H ? dv[r][h] : 0.f;
          if (h < H) acc[h] = fmaf(p_[h], d_[h], acc[h]);
        }
        xa_row_store<HP>(cs + n * CP + HP, p_);
        xa_row_store<HP>(cs + n * CP, d_);
      }
    }
  }
  XA_STAMP(4);
  xa_block_reduce<HP, false>(acc, wred, stat);
  XA_STAMP(5);
  for (int n = tid; n < N; n += XA_THREADS) {
    float p_[HP], d_[HP];
    xa_row_load<HP>(cs + n * CP + HP, p_);
    xa_row_load<HP>(cs + n * CP, d_);
#pragma unroll
    for (int h = 0; h < HP; ++h) d_[h] = h < H ? p_[h] * (d_[h] - stat[h]) : 0.f;
    xa_row_store<HP>(cs + n * CP, d_);
  }
  __syncthreads();
  XA_STAMP(1);
  if constexpr (NEED_SEQ) {
    const int d0 = blockIdx.x * XA_SLW + li * 8;
    if (d0 < D) {
      xa_f32x2 ra[HP][4], rb[HP][4];
#pragma unroll
      for (int h = 0; h < HP; ++h) {
        if (h < H) {
          xa_unpack<false>(xa_load_raw<false>(M.u, ((size_t)b * H + h) * D + d0), ra[h]);
          xa_unpack<false>(xa_load_raw<false>(M.t, ((size_t)b * H + h) * D + d0), rb[h]);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) ra[h][i] = rb[h][i] = xa_f32x2{0.f, 0.f};
        }
      }
      auto row = [&](int n, float (&o)[8]) {
        const float* cw = cs + (size_t)n * CP;
        float w[2 * HP];
        if constexpr (HP >= 2) {
#pragma unroll
          for (int h = 0; h < 2 * HP; h += 4) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(cw + h);
            w[h] = q.x; w[h + 1] = q.y; w[h + 2] = q.z; w[h + 3] = q.w;
          }
        } else {
          w[0] = cw[0]; w[1] = cw[1];
        }
        xa_f32x2 o2[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o2[i] = xa_f32x2{0.f, 0.f};
#pragma unroll
        for (int h = 0; h < HP; ++h)
#pragma unroll
          for (int i = 0; i < 4; ++i) o2[i] = xa_fma2(w[h], ra[h][i], xa_fma2(w[HP + h], rb[h][i], o2[i]));
#pragma unroll
        for (int i = 0; i < 4; ++i) { o[2 * i] = o2[i].x; o[2 * i + 1] = o2[i].y; }
      };
      // one row at a time: with u and t of eight heads in registers (128 of them) a second row in flight would
      // cost the third workgroup per CU
      for (int n = g; n < N; n += XA_ROWS) {
        float o0[8];
        row(n, o0);
        store8<BF16>(M.gseq, ((size_t)b * N + n) * D + d0, o0);
      }
    }
  }
  XA_STAMP(2);
  if constexpr (NEED_U && NEED_SEQ) xa_ws_fetch<BF16>(M.seq, b, N, D, g, first);
  if constexpr (NEED_U) xa_weighted_sum<BF16, HP, CP>(M.seq, cs, red, b, N, D, H, M.du, first);
  XA_STAMP(3);
}

inline int pad_heads(int H) {
  int p = 1;
  while (p < H) p <<= 1;
  return p;
}

}  // namespace

#define XA_BY_HP(KERNEL, ...)                                                      \
  switch (HP) {                                                                    \
    case 1: hipLaunchKernelGGL(KERNEL<1>, __VA_ARGS__); break;                     \
    case 2: hipLaunchKernelGGL(KERNEL<2>, __VA_ARGS__); break;                     \
    case 4: hipLaunchKernelGGL(KERNEL<4>, __VA_ARGS__); break;                     \
    case 8: hipLaunchKernelGGL(KERNEL<8>, __VA_ARGS__); break;                     \
    default: hipLaunchKernelGGL(KERNEL<16>, __VA_ARGS__); break;                   \
  }

extern "C" int vqaf_softmax_heads(const float* s, int B, int N, int H, float* p, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(s && p, VQAF_E_NULL);
  VQAF_REQUIRE(B > 0 && N > 0 && H > 0 && H <= 16, VQAF_E_SHAPE);
  const int HP = pad_heads(H);
  XA_BY_HP(softmax_heads_kernel, dim3(B), dim3(XA_THREADS), 0, (hipStream_t)stream, s, N, H, p);
  VQAF_LAUNCH_CHECK();
}

extern "C" int vqaf_softmax_heads_bwd(const float* p, const float* dp, int B, int N, int H, float* ds,
                                      vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(p && dp && ds, VQAF_E_NULL);
  VQAF_REQUIRE(B > 0 && N > 0 && H > 0 && H <= 16, VQAF_E_SHAPE);
  const int HP = pad_heads(H);
  XA_BY_HP(softmax_heads_bwd_kernel, dim3(B), dim3(XA_THREADS), 0, (hipStream_t)stream, p, dp, N, H, ds);
  VQAF_LAUNCH_CHECK();
}

extern "C" int vqaf_weighted_token_sum(const void* seq, const float* c, int B, int N, int D, int H,
                                       int dtype, float* z, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(seq && c && z, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_dtype_ok(dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(B > 0 && N > 0 && D > 0 && H > 0 && H <= 16 && B <= 65535, VQAF_E_SHAPE);
  VQAF_REQUIRE(D % 8 == 0 && vqaf_aligned16(seq), VQAF_E_ALIGN);
  const int HP = pad_heads(H);
  const size_t smem = ((size_t)HP * N + (size_t)XA_THREADS * 8) * sizeof(float);
  VQAF_REQUIRE(smem <= 64 * 1024, VQAF_E_SHAPE);  // H*N coefficients must fit LDS
  const dim3 grid((D + XA_SLW - 1) / XA_SLW, B), block(XA_THREADS);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VQAF_BF16) {
    switch (HP) {
      case 1: hipLaunchKernelGGL((weighted_token_sum_kernel<true, 1>), grid, block, smem, st, seq, c, N, D, H, z); break;
      case 2: hipLaunchKernelGGL((weighted_token_sum_kernel<true, 2>), grid, block, smem, st, seq, c, N, D, H, z); break;
      case 4: hipLaunchKernelGGL((weighted_token_sum_kernel<true, 4>), grid, block, smem, st, seq, c, N, D, H, z); break;
      case 8: hipLaunchKernelGGL((weighted_token_sum_kernel<true, 8>), grid, block, smem, st, seq, c, N, D, H, z); break;
      default: hipLaunchKernelGGL((weighted_token_sum_kernel<true, 16>), grid, block, smem, st, seq, c, N, D, H, z); break;
    }
  } else {
    switch (HP) {
      case 1: hipLaunchKernelGGL((weighted_token_sum_kernel<false, 1>), grid, block, smem, st, seq, c, N, D, H, z); break;
      case 2: hipLaunchKernelGGL((weighted_token_sum_kernel<false, 2>), grid, block, smem, st, seq, c, N, D, H, z); break;
      case 4: hipLaunchKernelGGL((weighted_token_sum_kernel<false, 4>), grid, block, smem, st, seq, c, N, D, H, z); break;
      case 8: hipLaunchKernelGGL((weighted_token_sum_kernel<false, 8>), grid, block, smem, st, seq, c, N, D, H, z); break;
      default: hipLaunchKernelGGL((weighted_token_sum_kernel<false, 16>), grid, block, smem, st, seq, c, N, D, H, z); break;
    }
  }
  VQAF_LAUNCH_CHECK();
}

extern "C" int vqaf_token_outer(const float* ca, const float* va, const float* cb, const float* vb,
                                int B, int N, int D, int H, int dtype, void* out,
                                vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(ca && va && cb && vb && out, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_dtype_ok(dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(B > 0 && N > 0 && D > 0 && H > 0 && H <= 8 && B <= 65535, VQAF_E_SHAPE);
  VQAF_REQUIRE(D % 8 == 0 && vqaf_aligned16(out) && vqaf_aligned16(va) && vqaf_aligned16(vb), VQAF_E_ALIGN);
  const int HP = pad_heads(H);
  const size_t smem = (size_t)2 * HP * N * sizeof(float);
  VQAF_REQUIRE(smem <= 64 * 1024, VQAF_E_SHAPE);
  const dim3 grid((D + XA_SLW - 1) / XA_SLW, B), block(XA_THREADS);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VQAF_BF16) {
    switch (HP) {
      case 1: hipLaunchKernelGGL((token_outer_kernel<true, 1>), grid, block, smem, st, ca, va, cb, vb, N, D, H, out); break;
      case 2: hipLaunchKernelGGL((token_outer_kernel<true, 2>), grid, block, smem, st, ca, va, cb, vb, N, D, H, out); break;
      case 4: hipLaunchKernelGGL((token_outer_kernel<true, 4>), grid, block, smem, st, ca, va, cb, vb, N, D, H, out); break;
      default: hipLaunchKernelGGL((token_outer_kernel<true, 8>), grid, block, smem, st, ca, va, cb, vb, N, D, H, out); break;
    }
  } else {
    switch (HP) {
      case 1: hipLaunchKernelGGL((token_outer_kernel<false, 1>), grid, block, smem, st, ca, va, cb, vb, N, D, H, out); break;
      case 2: hipLaunchKernelGGL((token_outer_kernel<false, 2>), grid, block, smem, st, ca, va, cb, vb, N, D, H, out); break;
      case 4: hipLaunchKernelGGL((token_outer_kernel<false, 4>), grid, block, smem, st, ca, va, cb, vb, N, D, H, out); break;
      default: hipLaunchKernelGGL((token_outer_kernel<false, 8>), grid, block, smem, st, ca, va, cb, vb, N, D, H, out); break;
    }
  }
  VQAF_LAUNCH_CHECK();
}

#define XA_BY_DT_HP(KERNEL, ...)                                                                      \
  do {                                                                                                \
    if (dtype == VQAF_BF16) {                                                                         \
      switch (HP) {                                                                                   \
        case 1: XA_LAUNCH_ONE((KERNEL<true, 1 XA_EXTRA>), __VA_ARGS__); break;                        \
        case 2: XA_LAUNCH_ONE((KERNEL<true, 2 XA_EXTRA>), __VA_ARGS__); break;                        \
        case 4: XA_LAUNCH_ONE((KERNEL<true, 4 XA_EXTRA>), __VA_ARGS__); break;                        \
        default: XA_LAUNCH_ONE((KERNEL<true, 8 XA_EXTRA>), __VA_ARGS__); break;                       \
      }                                                                                               \
    } else {                                                                                          \
      switch (HP) {                                                                                   \
        case 1: XA_LAUNCH_ONE((KERNEL<false, 1 XA_EXTRA>), __VA_ARGS__); break;                       \
        case 2: XA_LAUNCH_ONE((KERNEL<false, 2 XA_EXTRA>), __VA_ARGS__); break;                       \
        case 4: XA_LAUNCH_ONE((KERNEL<false, 4 XA_EXTRA>), __VA_ARGS__); break;                       \
        default: XA_LAUNCH_ONE((KERNEL<false, 8 XA_EXTRA>), __VA_ARGS__); break;                      \
      }                                                                                               \
    }                                                                                                 \
  } while (0)
// (a kernel whose LDS request exceeds 64 KiB needs the attribute; setting it every time is a host-side no-op)
#define XA_LAUNCH_ONE(K, grid, block, smem, st, ...)                                                  \
  do {                                                                                                \
    if (smem > 48 * 1024 &&                                                                           \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, \
                            (int)smem) != hipSuccess)                                                 \
      return VQAF_E_LAUNCH;                                                                           \
    hipLaunchKernelGGL(K, grid, block, smem, st, __VA_ARGS__);                                        \
  } while (0)

constexpr size_t XA_MAX_LDS = 128 * 1024;

extern "C" int vqaf_softmax_wsum_multi(const vqaf_dense_fwd_mod* mods_host, int n, int B, int D, int H, int dtype,
                                       float* zero_ptr, size_t zero_count, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(mods_host != nullptr && (zero_count == 0 || zero_ptr != nullptr), VQAF_E_NULL);
  VQAF_REQUIRE(zero_count < (1ull << 32), VQAF_E_SHAPE);
  VQAF_REQUIRE(vqaf_dtype_ok(dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(n > 0 && n <= XA_MAX_MODS && B > 0 && B <= 65535 && D > 0 && H > 0 && H <= 8, VQAF_E_SHAPE);
  VQAF_REQUIRE(D % 8 == 0, VQAF_E_ALIGN);
  const int HP = pad_heads(H);
  XaFwdMods mods = {};
  int n_max = 0;
  for (int i = 0; i < n; ++i) {
    const vqaf_dense_fwd_mod& h = mods_host[i];
    VQAF_REQUIRE(h.seq && h.s && h.p && h.z, VQAF_E_NULL);
    VQAF_REQUIRE(h.N > 0, VQAF_E_SHAPE);
    VQAF_REQUIRE(vqaf_aligned16(h.seq), VQAF_E_ALIGN);
    mods.m[i] = XaFwdMod{h.seq, h.s, h.p, h.z, h.N};
    n_max = h.N > n_max ? h.N : n_max;
  }
  const size_t smem = (std::max((size_t)(HP + 4) * n_max, (size_t)XA_THREADS * 8) + 6 * HP) * sizeof(float);
  VQAF_REQUIRE(smem <= XA_MAX_LDS, VQAF_E_SHAPE);
  const dim3 grid((D + XA_SLW - 1) / XA_SLW, B, n), block(XA_THREADS);
  hipStream_t st = (hipStream_t)stream;
#define XA_EXTRA
  XA_BY_DT_HP(softmax_wsum_multi_kernel, grid, block, smem, st, mods, D, H, zero_ptr, (unsigned int)zero_count);
#undef XA_EXTRA
  VQAF_LAUNCH_CHECK();
}

extern "C" int vqaf_token_bwd_multi(const vqaf_dense_bwd_mod* mods_host, int n, int B, int D, int H, int dtype,
                                    vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(mods_host != nullptr, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_dtype_ok(dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(n > 0 && n <= XA_MAX_MODS && B > 0 && B <= 65535 && D > 0 && H > 0 && H <= 8, VQAF_E_SHAPE);
  VQAF_REQUIRE(D % 8 == 0, VQAF_E_ALIGN);
  const int HP = pad_heads(H);
  XaBwdMods mods = {};
  int n_max = 0;
  const bool need_u = mods_host[0].du != nullptr, need_seq = mods_host[0].grad_seq != nullptr;
  VQAF_REQUIRE(need_u || need_seq, VQAF_E_NULL);
  for (int i = 0; i < n; ++i) {
    const vqaf_dense_bwd_mod& h = mods_host[i];
    VQAF_REQUIRE(h.seq && h.p && h.dp && h.u && h.t, VQAF_E_NULL);
    // one instantiation serves the launch: every problem asks for the same outputs
    VQAF_REQUIRE((h.du != nullptr) == need_u && (h.grad_seq != nullptr) == need_seq, VQAF_E_NULL);
    VQAF_REQUIRE(h.N > 0, VQAF_E_SHAPE);
    VQAF_REQUIRE(vqaf_aligned16(h.seq) && vqaf_aligned16(h.u) && vqaf_aligned16(h.t) &&
                     (h.grad_seq == nullptr || vqaf_aligned16(h.grad_seq)), VQAF_E_ALIGN);
    mods.m[i] = XaBwdMod{h.seq, h.p, h.dp, h.u, h.t, h.du, h.grad_seq, h.N};
    n_max = h.N > n_max ? h.N : n_max;
  }
  const size_t smem = (std::max((size_t)(2 * HP + 4) * n_max, (size_t)XA_THREADS * 8) + 5 * HP) * sizeof(float);
  VQAF_REQUIRE(smem <= XA_MAX_LDS, VQAF_E_SHAPE);
  const dim3 grid((D + XA_SLW - 1) / XA_SLW, B, n), block(XA_THREADS);
  hipStream_t st = (hipStream_t)stream;
  if (need_u && need_seq) {
#define XA_EXTRA , true, true
    XA_BY_DT_HP(token_bwd_multi_kernel, grid, block, smem, st, mods, D, H);
#undef XA_EXTRA
  } else if (need_seq) {
#define XA_EXTRA , false, true
    XA_BY_DT_HP(token_bwd_multi_kernel, grid, block, smem, st, mods, D, H);
#undef XA_EXTRA
  } else {
#define XA_EXTRA , true, false
    XA_BY_DT_HP(token_bwd_multi_kernel, grid, block, smem, st, mods, D, H);
#undef XA_EXTRA
  }
  VQAF_LAUNCH_CHECK();
}

#ifdef XA_STAMPS
extern "C" int vqaf_debug_xa_stamps(unsigned long long* host_out, size_t count) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(xa_stamp_buf), count * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
