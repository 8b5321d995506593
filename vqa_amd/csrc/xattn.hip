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
