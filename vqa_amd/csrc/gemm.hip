// Small strided-batched f32 GEMM on the vector ALU (LDS-tiled).  It serves the per-head
// projections that surround the token kernels (query folding, value/out projections of the
// dense branch): a few hundred MFLOP at most, with arbitrary operand strides.  The big
// batch-contraction GEMMs of the DP-SGD step live in dp.hip and use MFMA.
#include "common.h"

namespace {

constexpr int GT = 32;   // output tile edge
constexpr int GK = 32;   // k step
constexpr int GTHREADS = 256;

template <bool BF16>
__device__ __forceinline__ float ld(const void* p, long off) {
  return load1<BF16>(p, (size_t)off);
}

template <bool A_BF16, bool B_BF16>
__global__ void __launch_bounds__(GTHREADS)
gemm_f32_kernel(const void* __restrict__ A, long a_sg, long a_sm, long a_sk,
                const void* __restrict__ Bm, long b_sg, long b_sk, long b_sn,
                const void* __restrict__ bias, int bias_bf16, long bias_sg, float* __restrict__ C,
                long c_sg, long c_sm, int M, int N, int K, float alpha, int beta_one) {
  __shared__ float As[GK][GT + 1];
  __shared__ float Bs[GK][GT + 1];
  const int g = blockIdx.z, tid = threadIdx.x;
  const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
  const int tx = tid & 15, ty = tid >> 4;  // 16 x 16 threads, 2 x 2 outputs each
  float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  const long abase = (long)g * a_sg, bbase = (long)g * b_sg;
  for (int k0 = 0; k0 < K; k0 += GK) {
#pragma unroll
    for (int e = tid; e < GT * GK; e += GTHREADS) {
      int m, k;
      if (a_sk == 1) { k = e % GK; m = e / GK; } else { m = e % GT; k = e / GT; }
      const int gm = m0 + m, gk = k0 + k;
      As[k][m] = (gm < M && gk < K) ? ld<A_BF16>(A, abase + gm * a_sm + gk * a_sk) : 0.f;
      int n, kb;
      if (b_sn == 1) { n = e % GT; kb = e / GT; } else { kb = e % GK; n = e / GK; }
      const int gn = n0 + n, gkb = k0 + kb;
      Bs[kb][n] = (gn < N && gkb < K) ? ld<B_BF16>(Bm, bbase + gkb * b_sk + gn * b_sn) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < GK; ++k) {
      const float a0 = As[k][ty * 2], a1 = As[k][ty * 2 + 1];
      const float b0 = Bs[k][tx * 2], b1 = Bs[k][tx * 2 + 1];
      acc[0][0] = fmaf(a0, b0, acc[0][0]);
      acc[0][1] = fmaf(a0, b1, acc[0][1]);
      acc[1][0] = fmaf(a1, b0, acc[1][0]);
      acc[1][1] = fmaf(a1, b1, acc[1][1]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int gm = m0 + ty * 2 + i, gn = n0 + tx * 2 + j;
      if (gm < M && gn < N) {
        float v = alpha * acc[i][j];
        if (bias != nullptr)
          v += bias_bf16 ? load1<true>(bias, (size_t)(g * bias_sg + gn))
                         : load1<false>(bias, (size_t)(g * bias_sg + gn));
        float* c = C + (long)g * c_sg + gm * c_sm + gn;
        if (beta_one) v += *c;
        *c = v;
      }
    }
}

}  // namespace

extern "C" int vqaf_gemm_f32(const void* A, int a_dtype, long a_sg, long a_sm, long a_sk,
                             const void* Bm, int b_dtype, long b_sg, long b_sk, long b_sn,
                             const void* bias, int bias_dtype, long bias_sg, float* C, long c_sg,
                             long c_sm, int G, int M, int N, int K, float alpha, int beta_one,
                             vqaf_stream_t stream) {
  VQAF_REQUIRE(A != nullptr && Bm != nullptr && C != nullptr, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_dtype_ok(a_dtype) && vqaf_dtype_ok(b_dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(bias == nullptr || vqaf_dtype_ok(bias_dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(G > 0 && M > 0 && N > 0 && K > 0 && G <= 65535, VQAF_E_SHAPE);
  const dim3 grid((N + GT - 1) / GT, (M + GT - 1) / GT, G), block(GTHREADS);
  VQAF_REQUIRE(grid.y <= 65535, VQAF_E_SHAPE);
  hipStream_t st = (hipStream_t)stream;
  const int bias_bf16 = bias_dtype == VQAF_BF16;
#define VQAF_GEMM_LAUNCH(AB, BB)                                                                  \
  hipLaunchKernelGGL((gemm_f32_kernel<AB, BB>), grid, block, 0, st, A, a_sg, a_sm, a_sk, Bm, b_sg, \
                     b_sk, b_sn, bias, bias_bf16, bias_sg, C, c_sg, c_sm, M, N, K, alpha, beta_one)
  const bool ab = a_dtype == VQAF_BF16, bb = b_dtype == VQAF_BF16;
  if (ab && bb) VQAF_GEMM_LAUNCH(true, true);
  else if (ab) VQAF_GEMM_LAUNCH(true, false);
  else if (bb) VQAF_GEMM_LAUNCH(false, true);
  else VQAF_GEMM_LAUNCH(false, false);
#undef VQAF_GEMM_LAUNCH
  VQAF_LAUNCH_CHECK();
}
