// Small strided-batched f32 GEMM on the f32 matrix cores (LDS-tiled).  It serves the per-head
// projections that surround the token kernels (query folding, value/out projections of the
// dense branch): a few hundred MFLOP at most, with arbitrary operand strides.  The big
// batch-contraction GEMMs of the DP-SGD step live in dp.hip.
#include "common.h"

namespace {

constexpr int GT = 64;   // output tile edge (four waves of 32 x 32)
constexpr int GK = 32;   // k rows per stage
constexpr int GP = GT + 1;  // LDS row pitch (floats): spreads the transposing stores over banks
constexpr int GTHREADS = 256;
constexpr int GVECS = GT * GK / 4 / GTHREADS;  // 4-element groups per thread per operand

enum { G_KVEC = 0, G_XVEC = 1, G_SCALAR = 2 };  // contiguous along k / along m|n / neither

__device__ __forceinline__ float ld1(const void* p, bool bf16, long off) {
  return bf16 ? load1<true>(p, (size_t)off) : load1<false>(p, (size_t)off);
}
// four consecutive elements starting at element offset `off` (a multiple of 4)
__device__ __forceinline__ void ld4(const void* p, bool bf16, long off, float (&v)[4]) {
  if (bf16) {
    const u32x2 r = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(p) + off);
    v[0] = bf16_lo(r.x); v[1] = bf16_hi(r.x); v[2] = bf16_lo(r.y); v[3] = bf16_hi(r.y);
  } else {
    const f32x4 r = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p) + off);
    v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
  }
}

// Loads this thread's share of a GT x GK operand tile.  x is the m (or n) index, k the
// contraction index; element (x, k) lives at base + x*sx + k*sk.
template <int MODE>
__device__ __forceinline__ void fetch_tile(const void* p, bool bf16, long base, long sx, long sk,
                                           int x0, int X, int k0, int K, int tid,
                                           float (&r)[GVECS][4]) {
#pragma unroll
  for (int i = 0; i < GVECS; ++i) {
    const int v = tid + i * GTHREADS;
    int x, k;
    if (MODE == G_KVEC) { x = v / (GK / 4); k = (v % (GK / 4)) * 4; }
    else if (MODE == G_XVEC) { x = (v % (GT / 4)) * 4; k = v / (GT / 4); }
    else { x = v % GT; k = (v / GT) * 4; }
    const int gx = x0 + x, gk = k0 + k;
    if (MODE == G_KVEC) {
      if (gx < X && gk + 3 < K) ld4(p, bf16, base + gx * sx + gk, r[i]);
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) r[i][e] = (gx < X && gk + e < K) ? ld1(p, bf16, base + gx * sx + gk + e) : 0.f;
      }
    } else if (MODE == G_XVEC) {
      if (gx + 3 < X && gk < K) ld4(p, bf16, base + gx + gk * sk, r[i]);
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) r[i][e] = (gx + e < X && gk < K) ? ld1(p, bf16, base + gx + e + gk * sk) : 0.f;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        r[i][e] = (gx < X && gk + e < K) ? ld1(p, bf16, base + gx * sx + (gk + e) * sk) : 0.f;
    }
  }
}
template <int MODE>
__device__ __forceinline__ void store_tile(float (*T)[GP], int tid, const float (&r)[GVECS][4]) {
#pragma unroll
  for (int i = 0; i < GVECS; ++i) {
    const int v = tid + i * GTHREADS;
    if (MODE == G_XVEC) {
      const int x = (v % (GT / 4)) * 4, k = v / (GT / 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) T[k][x + e] = r[i][e];
    } else {
      int x, k;
      if (MODE == G_KVEC) { x = v / (GK / 4); k = (v % (GK / 4)) * 4; }
      else { x = v % GT; k = (v / GT) * 4; }
#pragma unroll
      for (int e = 0; e < 4; ++e) T[k + e][x] = r[i][e];
    }
  }
}

// Operands are staged k-major into LDS (As[k][m], Bs[k][n]) whatever their strides -- with
// 16-byte (f32) / 8-byte (bf16) loads along whichever index is contiguous -- one stage ahead
// through registers, then each wave feeds v_mfma_f32_32x32x2_f32 (exact f32, k-ordered fmaf
// chain) one k pair at a time: lane l supplies A[m = l & 31][k = l >> 5] and
// B[k = l >> 5][n = l & 31].
template <int AMODE, int BMODE>
__global__ void __launch_bounds__(GTHREADS)
gemm_f32_kernel(const void* __restrict__ A, int a_bf16, long a_sg, long a_sm, long a_sk,
                const void* __restrict__ Bm, int b_bf16, long b_sg, long b_sk, long b_sn,
                const void* __restrict__ bias, int bias_bf16, long bias_sg, float* __restrict__ C,
                long c_sg, long c_sm, int M, int N, int K, float alpha, int beta_one) {
  __shared__ float As[2][GK][GP];
  __shared__ float Bs[2][GK][GP];
  const int g = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const long abase = (long)g * a_sg, bbase = (long)g * b_sg;
  float ra[GVECS][4], rb[GVECS][4];
  const int stages = (K + GK - 1) / GK;
  fetch_tile<AMODE>(A, a_bf16, abase, a_sm, a_sk, m0, M, 0, K, tid, ra);
  fetch_tile<BMODE>(Bm, b_bf16, bbase, b_sn, b_sk, n0, N, 0, K, tid, rb);
  store_tile<AMODE>(As[0], tid, ra);
  store_tile<BMODE>(Bs[0], tid, rb);
  __syncthreads();
  for (int s = 0; s < stages; ++s) {
    const int buf = s & 1;
    if (s + 1 < stages) {
      fetch_tile<AMODE>(A, a_bf16, abase, a_sm, a_sk, m0, M, (s + 1) * GK, K, tid, ra);
      fetch_tile<BMODE>(Bm, b_bf16, bbase, b_sn, b_sk, n0, N, (s + 1) * GK, K, tid, rb);
    }
#pragma unroll
    for (int kk = 0; kk < GK; kk += 2) {
      const float af = As[buf][kk + (lane >> 5)][wm * 32 + (lane & 31)];
      const float bf = Bs[buf][kk + (lane >> 5)][wn * 32 + (lane & 31)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc, 0, 0, 0);
    }
    if (s + 1 < stages) {
      store_tile<AMODE>(As[buf ^ 1], tid, ra);
      store_tile<BMODE>(Bs[buf ^ 1], tid, rb);
    }
    __syncthreads();
  }
  // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int gm = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const int gn = n0 + wn * 32 + (lane & 31);
    if (gm < M && gn < N) {
      float v = alpha * acc[r];
      if (bias != nullptr) v += ld1(bias, bias_bf16 != 0, (long)g * bias_sg + gn);
      float* c = C + (long)g * c_sg + gm * c_sm + gn;
      if (beta_one) v += *c;
      *c = v;
    }
  }
}

// Vector loads need every 4-element group to start on an 8-byte (bf16) / 16-byte (f32) boundary.
int pick_mode(const void* p, int dtype, long sg, long sx, long sk) {
  const size_t es = vqaf_esize(dtype);
  const bool base_ok = (reinterpret_cast<uintptr_t>(p) % (4 * es)) == 0 && sg % 4 == 0;
  if (sk == 1 && base_ok && sx % 4 == 0) return G_KVEC;
  if (sx == 1 && base_ok && sk % 4 == 0) return G_XVEC;
  return G_SCALAR;
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
  const int ab = a_dtype == VQAF_BF16, bb = b_dtype == VQAF_BF16;
  const int am = pick_mode(A, a_dtype, a_sg, a_sm, a_sk), bm = pick_mode(Bm, b_dtype, b_sg, b_sn, b_sk);
#define VQAF_GEMM_LAUNCH(AM, BM_)                                                                   \
  hipLaunchKernelGGL((gemm_f32_kernel<AM, BM_>), grid, block, 0, st, A, ab, a_sg, a_sm, a_sk, Bm, bb, \
                     b_sg, b_sk, b_sn, bias, bias_bf16, bias_sg, C, c_sg, c_sm, M, N, K, alpha, beta_one)
  switch (am * 3 + bm) {
    case 0: VQAF_GEMM_LAUNCH(G_KVEC, G_KVEC); break;
    case 1: VQAF_GEMM_LAUNCH(G_KVEC, G_XVEC); break;
    case 2: VQAF_GEMM_LAUNCH(G_KVEC, G_SCALAR); break;
    case 3: VQAF_GEMM_LAUNCH(G_XVEC, G_KVEC); break;
    case 4: VQAF_GEMM_LAUNCH(G_XVEC, G_XVEC); break;
    case 5: VQAF_GEMM_LAUNCH(G_XVEC, G_SCALAR); break;
    case 6: VQAF_GEMM_LAUNCH(G_SCALAR, G_KVEC); break;
    case 7: VQAF_GEMM_LAUNCH(G_SCALAR, G_XVEC); break;
    default: VQAF_GEMM_LAUNCH(G_SCALAR, G_SCALAR); break;
  }
#undef VQAF_GEMM_LAUNCH
  VQAF_LAUNCH_CHECK();
}
