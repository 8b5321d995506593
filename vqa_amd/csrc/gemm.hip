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

// Raw (unconverted) loads: the conversion to f32 happens when the registers are written to LDS,
// one or two stages later, so the compiler has no reason to wait for a load right after issuing it.
template <bool BF16>
__device__ __forceinline__ unsigned int ldraw1(const void* p, long off) {
  if constexpr (BF16) return (unsigned int)reinterpret_cast<const bf16_t*>(p)[off];
  else return reinterpret_cast<const unsigned int*>(p)[off];
}
template <bool BF16>
__device__ __forceinline__ u32x4 ldraw4(const void* p, long off) {  // off: multiple of 4 elements
  if constexpr (BF16) {
    const u32x2 r = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(p) + off);
    return u32x4{r.x, r.y, 0u, 0u};
  } else {
    return *reinterpret_cast<const u32x4*>(reinterpret_cast<const float*>(p) + off);
  }
}
template <bool BF16, bool VEC>
__device__ __forceinline__ void unpack4(const u32x4 r, float (&v)[4]) {
  if constexpr (!BF16) {
    v[0] = __uint_as_float(r.x); v[1] = __uint_as_float(r.y); v[2] = __uint_as_float(r.z); v[3] = __uint_as_float(r.w);
  } else if constexpr (VEC) {
    v[0] = bf16_lo(r.x); v[1] = bf16_hi(r.x); v[2] = bf16_lo(r.y); v[3] = bf16_hi(r.y);
  } else {
    v[0] = bf16_lo(r.x); v[1] = bf16_lo(r.y); v[2] = bf16_lo(r.z); v[3] = bf16_lo(r.w);
  }
}

template <int MODE>
__device__ __forceinline__ void tile_coords(int v, int& x, int& k) {
  if (MODE == G_KVEC) { x = v / (GK / 4); k = (v % (GK / 4)) * 4; }
  else if (MODE == G_XVEC) { x = (v % (GT / 4)) * 4; k = v / (GT / 4); }
  else { x = v % GT; k = (v / GT) * 4; }
}

// Loads this thread's share of a GT x GK operand tile.  x is the m (or n) index, k the
// contraction index; element (x, k) lives at base + x*sx + k*sk.  Every load is unconditional:
// indices outside the matrix are clamped to a valid element (the value is zeroed when the tile is
// written to LDS), so the loads of a stage issue back to back.  Vector modes are only selected
// when the vectorised extent is a multiple of 4: a group is entirely inside or outside.
template <int MODE, bool BF16>
__device__ __forceinline__ void fetch_tile(const void* p, long base, long sx, long sk, int x0, int X,
                                           int k0, int K, int tid, u32x4 (&r)[GVECS]) {
#pragma unroll
  for (int i = 0; i < GVECS; ++i) {
    int x, k;
    tile_coords<MODE>(tid + i * GTHREADS, x, k);
    const int gx = x0 + x, gk = k0 + k;
    if (MODE == G_KVEC) {
      r[i] = ldraw4<BF16>(p, base + (long)min(gx, X - 1) * sx + min(gk, K - 4));
    } else if (MODE == G_XVEC) {
      r[i] = ldraw4<BF16>(p, base + min(gx, X - 4) + (long)min(gk, K - 1) * sk);
    } else {
      const long rowb = base + (long)min(gx, X - 1) * sx;
      r[i] = u32x4{ldraw1<BF16>(p, rowb + (long)min(gk, K - 1) * sk),
                   ldraw1<BF16>(p, rowb + (long)min(gk + 1, K - 1) * sk),
                   ldraw1<BF16>(p, rowb + (long)min(gk + 2, K - 1) * sk),
                   ldraw1<BF16>(p, rowb + (long)min(gk + 3, K - 1) * sk)};
    }
  }
}
// Converts, zeroes what lies outside the matrix and writes the tile k-major into LDS.
template <int MODE, bool BF16>
__device__ __forceinline__ void store_tile(float (*T)[GP], int tid, const u32x4 (&r)[GVECS], int x0,
                                           int X, int k0, int K) {
#pragma unroll
  for (int i = 0; i < GVECS; ++i) {
    int x, k;
    tile_coords<MODE>(tid + i * GTHREADS, x, k);
    const int gx = x0 + x, gk = k0 + k;
    float v[4];
    unpack4<BF16, MODE != G_SCALAR>(r[i], v);
    if (MODE == G_XVEC) {
      const bool ok = gx < X && gk < K;
#pragma unroll
      for (int e = 0; e < 4; ++e) T[k][x + e] = ok ? v[e] : 0.f;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) T[k + e][x] = (gx < X && gk + e < K) ? v[e] : 0.f;
    }
  }
}

// Operands are staged k-major into LDS (As[k][m], Bs[k][n]) whatever their strides -- with
// 16-byte (f32) / 8-byte (bf16) loads along whichever index is contiguous -- two stages ahead
// through registers, then each wave feeds v_mfma_f32_32x32x2_f32 (exact f32, k-ordered fmaf
// chain) one k pair at a time: lane l supplies A[m = l & 31][k = l >> 5] and
// B[k = l >> 5][n = l & 31].
// Up to four independent problems of the same shape (different operand pointers) share one launch:
// blockIdx.z = set * gper + g.  Small GEMMs that would each fill half the chip then fill it together.
constexpr int GEMM_MAX_SETS = 4;
struct GemmSets {
  const void* A[GEMM_MAX_SETS];
  const void* Bm[GEMM_MAX_SETS];
  const void* bias[GEMM_MAX_SETS];
  float* C[GEMM_MAX_SETS];
  int gper;  // batches per set
};

template <int AMODE, int BMODE, bool A_BF16, bool B_BF16>
__global__ void __launch_bounds__(GTHREADS)
gemm_f32_kernel(GemmSets sets, long a_sg, long a_sm, long a_sk, long b_sg, long b_sk, long b_sn,
                int bias_bf16, long bias_sg, long c_sg, long c_sm, int M, int N, int K_total, float alpha,
                int beta_one, int splits, int k_per_split) {
  __shared__ float As[2][GK][GP];
  __shared__ float Bs[2][GK][GP];
  // blockIdx.z = (set * gper + g) * splits + kz: a split owns k in [kz * k_per_split, ...) and adds its
  // part into C atomically (the launcher has zeroed C unless it accumulates anyway)
  const int kz = blockIdx.z % splits;
  const int zb = blockIdx.z / splits;
  const int k_lo = kz * k_per_split;
  const int K = min(K_total, k_lo + k_per_split);  // this split's upper bound (loads clamp to it)
  const int set = zb / sets.gper;
  const void* __restrict__ A = sets.A[set];
  const void* __restrict__ Bm = sets.Bm[set];
  const void* __restrict__ bias = sets.bias[set];
  float* __restrict__ C = sets.C[set];
  const int g = zb % sets.gper, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  // a lane's sixteen outputs share one column: its bias is one value, fetched before the k loop
  const int gn = n0 + wn * 32 + (lane & 31);
  float bias_v = 0.f;
  if (bias != nullptr && kz == 0 && gn < N)
    bias_v = bias_bf16 ? load1<true>(bias, (size_t)((long)g * bias_sg + gn)) : load1<false>(bias, (size_t)((long)g * bias_sg + gn));
  const long abase = (long)g * a_sg, bbase = (long)g * b_sg;
  // register staging runs two stages ahead of the MFMAs (stage t lives in register set t & 1),
  // LDS is double buffered: the loads of stage s+2 are in flight while stage s is multiplied
  u32x4 ra[2][GVECS], rb[2][GVECS];
  const int stages = (K - k_lo + GK - 1) / GK;
  fetch_tile<AMODE, A_BF16>(A, abase, a_sm, a_sk, m0, M, k_lo, K, tid, ra[0]);
  fetch_tile<BMODE, B_BF16>(Bm, bbase, b_sn, b_sk, n0, N, k_lo, K, tid, rb[0]);
  fetch_tile<AMODE, A_BF16>(A, abase, a_sm, a_sk, m0, M, k_lo + GK, K, tid, ra[1]);
  fetch_tile<BMODE, B_BF16>(Bm, bbase, b_sn, b_sk, n0, N, k_lo + GK, K, tid, rb[1]);
  store_tile<AMODE, A_BF16>(As[0], tid, ra[0], m0, M, k_lo, K);
  store_tile<BMODE, B_BF16>(Bs[0], tid, rb[0], n0, N, k_lo, K);
  __syncthreads();
  // The loop body is branch-free (an odd stage count is padded with one all-zero stage: indices
  // are clamped on load and masked to zero on store), so every wait the compiler inserts is a
  // counted one and the loads of two stages really stay in flight.
  const int stages2 = (stages + 1) & ~1;
#pragma unroll 1
  for (int s = 0; s < stages2; s += 2) {
#pragma unroll
    for (int P = 0; P < 2; ++P) {
      const int t = s + P;
      fetch_tile<AMODE, A_BF16>(A, abase, a_sm, a_sk, m0, M, k_lo + (t + 2) * GK, K, tid, ra[P]);
      fetch_tile<BMODE, B_BF16>(Bm, bbase, b_sn, b_sk, n0, N, k_lo + (t + 2) * GK, K, tid, rb[P]);
      // all operand reads of the stage are issued before the first MFMA: one LDS latency per stage,
      // not one per k pair
      float af[GK / 2], bf[GK / 2];
#pragma unroll
      for (int i = 0; i < GK / 2; ++i) {
        af[i] = As[P][2 * i + (lane >> 5)][wm * 32 + (lane & 31)];
        bf[i] = Bs[P][2 * i + (lane >> 5)][wn * 32 + (lane & 31)];
      }
      __builtin_amdgcn_sched_barrier(0);  // keeps the scheduler from sinking the reads back to their uses
#pragma unroll
      for (int i = 0; i < GK / 2; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[i], acc, 0, 0, 0);
      store_tile<AMODE, A_BF16>(As[P ^ 1], tid, ra[P ^ 1], m0, M, k_lo + (t + 1) * GK, K);
      store_tile<BMODE, B_BF16>(Bs[P ^ 1], tid, rb[P ^ 1], n0, N, k_lo + (t + 1) * GK, K);
      __syncthreads();
    }
  }
  // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  float* const c0 = C + (long)g * c_sg + gn;
  const int gm0 = m0 + wm * 32 + 4 * (lane >> 5);
  float prev[16];
  if (beta_one && splits == 1) {  // the sixteen reads issue together
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int gm = gm0 + (r & 3) + 8 * (r >> 2);
      prev[r] = (gm < M && gn < N) ? c0[(long)gm * c_sm] : 0.f;
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int gm = gm0 + (r & 3) + 8 * (r >> 2);
    if (gm < M && gn < N) {
      const float v = alpha * acc[r] + bias_v;
      float* c = c0 + (long)gm * c_sm;
      if (splits > 1) atomicAdd(c, v);
      else *c = beta_one ? v + prev[r] : v;
    }
  }
}

// Vector loads need every 4-element group to start on an 8-byte (bf16) / 16-byte (f32) boundary.
int pick_mode(const void* p, int dtype, long sg, long sx, long sk, int X, int K) {
  const size_t es = vqaf_esize(dtype);
  const bool base_ok = (reinterpret_cast<uintptr_t>(p) % (4 * es)) == 0 && sg % 4 == 0;
  if (sk == 1 && base_ok && sx % 4 == 0 && K % 4 == 0) return G_KVEC;
  if (sx == 1 && base_ok && sk % 4 == 0 && X % 4 == 0) return G_XVEC;
  return G_SCALAR;
}

// Launches `nsets` problems of identical shape / strides / dtypes in one grid.
int gemm_launch(const GemmSets& sets, int nsets, int a_dtype, long a_sg, long a_sm, long a_sk, int b_dtype,
                long b_sg, long b_sk, long b_sn, int bias_dtype, long bias_sg, long c_sg, long c_sm, int G,
                int M, int N, int K, float alpha, int beta_one, hipStream_t st, bool allow_split = false) {
  VQAF_REQUIRE(nsets > 0 && nsets <= GEMM_MAX_SETS, VQAF_E_SHAPE);
  VQAF_REQUIRE(vqaf_dtype_ok(a_dtype) && vqaf_dtype_ok(b_dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(G > 0 && M > 0 && N > 0 && K > 0 && (long)G * nsets <= 65535, VQAF_E_SHAPE);
  int am = G_KVEC, bm = G_KVEC;
  bool first = true;
  for (int i = 0; i < nsets; ++i) {
    VQAF_REQUIRE(sets.A[i] != nullptr && sets.Bm[i] != nullptr && sets.C[i] != nullptr, VQAF_E_NULL);
    VQAF_REQUIRE(sets.bias[i] == nullptr || vqaf_dtype_ok(bias_dtype), VQAF_E_DTYPE);
    const int ai = pick_mode(sets.A[i], a_dtype, a_sg, a_sm, a_sk, M, K);
    const int bi = pick_mode(sets.Bm[i], b_dtype, b_sg, b_sn, b_sk, N, K);
    if (first) { am = ai; bm = bi; first = false; }
    if (ai != am) am = G_SCALAR;  // the sets must share one staging mode
    if (bi != bm) bm = G_SCALAR;
  }
  // Few output tiles and a long contraction (the (B,E)-sized projections at small batches): where the
  // caller allows it, split k over workgroups, which add their parts atomically (the sum then varies in
  // its last bits from run to run, so the scoring path, whose results rank tokens, never asks for it).  Needs C zeroed first, which is one memset when
  // the written elements form one dense block (plain row-major, or head slices that tile the rows).
  int splits = 1;
  const long tiles = (long)((N + GT - 1) / GT) * ((M + GT - 1) / GT) * G * nsets;
  const bool dense_c = (c_sm == N && (G == 1 || c_sg == (long)M * N)) || (c_sg == N && c_sm == (long)G * N);
  if (allow_split && tiles < 128 && K >= 4 * GK && (beta_one || dense_c)) {
    while (splits < 16 && tiles * splits < 256 && K / (splits * 2) >= 2 * GK) splits *= 2;
  }
  const int k_per_split = ((K + splits - 1) / splits + GK - 1) / GK * GK;
  if (splits > 1 && !beta_one) {
    for (int i = 0; i < nsets; ++i)
      if (hipMemsetAsync(sets.C[i], 0, (size_t)G * M * N * sizeof(float), st) != hipSuccess) return VQAF_E_LAUNCH;
  }
  const dim3 grid((N + GT - 1) / GT, (M + GT - 1) / GT, G * nsets * splits), block(GTHREADS);
  VQAF_REQUIRE(grid.y <= 65535 && grid.z <= 65535, VQAF_E_SHAPE);
  const int bias_bf16 = bias_dtype == VQAF_BF16;
  const bool ab = a_dtype == VQAF_BF16, bb = b_dtype == VQAF_BF16;
#define VQAF_GEMM_LAUNCH(AM, BM_, AB, BB)                                                             \
  hipLaunchKernelGGL((gemm_f32_kernel<AM, BM_, AB, BB>), grid, block, 0, st, sets, a_sg, a_sm, a_sk,   \
                     b_sg, b_sk, b_sn, bias_bf16, bias_sg, c_sg, c_sm, M, N, K, alpha, beta_one, splits, k_per_split)
#define VQAF_GEMM_BY_DT(AM, BM_)                            \
  do {                                                      \
    if (ab && bb) VQAF_GEMM_LAUNCH(AM, BM_, true, true);    \
    else if (ab) VQAF_GEMM_LAUNCH(AM, BM_, true, false);    \
    else if (bb) VQAF_GEMM_LAUNCH(AM, BM_, false, true);    \
    else VQAF_GEMM_LAUNCH(AM, BM_, false, false);           \
  } while (0)
  switch (am * 3 + bm) {
    case 0: VQAF_GEMM_BY_DT(G_KVEC, G_KVEC); break;
    case 1: VQAF_GEMM_BY_DT(G_KVEC, G_XVEC); break;
    case 2: VQAF_GEMM_BY_DT(G_KVEC, G_SCALAR); break;
    case 3: VQAF_GEMM_BY_DT(G_XVEC, G_KVEC); break;
    case 4: VQAF_GEMM_BY_DT(G_XVEC, G_XVEC); break;
    case 5: VQAF_GEMM_BY_DT(G_XVEC, G_SCALAR); break;
    case 6: VQAF_GEMM_BY_DT(G_SCALAR, G_KVEC); break;
    case 7: VQAF_GEMM_BY_DT(G_SCALAR, G_XVEC); break;
    default: VQAF_GEMM_BY_DT(G_SCALAR, G_SCALAR); break;
  }
#undef VQAF_GEMM_BY_DT
#undef VQAF_GEMM_LAUNCH
  VQAF_LAUNCH_CHECK();
}


// ---- Several DIFFERENT small products in one launch ------------------------------------------------------
// The backward of the dense co-attention has groups of short products that do not depend on each other (an
// input gradient and two weight gradients of one incoming gradient, ...).  Each is a few microseconds of
// load -> MFMA -> atomic add, and launches on one stream do not overlap, so they share a launch: every problem
// brings its own shape, strides, operand types and staging modes, a workgroup finds its problem from the
// running tile count, and the body is gemm_f32_kernel's with the staging mode chosen at run time (uniform per
// workgroup).  Outputs are ADDED into C (the caller provides it zero-filled); long contractions are split over
// workgroups with f32 atomics, as in vqaf_gemm_f32_atomic -- or, per problem, written with plain stores by
// workgroups that each take a whole contraction (the f32 clipped sums of the DP step: many tiles, nothing to clear).
constexpr int GEMM_MAX_GROUP = 12;
struct GemmProb {
  const void* A;
  const void* Bm;
  const void* bias;
  float* C;
  long a_sg, a_sm, a_sk, b_sg, b_sk, b_sn, bias_sg, c_sg, c_sm;
  int G, M, N, K;
  float alpha;
  int amode, bmode, a_bf16, b_bf16, bias_bf16;
  int splits, k_per_split, tiles_x, tiles_y;
  int wg_begin;  // first workgroup of this problem
  int store;     // C = result with plain stores (the contraction is then never split); else C += result (atomics)
  int c_bf16;    // (store only) C holds bf16: the product leaves in its consumer's type, no cast launch after it
};
struct GemmGroupArgs {
  GemmProb p[GEMM_MAX_GROUP];
  int n;
};

template <bool BF16>
__device__ __forceinline__ void fetch_tile_rt(int mode, const void* p, long base, long sx, long sk, int x0, int X,
                                              int k0, int K, int tid, u32x4 (&r)[GVECS]) {
  if (mode == G_KVEC) fetch_tile<G_KVEC, BF16>(p, base, sx, sk, x0, X, k0, K, tid, r);
  else if (mode == G_XVEC) fetch_tile<G_XVEC, BF16>(p, base, sx, sk, x0, X, k0, K, tid, r);
  else fetch_tile<G_SCALAR, BF16>(p, base, sx, sk, x0, X, k0, K, tid, r);
}
__device__ __forceinline__ void fetch_tile_any(int mode, int bf16, const void* p, long base, long sx, long sk, int x0,
                                               int X, int k0, int K, int tid, u32x4 (&r)[GVECS]) {
  if (bf16) fetch_tile_rt<true>(mode, p, base, sx, sk, x0, X, k0, K, tid, r);
  else fetch_tile_rt<false>(mode, p, base, sx, sk, x0, X, k0, K, tid, r);
}
template <bool BF16>
__device__ __forceinline__ void store_tile_rt(int mode, float (*T)[GP], int tid, const u32x4 (&r)[GVECS], int x0, int X,
                                              int k0, int K) {
  if (mode == G_KVEC) store_tile<G_KVEC, BF16>(T, tid, r, x0, X, k0, K);
  else if (mode == G_XVEC) store_tile<G_XVEC, BF16>(T, tid, r, x0, X, k0, K);
  else store_tile<G_SCALAR, BF16>(T, tid, r, x0, X, k0, K);
}
__device__ __forceinline__ void store_tile_any(int mode, int bf16, float (*T)[GP], int tid, const u32x4 (&r)[GVECS],
                                               int x0, int X, int k0, int K) {
  if (bf16) store_tile_rt<true>(mode, T, tid, r, x0, X, k0, K);
  else store_tile_rt<false>(mode, T, tid, r, x0, X, k0, K);
}

__global__ void __launch_bounds__(GTHREADS)
gemm_f32_group_kernel(GemmGroupArgs args) {
  __shared__ float As[2][GK][GP];
  __shared__ float Bs[2][GK][GP];
  int pi = 0;
#pragma unroll
  for (int i = 1; i < GEMM_MAX_GROUP; ++i)
    if (i < args.n && (int)blockIdx.x >= args.p[i].wg_begin) pi = i;
  const GemmProb& P = args.p[pi];
  int local = blockIdx.x - P.wg_begin;
  const int tx = local % P.tiles_x;
  local /= P.tiles_x;
  const int ty = local % P.tiles_y;
  const int z = local / P.tiles_y;
  const int kz = z % P.splits, g = z / P.splits;
  const int M = P.M, N = P.N;
  const int k_lo = kz * P.k_per_split;
  const int K = min(P.K, k_lo + P.k_per_split);
  const void* __restrict__ A = P.A;
  const void* __restrict__ Bm = P.Bm;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = ty * GT, n0 = tx * GT;
  const int amode = P.amode, bmode = P.bmode, abf = P.a_bf16, bbf = P.b_bf16;
  const long a_sm = P.a_sm, a_sk = P.a_sk, b_sn = P.b_sn, b_sk = P.b_sk;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int gn = n0 + wn * 32 + (lane & 31);
  float bias_v = 0.f;
  if (P.bias != nullptr && kz == 0 && gn < N)
    bias_v = P.bias_bf16 ? load1<true>(P.bias, (size_t)((long)g * P.bias_sg + gn))
                         : load1<false>(P.bias, (size_t)((long)g * P.bias_sg + gn));
  const long abase = (long)g * P.a_sg, bbase = (long)g * P.b_sg;
  u32x4 ra[2][GVECS], rb[2][GVECS];
  const int stages = (K - k_lo + GK - 1) / GK;
  fetch_tile_any(amode, abf, A, abase, a_sm, a_sk, m0, M, k_lo, K, tid, ra[0]);
  fetch_tile_any(bmode, bbf, Bm, bbase, b_sn, b_sk, n0, N, k_lo, K, tid, rb[0]);
  fetch_tile_any(amode, abf, A, abase, a_sm, a_sk, m0, M, k_lo + GK, K, tid, ra[1]);
  fetch_tile_any(bmode, bbf, Bm, bbase, b_sn, b_sk, n0, N, k_lo + GK, K, tid, rb[1]);
  store_tile_any(amode, abf, As[0], tid, ra[0], m0, M, k_lo, K);
  store_tile_any(bmode, bbf, Bs[0], tid, rb[0], n0, N, k_lo, K);
  __syncthreads();
  const int stages2 = (stages + 1) & ~1;
#pragma unroll 1
  for (int s = 0; s < stages2; s += 2) {
#pragma unroll
    for (int PP = 0; PP < 2; ++PP) {
      const int t = s + PP;
      fetch_tile_any(amode, abf, A, abase, a_sm, a_sk, m0, M, k_lo + (t + 2) * GK, K, tid, ra[PP]);
      fetch_tile_any(bmode, bbf, Bm, bbase, b_sn, b_sk, n0, N, k_lo + (t + 2) * GK, K, tid, rb[PP]);
      float af[GK / 2], bf[GK / 2];
#pragma unroll
      for (int i = 0; i < GK / 2; ++i) {
        af[i] = As[PP][2 * i + (lane >> 5)][wm * 32 + (lane & 31)];
        bf[i] = Bs[PP][2 * i + (lane >> 5)][wn * 32 + (lane & 31)];
      }
#pragma unroll
      for (int i = 0; i < GK / 2; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[i], acc, 0, 0, 0);
      store_tile_any(amode, abf, As[PP ^ 1], tid, ra[PP ^ 1], m0, M, k_lo + (t + 1) * GK, K);
      store_tile_any(bmode, bbf, Bs[PP ^ 1], tid, rb[PP ^ 1], n0, N, k_lo + (t + 1) * GK, K);
      __syncthreads();
    }
  }
  float* const c0 = P.C + (long)g * P.c_sg + gn;
  const int gm0 = m0 + wm * 32 + 4 * (lane >> 5);
  const float alpha = P.alpha;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int gm = gm0 + (r & 3) + 8 * (r >> 2);
    if (gm < M && gn < N) {
      const float v = alpha * acc[r] + bias_v;
      if (P.store) {
        if (P.c_bf16) reinterpret_cast<bf16_t*>(P.C)[(long)g * P.c_sg + gn + (long)gm * P.c_sm] = f32_to_bf16(v);
        else c0[(long)gm * P.c_sm] = v;
      } else {
        atomicAdd(c0 + (long)gm * P.c_sm, v);
      }
    }
  }
}

// ---- Query fold with bf16 operands (bf16 queries, bf16 in_proj, head width 64, D = 512) on the bf16
// matrix cores, one launch.  A product of two bf16 values is exact in f32, so
//   qproj = q Wq_h^T + bq_h            (one bf16 MFMA per 32 k, f32 accumulation)
// has f32 accuracy as it stands, and the second product keeps it by entering qproj as THREE bf16 terms
// (8 + 8 + 8 mantissa bits: the split is exact) against the exact bf16 rows of Wk:
//   u[b,h,:] = (qproj_hi + qproj_mid + qproj_lo)[b, h*64 : h*64+64] @ Wk[h*64 : h*64+64, :] / 8.
// A workgroup owns 32 queries of one head: its 64 rows of Wk arrive in LDS by LDS-DMA (one 1 KiB row
// per instruction) while the first product runs straight from global memory (both operands are
// k-contiguous there, which is the MFMA's own fragment layout); the 32 x 64 qproj tile crosses to the
// A-operand layout through LDS, and Wk is read back with the transposing LDS read.  Every sum has a
// fixed order: the result does not vary from run to run (the scores rank tokens).
constexpr int FQ_ROWS = 32, FQ_THREADS = 256, FQ_D = 512, FQ_DH = 64;
constexpr int FQ_WP = (FQ_D + 8) * 2;  // bytes per staged Wk row (the pad spreads the transposing reads over banks)
constexpr int FQ_PP = (FQ_DH + 8) * 2;  // bytes per row of a staged qproj term
constexpr int FQ_PLANE = FQ_ROWS * FQ_PP;
constexpr size_t FQ_LDS = (size_t)FQ_DH * FQ_WP + 3 * (size_t)FQ_PLANE;

typedef __bf16 fq_frag __attribute__((ext_vector_type(8)));
typedef short fq_s16x4 __attribute__((ext_vector_type(4)));

struct FoldSets {
  const bf16_t* q[GEMM_MAX_SETS];
  const bf16_t* w[GEMM_MAX_SETS];
  const bf16_t* bias[GEMM_MAX_SETS];
  float* qproj[GEMM_MAX_SETS];
  float* u[GEMM_MAX_SETS];
};

__device__ __forceinline__ fq_frag fq_as_frag(u32x4 v) {
  union { u32x4 u; fq_frag f; } c;
  c.u = v;
  return c.f;
}
// Lane i of a 16-lane group points at row i / 4, columns 4 * (i % 4) .. + 3 of a 4 x 16 block of 16-bit
// elements and receives column i of the four rows.
__device__ __forceinline__ u32x2 fq_read_tr(const unsigned char* p) {
  const fq_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) fq_s16x4*)(const_cast<unsigned char*>(p)));
  union { fq_s16x4 s; u32x2 u; } c;
  c.s = v;
  return c.u;
}
// x = hi + mid + lo exactly, each term a bf16 (truncating splits of the 24-bit mantissa); writes the
// three terms to the same element of the three planes.
__device__ __forceinline__ void fq_split_store(float x, unsigned char* plane0_elem) {
  const unsigned int hi = __float_as_uint(x) & 0xffff0000u;
  const float r1 = x - __uint_as_float(hi);
  const unsigned int mid = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(mid);
  *reinterpret_cast<bf16_t*>(plane0_elem) = (bf16_t)(hi >> 16);
  *reinterpret_cast<bf16_t*>(plane0_elem + FQ_PLANE) = (bf16_t)(mid >> 16);
  *reinterpret_cast<bf16_t*>(plane0_elem + 2 * FQ_PLANE) = (bf16_t)(__float_as_uint(r2) >> 16);
}

__global__ void __launch_bounds__(FQ_THREADS)
fold_query_bf16_kernel(FoldSets sets, int B, int H, unsigned long long* count_state, unsigned long long delta) {
  // a {seed, counter} pair advanced on the side (nothing in this launch reads it)
  if (count_state != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0)
    count_state[1] = count_state[1] + delta;

  constexpr int D = FQ_D;
  extern __shared__ __attribute__((aligned(16))) unsigned char fq_lds[];
  unsigned char* Ws = fq_lds;                            // [64][FQ_WP] bf16 rows of Wk_h
  unsigned char* Ps = fq_lds + (size_t)FQ_DH * FQ_WP;    // [3][32][FQ_PP] the qproj tile's hi / mid / lo terms
  const int set = blockIdx.z, h = blockIdx.y, r0 = blockIdx.x * FQ_ROWS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
  const bf16_t* __restrict__ q = sets.q[set];
  const bf16_t* __restrict__ w = sets.w[set];
  const int col = 16 * wave + li;  // this lane's column of the head in the first product
  const unsigned int bias_raw = sets.bias[set][h * FQ_DH + col];
  // Wk rows of this head -> LDS; wave takes rows 16 * wave .. + 15
  const bf16_t* wk = w + (size_t)(D + h * FQ_DH) * D;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int k = 16 * wave + r;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wk + (size_t)k * D + 8 * lane),
                                     (__attribute__((address_space(3))) void*)(Ws + (size_t)k * FQ_WP), 16, 0, 0);
  }
  // first product: wave owns columns 16 * wave .. + 15 of the head, both 16-row tiles
  f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, p1 = {0.f, 0.f, 0.f, 0.f};
  {
    const bf16_t* a0 = q + (size_t)min(r0 + li, B - 1) * D + 8 * g;
    const bf16_t* a1 = q + (size_t)min(r0 + 16 + li, B - 1) * D + 8 * g;
    const bf16_t* bp = w + (size_t)(h * FQ_DH + col) * D + 8 * g;
    // every fragment of the contraction is requested before the first product: one memory latency for
    // the phase (the workgroup runs two waves per SIMD, so the registers are there)
    fq_frag fa0[D / 32], fa1[D / 32], fb[D / 32];
#pragma unroll
    for (int s = 0; s < D / 32; ++s) {
      fb[s] = *reinterpret_cast<const fq_frag*>(bp + 32 * s);
      fa0[s] = *reinterpret_cast<const fq_frag*>(a0 + 32 * s);
      fa1[s] = *reinterpret_cast<const fq_frag*>(a1 + 32 * s);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < D / 32; ++s) {
      p0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa0[s], fb[s], p0, 0, 0, 0);
      p1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa1[s], fb[s], p1, 0, 0, 0);
    }
    // accumulator layout: column = lane & 15, rows 4 * (lane >> 4) + r
    const float bv = bf16_lo(bias_raw);
    float* qp = sets.qproj[set];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v0 = p0[r] + bv, v1 = p1[r] + bv;
      const int row = 4 * g + r;
      fq_split_store(v0, Ps + row * FQ_PP + 2 * col);
      fq_split_store(v1, Ps + (16 + row) * FQ_PP + 2 * col);
      if (r0 + row < B) qp[(size_t)(r0 + row) * D + h * FQ_DH + col] = v0;
      if (r0 + 16 + row < B) qp[(size_t)(r0 + 16 + row) * D + h * FQ_DH + col] = v1;
    }
  }
  __syncthreads();  // also retires the LDS-DMA rows
  // second product: A = the qproj tile's three terms, [term][row tile][k step]
  fq_frag af[3][2][2];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int s = 0; s < 2; ++s)
        af[p][t][s] = *reinterpret_cast<const fq_frag*>(Ps + p * FQ_PLANE + (16 * t + li) * FQ_PP + (32 * s + 8 * g) * 2);
  // wave owns output columns 128 * wave .. + 127 in eight tiles of 16
  float* u = sets.u[set];
  const float scale = 0.125f;  // 1 / sqrt(64)
  const bool whole = r0 + FQ_ROWS <= B;  // uniform: the common case stores without per-row tests
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int colbase = 128 * wave + 16 * c;
    f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const unsigned char* rd = Ws + (size_t)(32 * s + 8 * g + (li >> 2)) * FQ_WP + (colbase + 4 * (li & 3)) * 2;
      const u32x2 b_lo = fq_read_tr(rd), b_hi = fq_read_tr(rd + 4 * FQ_WP);
      const fq_frag fb = fq_as_frag(u32x4{b_lo.x, b_lo.y, b_hi.x, b_hi.y});
#pragma unroll
      // The two operands of this MFMA have the same register layout (row or column = lane & 15, k = 8 (lane >> 4)
      // + e), so passing them in the other order yields the TRANSPOSED tile with the same sums: a lane ends with
      // four consecutive output columns (4 g + r) of ONE query (lane & 15), which leave as one 16-byte store
      // instead of four 4-byte ones (the 33 MB of u were this kernel's 147 k store instructions).
      for (int p = 2; p >= 0; --p) {  // smallest term first
        o0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, af[p][0][s], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, af[p][1][s], o1, 0, 0, 0);
      }
    }
    float* ur = u + ((size_t)(r0 + li) * H + h) * D + colbase + 4 * g;
    const f32x4 v0 = {o0[0] * scale, o0[1] * scale, o0[2] * scale, o0[3] * scale};
    const f32x4 v1 = {o1[0] * scale, o1[1] * scale, o1[2] * scale, o1[3] * scale};
    if (whole || r0 + li < B) *reinterpret_cast<f32x4*>(ur) = v0;
    if (whole || r0 + 16 + li < B) *reinterpret_cast<f32x4*>(ur + (size_t)16 * H * D) = v1;
  }
}

bool fold_bf16_applies(int D, int H, int dtype, int w_dtype) {
  return dtype == VQAF_BF16 && w_dtype == VQAF_BF16 && D == FQ_D && H * FQ_DH == D;
}

int fold_bf16_launch(const FoldSets& sets, int n, int B, int H, hipStream_t st, uint64_t* count_state = nullptr,
                     uint64_t delta = 0) {
  static int attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(fold_query_bf16_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)FQ_LDS);
  if (attr_rc != hipSuccess) return VQAF_E_LAUNCH;
  const dim3 grid((B + FQ_ROWS - 1) / FQ_ROWS, H, n), block(FQ_THREADS);
  hipLaunchKernelGGL(fold_query_bf16_kernel, grid, block, FQ_LDS, st, sets, B, H,
                     reinterpret_cast<unsigned long long*>(count_state), (unsigned long long)delta);
  VQAF_LAUNCH_CHECK();
}

}  // namespace

extern "C" int vqaf_gemm_f32(const void* A, int a_dtype, long a_sg, long a_sm, long a_sk,
                             const void* Bm, int b_dtype, long b_sg, long b_sk, long b_sn,
                             const void* bias, int bias_dtype, long bias_sg, float* C, long c_sg,
                             long c_sm, int G, int M, int N, int K, float alpha, int beta_one,
                             vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(A != nullptr && Bm != nullptr && C != nullptr, VQAF_E_NULL);
  GemmSets sets = {};
  sets.A[0] = A;
  sets.Bm[0] = Bm;
  sets.bias[0] = bias;
  sets.C[0] = C;
  sets.gper = G;
  return gemm_launch(sets, 1, a_dtype, a_sg, a_sm, a_sk, b_dtype, b_sg, b_sk, b_sn, bias_dtype, bias_sg, c_sg,
                     c_sm, G, M, N, K, alpha, beta_one, (hipStream_t)stream);
}

extern "C" int vqaf_gemm_f32_atomic(const void* A, int a_dtype, long a_sg, long a_sm, long a_sk,
                                    const void* Bm, int b_dtype, long b_sg, long b_sk, long b_sn,
                                    const void* bias, int bias_dtype, long bias_sg, float* C, long c_sg,
                                    long c_sm, int G, int M, int N, int K, float alpha, int beta_one,
                                    vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(A != nullptr && Bm != nullptr && C != nullptr, VQAF_E_NULL);
  GemmSets sets = {};
  sets.A[0] = A;
  sets.Bm[0] = Bm;
  sets.bias[0] = bias;
  sets.C[0] = C;
  sets.gper = G;
  return gemm_launch(sets, 1, a_dtype, a_sg, a_sm, a_sk, b_dtype, b_sg, b_sk, b_sn, bias_dtype, bias_sg, c_sg,
                     c_sm, G, M, N, K, alpha, beta_one, (hipStream_t)stream, true);
}

extern "C" int vqaf_gemm_f32_atomic_sets(int n_sets, const void* const* A_host, int a_dtype, long a_sg, long a_sm,
                                         long a_sk, const void* const* Bm_host, int b_dtype, long b_sg, long b_sk,
                                         long b_sn, const void* const* bias_host, int bias_dtype, long bias_sg,
                                         float* const* C_host, long c_sg, long c_sm, int G, int M, int N, int K,
                                         float alpha, int beta_one, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(A_host != nullptr && Bm_host != nullptr && C_host != nullptr, VQAF_E_NULL);
  VQAF_REQUIRE(n_sets > 0 && n_sets <= GEMM_MAX_SETS, VQAF_E_SHAPE);
  GemmSets sets = {};
  for (int i = 0; i < n_sets; ++i) {
    sets.A[i] = A_host[i];
    sets.Bm[i] = Bm_host[i];
    sets.bias[i] = bias_host != nullptr ? bias_host[i] : nullptr;
    sets.C[i] = C_host[i];
  }
  sets.gper = G;
  return gemm_launch(sets, n_sets, a_dtype, a_sg, a_sm, a_sk, b_dtype, b_sg, b_sk, b_sn, bias_dtype, bias_sg, c_sg,
                     c_sm, G, M, N, K, alpha, beta_one, (hipStream_t)stream, true);
}

extern "C" int vqaf_gemm_f32_group(int n, const vqaf_gemm_desc* probs_host, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(probs_host != nullptr, VQAF_E_NULL);
  VQAF_REQUIRE(n > 0 && n <= GEMM_MAX_GROUP, VQAF_E_SHAPE);
  GemmGroupArgs args = {};
  args.n = n;
  long total_tiles = 0;
  for (int i = 0; i < n; ++i) {
    const vqaf_gemm_desc& d = probs_host[i];
    total_tiles += (long)((d.N + GT - 1) / GT) * ((d.M + GT - 1) / GT) * d.G;
  }
  int wg = 0;
  for (int i = 0; i < n; ++i) {
    const vqaf_gemm_desc& d = probs_host[i];
    VQAF_REQUIRE(d.A != nullptr && d.Bm != nullptr && d.C != nullptr, VQAF_E_NULL);
    VQAF_REQUIRE(vqaf_dtype_ok(d.a_dtype) && vqaf_dtype_ok(d.b_dtype) && (d.bias == nullptr || vqaf_dtype_ok(d.bias_dtype)),
                 VQAF_E_DTYPE);
    VQAF_REQUIRE(d.G > 0 && d.M > 0 && d.N > 0 && d.K > 0, VQAF_E_SHAPE);
    // sums built with atomic adds are f32; a stored product may leave as bf16
    VQAF_REQUIRE(d.c_dtype == VQAF_F32 || (d.c_dtype == VQAF_BF16 && d.store), VQAF_E_DTYPE);
    GemmProb& P = args.p[i];
    P.A = d.A; P.Bm = d.Bm; P.bias = d.bias; P.C = reinterpret_cast<float*>(d.C);
    P.a_sg = d.a_sg; P.a_sm = d.a_sm; P.a_sk = d.a_sk;
    P.b_sg = d.b_sg; P.b_sk = d.b_sk; P.b_sn = d.b_sn;
    P.bias_sg = d.bias_sg; P.c_sg = d.c_sg; P.c_sm = d.c_sm;
    P.G = d.G; P.M = d.M; P.N = d.N; P.K = d.K;
    P.alpha = d.alpha;
    P.amode = pick_mode(d.A, d.a_dtype, d.a_sg, d.a_sm, d.a_sk, d.M, d.K);
    P.bmode = pick_mode(d.Bm, d.b_dtype, d.b_sg, d.b_sn, d.b_sk, d.N, d.K);
    P.a_bf16 = d.a_dtype == VQAF_BF16;
    P.b_bf16 = d.b_dtype == VQAF_BF16;
    P.bias_bf16 = d.bias_dtype == VQAF_BF16;
    P.tiles_x = (d.N + GT - 1) / GT;
    P.tiles_y = (d.M + GT - 1) / GT;
    // the whole group should fill the chip about once: long contractions are split over workgroups
    const long tiles = (long)P.tiles_x * P.tiles_y * d.G;
    int splits = 1;
    if (!d.store && total_tiles < 256 && d.K >= 4 * GK)
      while (splits < 16 && total_tiles * splits < 512 && d.K / (splits * 2) >= 2 * GK) splits *= 2;
    P.splits = splits;
    P.store = d.store != 0;
    P.c_bf16 = d.c_dtype == VQAF_BF16;
    P.k_per_split = ((d.K + splits - 1) / splits + GK - 1) / GK * GK;
    P.wg_begin = wg;
    VQAF_REQUIRE(tiles * splits < (1l << 24), VQAF_E_SHAPE);
    wg += (int)(tiles * splits);
  }
  hipLaunchKernelGGL(gemm_f32_group_kernel, dim3(wg), dim3(GTHREADS), 0, (hipStream_t)stream, args);
  VQAF_LAUNCH_CHECK();
}

// u_i (B,H,D) for several (query, in_proj) pairs of one shape: two launches in total (see
// vqaf_fold_query for the arithmetic).
static int fold_query_multi(const vqaf_fold_desc* descs_host, int n, int B, int D, int H, int dtype, int w_dtype,
                            uint64_t* count_state, uint64_t delta, vqaf_stream_t stream);

extern "C" int vqaf_fold_query_multi(const vqaf_fold_desc* descs_host, int n, int B, int D, int H, int dtype,
                                     int w_dtype, vqaf_stream_t stream) {
  VQAF_ENTRY();
  return fold_query_multi(descs_host, n, B, D, H, dtype, w_dtype, nullptr, 0, stream);
}

extern "C" int vqaf_fold_query_multi_advance(const vqaf_fold_desc* descs_host, int n, int B, int D, int H, int dtype,
                                             int w_dtype, uint64_t* count_state, uint64_t delta,
                                             vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(count_state != nullptr, VQAF_E_NULL);
  return fold_query_multi(descs_host, n, B, D, H, dtype, w_dtype, count_state, delta, stream);
}

static int fold_query_multi(const vqaf_fold_desc* descs_host, int n, int B, int D, int H, int dtype, int w_dtype,
                            uint64_t* count_state, uint64_t delta, vqaf_stream_t stream) {
  VQAF_REQUIRE(descs_host != nullptr, VQAF_E_NULL);
  VQAF_REQUIRE(n > 0 && n <= GEMM_MAX_SETS && B > 0 && D > 0 && H > 0 && D % H == 0, VQAF_E_SHAPE);
  VQAF_REQUIRE(vqaf_dtype_ok(dtype) && vqaf_dtype_ok(w_dtype), VQAF_E_DTYPE);
  bool fast = fold_bf16_applies(D, H, dtype, w_dtype);
  for (int i = 0; i < n && fast; ++i)  // 16-byte fragment loads
    fast = reinterpret_cast<uintptr_t>(descs_host[i].q) % 16 == 0 &&
           reinterpret_cast<uintptr_t>(descs_host[i].in_proj_weight) % 16 == 0 &&
           reinterpret_cast<uintptr_t>(descs_host[i].u) % 16 == 0;  // 16-byte stores of u
  if (fast) {
    FoldSets fs = {};
    for (int i = 0; i < n; ++i) {
      const vqaf_fold_desc& d = descs_host[i];
      VQAF_REQUIRE(d.q && d.in_proj_weight && d.in_proj_bias && d.qproj_ws && d.u, VQAF_E_NULL);
      fs.q[i] = static_cast<const bf16_t*>(d.q);
      fs.w[i] = static_cast<const bf16_t*>(d.in_proj_weight);
      fs.bias[i] = static_cast<const bf16_t*>(d.in_proj_bias);
      fs.qproj[i] = d.qproj_ws;
      fs.u[i] = d.u;
    }
    return fold_bf16_launch(fs, n, B, H, (hipStream_t)stream, count_state, delta);
  }
  if (count_state != nullptr) {  // the two-launch form has no side job: the count takes its own launch
    const int rc = vqaf_rng_advance(count_state, delta, stream);
    if (rc != VQAF_OK) return rc;
  }
  const int dh = D / H;
  const size_t esz = vqaf_esize(w_dtype);
  GemmSets s1 = {}, s2 = {};
  for (int i = 0; i < n; ++i) {
    const vqaf_fold_desc& d = descs_host[i];
    VQAF_REQUIRE(d.q && d.in_proj_weight && d.in_proj_bias && d.qproj_ws && d.u, VQAF_E_NULL);
    s1.A[i] = d.q;
    s1.Bm[i] = d.in_proj_weight;
    s1.bias[i] = d.in_proj_bias;
    s1.C[i] = d.qproj_ws;
    s2.A[i] = d.qproj_ws;
    s2.Bm[i] = reinterpret_cast<const unsigned char*>(d.in_proj_weight) + (size_t)D * D * esz;
    s2.bias[i] = nullptr;
    s2.C[i] = d.u;
  }
  s1.gper = 1;
  s2.gper = H;
  hipStream_t st = (hipStream_t)stream;
  int rc = gemm_launch(s1, n, dtype, 0, D, 1, w_dtype, 0, 1, D, w_dtype, 0, 0, D, 1, B, D, D, 1.0f, 0, st);
  if (rc != VQAF_OK) return rc;
  return gemm_launch(s2, n, VQAF_F32, dh, D, 1, w_dtype, (long)dh * D, D, 1, 0, 0, D, (long)H * D, H, B, D, dh,
                     1.0f / sqrtf((float)dh), 0, st);
}
