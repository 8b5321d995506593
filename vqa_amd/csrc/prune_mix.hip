// PrivacyAwareTokenPruning.forward in ONE launch: attention weights -> keep indices (ranking,
// adversarial swap) -> gather (+ noise) -> mix token.  (utils/fusion.py:26-94 of the reference.)
//
// A workgroup of 512 threads owns one (sample, 256-column slice): it ranks the sample's weights in
// LDS and registers, then streams the slice's rows once, each row going either to its keep slot or
// into the mix-token sum (row order, fixed summation order, no atomics).  At the sizes this op runs
// (128 samples x 577 tokens x 512 columns: one workgroup per CU) the separate ranking launch costs
// as much as the whole transfer; here its cost is the ranking of one row per CU, repeated by the
// sample's two slices.
//
// The streaming loop is written so that every memory instruction issues unconditionally: rows
// and slots that must not be touched get a buffer offset beyond the end of the buffer, which the
// hardware drops.  Loads and stores share one in-order counter, and a store under a branch makes the
// compiler assume it was not issued when it later waits for a load, so that wait would also wait
// for every store before it; with nothing conditional the waits are exact and the next batch of rows
// is in flight while the current one is stored.  (Issuing ALL rows before the ranking, to hide the
// ranking under them, was measured and is worse: hundreds of KB of queued loads per CU delay the
// instruction fetches of workgroups that start later by tens of microseconds.)
#include <stdlib.h>

#include "select.h"

namespace {

constexpr int PM_THREADS = 512;
constexpr int PM_LPR = 32;                    // lanes per row slice (8 columns each)
constexpr int PM_ROWS = PM_THREADS / PM_LPR;  // 16 rows per step
constexpr int PM_SLW = 8 * PM_LPR;            // 256 columns per slice
constexpr int PM_U = 4;                       // rows per lane in one batch of loads
constexpr int PM_NONE = 0x7fffffff;
constexpr unsigned int PM_OOB = 0x80000000u;  // beyond any buffer this kernel accepts (< 2 GiB per sample)
constexpr int PM_RSRC_FLAGS = 0x00020000;     // raw buffer, 32-bit data format

struct PruneMixArgs {
  const void* seq;         // (B,N,D)
  const float* w;          // (B,N) attention weights
  const int64_t* pos;      // (B,k_adv) injected keep-list positions (adv_mode 1)
  const int64_t* src;      // (B,k_adv) injected bottom-list positions
  const float* noise;      // (B,K,D) injected noise or NULL
  const uint64_t* sel_rng; // {seed, counter} of the adversarial draws (adv_mode 2)
  const uint64_t* noise_rng;
  void* out;               // (B, K + (alpha > 0), D)
  int64_t* idx_out;        // (B,K)
  unsigned long long salt;
  float alpha, noise_scale;
  int noise_kind, N, K, D, NP, adv_mode, k_adv, NPK, NPR, noise_ctr_off;
};

__host__ __device__ inline size_t pm_align16(size_t x) { return (x + 15) & ~(size_t)15; }
struct PmCarve {
  size_t keys, rank, s_pos, wl, pos, sidx, red, stat, misc, total;
};
__host__ __device__ inline PmCarve pm_carve(int N, int K, int NP, int k_adv) {
  PmCarve c;
  size_t o = 0;
  c.keys = o;  o += pm_align16((size_t)(NP + 2) * 8);
  c.rank = o;  o += pm_align16((size_t)N * 4);
  c.s_pos = o; o += pm_align16((size_t)(k_adv > 0 ? 2 * k_adv : 1) * 4);
  c.wl = o;    o += pm_align16((size_t)N * 4);
  c.pos = o;   o += pm_align16((size_t)N * 4);
  c.sidx = o;  o += pm_align16((size_t)K * 4);
  c.red = o;   o += (size_t)PM_THREADS * 8 * 4;
  c.stat = o;  o += pm_align16((size_t)2 * SEL_MAX_H * 4);
  c.misc = o;  o += 16;
  c.total = o;
  return c;
}

__device__ __forceinline__ void pm_noise8(float (&v)[8], const float* noise, int noise_kind, float noise_scale,
                                          unsigned long long seed, unsigned long long counter, size_t e) {
  vqaf_token_noise8(v, noise, noise_kind, noise_scale, seed, counter, e);
}

template <bool BF16, bool BF16_OUT>
__global__ void __launch_bounds__(PM_THREADS)
prune_mix_kernel(PruneMixArgs A) {
  constexpr int W = BF16 ? 1 : 2;        // 16-byte pieces per lane and input row
  constexpr unsigned int esz = BF16 ? 2 : 4, osz = BF16_OUT ? 2 : 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char pm_lds[];
  const int N = A.N, K = A.K, D = A.D, NP = A.NP, k_adv = A.k_adv;
  const PmCarve cv = pm_carve(N, K, NP, k_adv);
  SelectLds L;
  L.keys = reinterpret_cast<unsigned long long*>(pm_lds + cv.keys);
  L.sl = nullptr;
  L.rank = reinterpret_cast<int*>(pm_lds + cv.rank);
  L.s_pos = reinterpret_cast<int*>(pm_lds + cv.s_pos);
  L.wl = reinterpret_cast<float*>(pm_lds + cv.wl);
  L.stat = reinterpret_cast<float*>(pm_lds + cv.stat);
  int* pos = reinterpret_cast<int*>(pm_lds + cv.pos);
  int* sidx = reinterpret_cast<int*>(pm_lds + cv.sidx);
  float* red = reinterpret_cast<float*>(pm_lds + cv.red);
  int* misc = reinterpret_cast<int*>(pm_lds + cv.misc);

  const int b = blockIdx.y, slice = blockIdx.x, tid = threadIdx.x;
  const int g = tid / PM_LPR, li = tid % PM_LPR;
  const int d0 = slice * PM_SLW + li * 8;
  const bool active = d0 < D;
  const bool with_mix = A.alpha > 0.f;
  const int Kout = K + (with_mix ? 1 : 0);
  const bool has_noise = (A.noise != nullptr) || (A.noise_kind != VQAF_NOISE_NONE);
  const size_t seq_b = (size_t)b * N * D;

  // ---- keep list: weights -> ranking -> adversarial swap (same code as the selection kernels) ----
  for (int n = tid; n < N; n += PM_THREADS) L.wl[n] = A.w[(size_t)b * N + n];
  if (tid < 4) misc[tid] = 0;
  unsigned long long seed = 0, counter = 0;
  if (A.adv_mode == 2) {
    seed = A.sel_rng[0] + A.salt;
    counter = A.sel_rng[1];
  }
  __syncthreads();
  select_core(L, 0, N, K, NP, A.adv_mode, k_adv, A.NPK, A.NPR,
              A.pos ? A.pos + (size_t)b * k_adv : nullptr, A.src ? A.src + (size_t)b * k_adv : nullptr,
              seed, counter, (unsigned long long)b, nullptr, nullptr, nullptr, nullptr);
  // pos[n] = first keep slot naming row n
  for (int n = tid; n < N; n += PM_THREADS) pos[n] = PM_NONE;
  __syncthreads();
  for (int j = tid; j < K; j += PM_THREADS) {
    const int n = L.rank[j];
    sidx[j] = n;
    atomicMin(&pos[n], j);
  }
  __syncthreads();
  {
    int local = 0, dup = 0;
    for (int n = tid; n < N; n += PM_THREADS) local += (pos[n] == PM_NONE) ? 1 : 0;
    for (int j = tid; j < K; j += PM_THREADS)
      if (pos[sidx[j]] != j) dup = 1;  // a row named twice (possible only with injected draws)
    if (local) atomicAdd(&misc[0], local);
    if (dup) atomicOr(&misc[1], 1);
  }
  __syncthreads();
  const int R = misc[0];
  const bool dup_idx = misc[1] != 0;

  // ---- one pass over the slice's rows ----
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(A.seq)) + seq_b * esz, 0,
      (int)((unsigned int)N * (unsigned int)D * esz), PM_RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<unsigned char*>(A.out) + (size_t)b * Kout * D * osz, 0,
      (int)((unsigned int)Kout * (unsigned int)D * osz), PM_RSRC_FLAGS);
  const bool raw_copy = (BF16 == BF16_OUT);
  const bool store_rows = active && !has_noise;  // with noise every slot goes through the slot loop below
  float accd[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) accd[i] = 0.f;

  struct Batch {
    u32x4 r[PM_U][W];
    int p[PM_U];
  };
  auto issue = [&](Batch& bt, int t0) {
#pragma unroll
    for (int u = 0; u < PM_U; ++u) {
      const int n = g + (t0 + u) * PM_ROWS;
      const int p = n < N ? pos[n] : PM_NONE;
      bt.p[u] = p;
      // dropped rows are read only for the mix token; rows past the sample read as zeros
      const bool need = active && n < N && (with_mix || (p != PM_NONE && !has_noise));
      const unsigned int off = need ? ((unsigned int)n * (unsigned int)D + (unsigned int)d0) * esz : PM_OOB;
#pragma unroll
      for (int w = 0; w < W; ++w) bt.r[u][w] = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off + 16u * w, 0, 0);
    }
  };
  auto consume = [&](const Batch& bt) {
#pragma unroll
    for (int u = 0; u < PM_U; ++u) {
      const int p = bt.p[u];
      const bool kept = p != PM_NONE;
      const unsigned int off = (kept && store_rows) ? ((unsigned int)p * (unsigned int)D + (unsigned int)d0) * osz : PM_OOB;
      if (raw_copy) {
        // kept rows are raw 16-byte copies: bit-exact for every payload, NaNs included
#pragma unroll
        for (int w = 0; w < W; ++w) __builtin_amdgcn_raw_buffer_store_b128(bt.r[u][w], out_rsrc, off + 16u * w, 0, 0);
      } else {
        // bf16 rows into an f32 output: eight floats, two pieces
        const u32x4 x = bt.r[u][0];
        const u32x4 lo = {x.x << 16, x.x & 0xffff0000u, x.y << 16, x.y & 0xffff0000u};
        const u32x4 hi = {x.z << 16, x.z & 0xffff0000u, x.w << 16, x.w & 0xffff0000u};
        __builtin_amdgcn_raw_buffer_store_b128(lo, out_rsrc, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(hi, out_rsrc, off + 16u, 0, 0);
      }
      // dropped rows into the mix token (kept rows count as zero: selected, not multiplied, so a
      // NaN in a kept row stays out of the sum)
      u32x4 z[W];
#pragma unroll
      for (int w = 0; w < W; ++w) {
        z[w].x = kept ? 0u : bt.r[u][w].x;
        z[w].y = kept ? 0u : bt.r[u][w].y;
        z[w].z = kept ? 0u : bt.r[u][w].z;
        z[w].w = kept ? 0u : bt.r[u][w].w;
      }
      if constexpr (BF16) {
        accd[0] += bf16_lo(z[0].x); accd[1] += bf16_hi(z[0].x); accd[2] += bf16_lo(z[0].y); accd[3] += bf16_hi(z[0].y);
        accd[4] += bf16_lo(z[0].z); accd[5] += bf16_hi(z[0].z); accd[6] += bf16_lo(z[0].w); accd[7] += bf16_hi(z[0].w);
      } else {
        accd[0] += __uint_as_float(z[0].x); accd[1] += __uint_as_float(z[0].y);
        accd[2] += __uint_as_float(z[0].z); accd[3] += __uint_as_float(z[0].w);
        accd[4] += __uint_as_float(z[W - 1].x); accd[5] += __uint_as_float(z[W - 1].y);
        accd[6] += __uint_as_float(z[W - 1].z); accd[7] += __uint_as_float(z[W - 1].w);
      }
    }
  };
  {
    const int T = (N + PM_ROWS - 1) / PM_ROWS;
    Batch ba, bb;
    issue(ba, 0);
    for (int t0 = 0; t0 < T; t0 += 2 * PM_U) {
      issue(bb, t0 + PM_U);
      consume(ba);
      issue(ba, t0 + 2 * PM_U);
      consume(bb);
    }
  }

  // (after the rows, not before: a barrier waits for outstanding stores, and the ones above follow
  // this store into the memory system without waiting for it)
  if (slice == 0)
    for (int j = tid; j < K; j += PM_THREADS) A.idx_out[(size_t)b * K + j] = (int64_t)sidx[j];

  if (active && (dup_idx || has_noise)) {
    // Slot by slot, for what the pass above leaves out: every slot when noise is added (the draw is
    // addressed by the slot), otherwise only the later slots of a row named more than once.
    unsigned long long nseed = 0, ncounter = 0;
    if (A.noise == nullptr && A.noise_kind != VQAF_NOISE_NONE) {
      nseed = A.noise_rng[0];
      ncounter = A.noise_rng[1] + (unsigned long long)A.noise_ctr_off;
    }
    for (int j = g; j < K; j += PM_ROWS) {
      const int n = sidx[j];
      if (!has_noise && pos[n] == j) continue;
      float v[8];
      load8<BF16>(A.seq, seq_b + (size_t)n * D + d0, v);
      if (has_noise)
        pm_noise8(v, A.noise, A.noise_kind, A.noise_scale, nseed, ncounter, ((size_t)b * K + j) * D + d0);
      if (raw_copy && !has_noise) {
        const u32x4* s2 = reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(A.seq) +
                                                         (seq_b + (size_t)n * D + d0) * esz);
        u32x4* dst = reinterpret_cast<u32x4*>(reinterpret_cast<unsigned char*>(A.out) +
                                              (((size_t)b * Kout + j) * D + d0) * osz);
#pragma unroll
        for (int w = 0; w < W; ++w) dst[w] = s2[w];
      } else {
        store8<BF16_OUT>(A.out, ((size_t)b * Kout + j) * D + d0, v);
      }
    }
  }
  if (!with_mix) return;
  // fixed-order fold of the 16 row groups: remaining_sum / (count + 1e-10), then * alpha
#pragma unroll
  for (int i = 0; i < 8; ++i) red[tid * 8 + i] = accd[i];
  __syncthreads();
  const float cnt = (float)R + 1e-10f;
  if (tid < PM_SLW) {
    const int c = tid;
    float dr = 0.f;
    for (int gg = 0; gg < PM_ROWS; ++gg) dr += red[(gg * PM_LPR + c / 8) * 8 + (c & 7)];
    const int d = slice * PM_SLW + c;
    if (d < D) store1<BF16_OUT>(A.out, ((size_t)b * Kout + K) * D + d, A.alpha * (dr / cnt));
  }
}

}  // namespace

extern "C" int vqaf_prune_mix_supported(int N, int D, int dtype) {
  if (N <= 0 || D <= 0 || D % 8 != 0 || N > TK_MAX_N) return 0;
  if (dtype != VQAF_BF16 && dtype != VQAF_F32) return 0;
  // buffer offsets are 32-bit and the out-of-range marker is 2^31: a sample stays below that
  if ((unsigned long long)N * (unsigned long long)D * 4ull >= 0x80000000ull) return 0;
  // the per-sample state must fit 64 KiB of LDS whatever K and k_adv the caller then asks for
  // (K <= N, k_adv <= N / 2), so that "supported" never turns into a shape error at the launch
  return pm_carve(N, N, sel_next_pow2(N), N / 2).total <= 64 * 1024;
}

extern "C" int vqaf_prune_mix(const void* seq, const float* w, int B, int N, int K, int D, int dtype,
                              int adv_mode, int k_adv, const int64_t* pos, const int64_t* src,
                              const uint64_t* sel_rng_state, uint64_t salt, const float* noise, int noise_kind,
                              float noise_scale, const uint64_t* noise_rng_state, int noise_counter_offset,
                              float alpha, int out_dtype, void* out, int64_t* idx_out, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(seq && w && out && idx_out, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_dtype_ok(dtype) && vqaf_dtype_ok(out_dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(!(out_dtype == VQAF_BF16 && dtype == VQAF_F32), VQAF_E_DTYPE);
  VQAF_REQUIRE(B > 0 && B <= 65535 && N > 0 && K > 0 && K <= N && D > 0, VQAF_E_SHAPE);
  VQAF_REQUIRE(vqaf_prune_mix_supported(N, D, dtype), VQAF_E_SHAPE);
  VQAF_REQUIRE(D % 8 == 0 && vqaf_aligned16(seq) && vqaf_aligned16(out) && (noise == nullptr || vqaf_aligned16(noise)),
               VQAF_E_ALIGN);
  VQAF_REQUIRE(noise_kind >= VQAF_NOISE_NONE && noise_kind <= VQAF_NOISE_UNIFORM, VQAF_E_SHAPE);
  VQAF_REQUIRE(noise != nullptr || noise_kind == VQAF_NOISE_NONE || noise_rng_state != nullptr, VQAF_E_NULL);
  VQAF_REQUIRE(adv_mode >= 0 && adv_mode <= 2, VQAF_E_SHAPE);
  const int R = N - K;
  if (adv_mode != 0) {
    VQAF_REQUIRE(k_adv > 0 && k_adv <= K && k_adv <= R, VQAF_E_SHAPE);
    if (adv_mode == 1) VQAF_REQUIRE(pos != nullptr && src != nullptr, VQAF_E_NULL);
    if (adv_mode == 2) VQAF_REQUIRE(sel_rng_state != nullptr, VQAF_E_NULL);
  } else {
    k_adv = 0;
  }
  const int NP = sel_next_pow2(N);
  PruneMixArgs A{seq, w, pos, src, noise, sel_rng_state, noise_rng_state, out, idx_out, (unsigned long long)salt, alpha,
                 noise_scale, noise_kind, N, K, D, NP, adv_mode, k_adv, sel_next_pow2(K), sel_next_pow2(R > 0 ? R : 1),
                 noise_counter_offset};
  const size_t smem = pm_carve(N, K, NP, k_adv).total;
  VQAF_REQUIRE(smem <= 64 * 1024, VQAF_E_SHAPE);
  const dim3 grid((D + PM_SLW - 1) / PM_SLW, B), block(PM_THREADS);
  hipStream_t st = (hipStream_t)stream;
  const bool bf = dtype == VQAF_BF16, bfo = out_dtype == VQAF_BF16;
#define VQAF_PM(BF, BFO) hipLaunchKernelGGL((prune_mix_kernel<BF, BFO>), grid, block, smem, st, A)
  if (bf && bfo) VQAF_PM(true, true);
  else if (bf) VQAF_PM(true, false);
  else VQAF_PM(false, false);
#undef VQAF_PM
  VQAF_LAUNCH_CHECK();
}
