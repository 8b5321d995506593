// Gather / noise / remaining-mean mix token, its pooled variant, and their backward passes.
//
// Data layout: seq (B,N,D) row-major, one token row = D*s contiguous bytes.  A workgroup owns
// one (sample, column slice) pair: a slice is 8*LPR consecutive columns, handled by LPR lanes
// that each move 8 elements (16 B of bf16, 32 B of f32) per row, so a 256-thread workgroup has
// 256/LPR rows in flight per step and every row is read from HBM exactly once by exactly one
// workgroup.  Because a workgroup sees every token of its sample, the mix token (a sum over
// the dropped rows) needs no cross-workgroup reduction and no atomics: the sum order is fixed.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int GM_THREADS = 256;
constexpr int GM_UNROLL = 8;

__device__ __forceinline__ void zero8(float (&v)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = 0.f;
}

// Adds the 8 noise values of elements [e, e+8) (e multiple of 8) to v.
__device__ __forceinline__ void add_noise8(float (&v)[8], const float* noise, int noise_kind,
                                           float noise_scale, unsigned long long seed,
                                           unsigned long long counter, size_t e) {
  vqaf_token_noise8(v, noise, noise_kind, noise_scale, seed, counter, e);
}

// Shared-memory carve for the forward kernels.
struct FwdSmem {
  int* sidx;   // [K]  clamped index or -1
  int* pos;    // [N]  smallest j with idx[j] == n, GM_NONE if the row is dropped
  float* red;  // [GM_THREADS * 8]
  int* misc;   // [4]  {dropped rows, duplicate flag}
};
constexpr int GM_NONE = 0x7fffffff;
__host__ __device__ inline size_t fwd_smem_bytes(int N, int K) {
  return (((size_t)K * 4 + 15) & ~(size_t)15) + (((size_t)N * 4 + 15) & ~(size_t)15) +
         (size_t)GM_THREADS * 8 * 4 + 16;
}
__device__ inline FwdSmem fwd_carve(unsigned char* base, int N, int K) {
  FwdSmem s;
  s.sidx = reinterpret_cast<int*>(base);
  base += ((size_t)K * 4 + 15) & ~(size_t)15;
  s.pos = reinterpret_cast<int*>(base);
  base += ((size_t)N * 4 + 15) & ~(size_t)15;
  s.red = reinterpret_cast<float*>(base);
  base += (size_t)GM_THREADS * 8 * 4;
  s.misc = reinterpret_cast<int*>(base);
  return s;
}

// One streaming pass over the sample's rows in memory order: a kept row goes to its slot of the
// output (or into the pooled sum), a dropped row into the mix-token sum.  Rows named more than
// once in idx (never produced by the pruning module, allowed by torch.gather) are completed in a
// short serial tail.
// POOL == false: out is (B, Kout, D) with Kout = K + (alpha > 0).
// POOL == true : out is (B, ld_out); writes columns [col0, col0 + D) with the token mean.
template <bool BF16, bool BF16_OUT, int LPR, bool POOL>
__device__ __forceinline__ void
gather_mix_body(const void* __restrict__ seq, const int64_t* __restrict__ idx,
                const float* __restrict__ noise, int noise_kind, float noise_scale,
                const uint64_t* __restrict__ rng_state, float alpha, int N, int K, int D,
                void* __restrict__ out, int ld_out, int col0, const int b, const int slice) {
  constexpr int ROWS = GM_THREADS / LPR;
  constexpr int SLW = 8 * LPR;
  constexpr int U = 8;  // rows in flight per lane
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const FwdSmem sm = fwd_carve(smem_raw, N, K);

  const int tid = threadIdx.x;
  const int g = tid / LPR;
  const int li = tid % LPR;
  const int d0 = slice * SLW + li * 8;
  const bool active = d0 < D;
  const bool with_mix = alpha > 0.f;
  const int Kout = K + (with_mix ? 1 : 0);
  const bool has_noise = (noise != nullptr) || (noise_kind != VQAF_NOISE_NONE);

  unsigned long long seed = 0, counter = 0;
  if (noise == nullptr && noise_kind != VQAF_NOISE_NONE) {
    seed = rng_state[0];
    counter = rng_state[1];
  }

  // Token rows are read in memory order, so the first batch of loads does not depend on the
  // indices: it is issued before the index staging and lands while the staging runs.
  const size_t seq_b = (size_t)b * N * D;
  constexpr size_t esz = BF16 ? 2 : 4;
  constexpr int W = BF16 ? 1 : 2;  // 16-byte pieces per lane and row
  // loads are unconditional (row / column clamped into the sample) so a batch issues back to back;
  // rows past the end are simply never consumed
  const unsigned char* row0 = reinterpret_cast<const unsigned char*>(seq) + (seq_b + min(d0, D - 8)) * esz;
  u32x4 cur[U][W], nxt[U][W];
  auto load_batch = [&](u32x4 (&dst)[U][W], int n0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int n = min(n0 + u * ROWS, N - 1);
      const u32x4* src = reinterpret_cast<const u32x4*>(row0 + (size_t)n * D * esz);
#pragma unroll
      for (int w = 0; w < W; ++w) dst[u][w] = src[w];
    }
  };
  load_batch(cur, g);

  // ---- index staging: pos[n] = first j naming row n ----
  for (int n = tid; n < N; n += GM_THREADS) sm.pos[n] = GM_NONE;
  if (tid < 2) sm.misc[tid] = 0;
  __syncthreads();
  const int64_t* idx_b = idx + (size_t)b * K;
  for (int j = tid; j < K; j += GM_THREADS) {
    const int64_t n = idx_b[j];
    const bool ok = n >= 0 && n < (int64_t)N;
    sm.sidx[j] = ok ? (int)n : -1;
    if (ok) atomicMin(&sm.pos[n], j);
  }
  __syncthreads();
  {
    int local = 0, odd = 0;
    for (int n = tid; n < N; n += GM_THREADS) local += (sm.pos[n] == GM_NONE) ? 1 : 0;
    for (int j = tid; j < K; j += GM_THREADS) {
      const int n = sm.sidx[j];
      if (n < 0 || sm.pos[n] != j) odd = 1;  // out-of-range or repeated index
    }
    if (local) atomicAdd(&sm.misc[0], local);
    if (odd) atomicOr(&sm.misc[1], 1);
  }
  __syncthreads();
  const int R = sm.misc[0];
  const bool odd_idx = sm.misc[1] != 0;

  float acck[8], accd[8];
  zero8(acck);
  zero8(accd);
  const bool raw_copy = !POOL && BF16 == BF16_OUT && !has_noise;

  auto unpack = [&](const u32x4 (&r)[W], float (&v)[8]) {
    if constexpr (BF16) {
      v[0] = bf16_lo(r[0].x); v[1] = bf16_hi(r[0].x); v[2] = bf16_lo(r[0].y); v[3] = bf16_hi(r[0].y);
      v[4] = bf16_lo(r[0].z); v[5] = bf16_hi(r[0].z); v[6] = bf16_lo(r[0].w); v[7] = bf16_hi(r[0].w);
    } else {
      v[0] = __uint_as_float(r[0].x); v[1] = __uint_as_float(r[0].y);
      v[2] = __uint_as_float(r[0].z); v[3] = __uint_as_float(r[0].w);
      v[4] = __uint_as_float(r[W - 1].x); v[5] = __uint_as_float(r[W - 1].y);
      v[6] = __uint_as_float(r[W - 1].z); v[7] = __uint_as_float(r[W - 1].w);
    }
  };

  if (active) {
    for (int n0 = g; n0 < N; n0 += U * ROWS) {
      const bool more = n0 + U * ROWS < N;
      load_batch(nxt, n0 + U * ROWS);  // unconditional (clamped): keeps the waits counted
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int n = n0 + u * ROWS;
        if (n < N) {
          const int p = sm.pos[n];
          float v[8];
          if (p != GM_NONE) {
            if (raw_copy) {
              // kept rows are raw 16-byte copies: bit-exact for every payload, NaNs included
              u32x4* dst = reinterpret_cast<u32x4*>(
                  reinterpret_cast<unsigned char*>(out) + (((size_t)b * Kout + p) * D + d0) * esz);
#pragma unroll
              for (int w = 0; w < W; ++w) dst[w] = cur[u][w];
            } else {
              unpack(cur[u], v);
              if (has_noise)
                add_noise8(v, noise, noise_kind, noise_scale, seed, counter, ((size_t)b * K + p) * D + d0);
              if constexpr (POOL) {
#pragma unroll
                for (int i = 0; i < 8; ++i) acck[i] += v[i];
              } else {
                store8<BF16_OUT>(out, ((size_t)b * Kout + p) * D + d0, v);
              }
            }
          } else if (with_mix) {
            unpack(cur[u], v);
#pragma unroll
            for (int i = 0; i < 8; ++i) accd[i] += v[i];
          }
        }
      }
      if (more) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int w = 0; w < W; ++w) cur[u][w] = nxt[u][w];
      }
    }
    if (odd_idx) {
      // slots whose index is out of range (zero row) or repeats an earlier one (same row again)
      for (int j = g; j < K; j += ROWS) {
        const int n = sm.sidx[j];
        if (n >= 0 && sm.pos[n] == j) continue;
        float v[8];
        zero8(v);
        if (n >= 0) {
          load8<BF16>(seq, seq_b + (size_t)n * D + d0, v);
          if (has_noise)
            add_noise8(v, noise, noise_kind, noise_scale, seed, counter, ((size_t)b * K + j) * D + d0);
        }
        if constexpr (POOL) {
#pragma unroll
          for (int i = 0; i < 8; ++i) acck[i] += v[i];
        } else if (BF16 == BF16_OUT && !has_noise && n >= 0) {
          const u32x4* src = reinterpret_cast<const u32x4*>(
              reinterpret_cast<const unsigned char*>(seq) + (seq_b + (size_t)n * D + d0) * esz);
          u32x4* dst = reinterpret_cast<u32x4*>(
              reinterpret_cast<unsigned char*>(out) + (((size_t)b * Kout + j) * D + d0) * esz);
          dst[0] = src[0];
          if (!BF16) dst[1] = src[1];
        } else {
          store8<BF16_OUT>(out, ((size_t)b * Kout + j) * D + d0, v);
        }
      }
    }
  }
  if (!with_mix && !POOL) return;

  // ---- fixed-order fold of the 256/LPR row groups ----
  // remaining_sum / (count + 1e-10), then * alpha  (same op order as the reference)
  const float cnt = (float)R + 1e-10f;
  float sel[(SLW + GM_THREADS - 1) / GM_THREADS];
  if constexpr (POOL) {
#pragma unroll
    for (int i = 0; i < 8; ++i) sm.red[tid * 8 + i] = acck[i];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < (SLW + GM_THREADS - 1) / GM_THREADS; ++q) {
      const int c = tid + q * GM_THREADS;  // column within the slice
      sel[q] = 0.f;
      if (c < SLW)
        for (int gg = 0; gg < ROWS; ++gg) sel[q] += sm.red[(gg * LPR + c / 8) * 8 + (c & 7)];
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) sm.red[tid * 8 + i] = accd[i];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < (SLW + GM_THREADS - 1) / GM_THREADS; ++q) {
    const int c = tid + q * GM_THREADS;
    if (c < SLW) {
      float dr = 0.f;
      for (int gg = 0; gg < ROWS; ++gg) dr += sm.red[(gg * LPR + c / 8) * 8 + (c & 7)];
      const int d = slice * SLW + c;
      if (d < D) {
        if constexpr (POOL) {
          float tot = sel[q];
          if (with_mix) tot += alpha * (dr / cnt);
          store1<BF16_OUT>(out, (size_t)b * ld_out + col0 + d, tot / (float)Kout);
        } else {
          store1<BF16_OUT>(out, ((size_t)b * Kout + K) * D + d, alpha * (dr / cnt));
        }
      }
    }
  }
}

template <bool BF16, bool BF16_OUT, int LPR, bool POOL>
__global__ void __launch_bounds__(GM_THREADS)
gather_mix_kernel(const void* __restrict__ seq, const int64_t* __restrict__ idx,
                  const float* __restrict__ noise, int noise_kind, float noise_scale,
                  const uint64_t* __restrict__ rng_state, float alpha, int N, int K, int D,
                  void* __restrict__ out, int ld_out, int col0) {
  gather_mix_body<BF16, BF16_OUT, LPR, POOL>(seq, idx, noise, noise_kind, noise_scale, rng_state, alpha, N,
                                             K, D, out, ld_out, col0, blockIdx.y, blockIdx.x);
}

// Several modalities (token sets of different length) pooled by ONE launch.
constexpr int GM_MAX_MODS = 4;
struct PoolMod {
  const void* seq;
  const int64_t* idx;
  const float* noise;
  const uint64_t* rng_state;
  float noise_scale, alpha;
  int noise_kind, N, K, col0;
};
struct PoolMods {
  PoolMod m[GM_MAX_MODS];
};
template <bool BF16, bool BF16_OUT, int LPR>
__global__ void __launch_bounds__(GM_THREADS)
gather_mix_pool_multi_kernel(PoolMods mods, int n_mod, int D, void* __restrict__ out, int ld_out) {
  // Modalities along z: the long (vision) workgroups are dispatched first and the short (text)
  // ones fill the tail of the launch.  Measured alternatives: interleaving the modalities along y
  // (+16 us) and pooling all modalities in turn inside one workgroup (+5 us).
  const PoolMod& m = mods.m[blockIdx.z];
  gather_mix_body<BF16, BF16_OUT, LPR, true>(m.seq, m.idx, m.noise, m.noise_kind, m.noise_scale,
                                             m.rng_state, m.alpha, m.N, m.K, D, out, ld_out, m.col0,
                                             blockIdx.y, blockIdx.x);
}

// ------------------------------- backward -------------------------------------------------
__host__ __device__ inline size_t bwd_smem_bytes(int N, int K) {
  return (((size_t)K * 4 + 15) & ~(size_t)15) + (((size_t)N * 4 + 15) & ~(size_t)15) + 16;
}

// POOL == false: grad_out is (B, Kout, D).  POOL == true: grad_out is (B, ld_out), columns
// [col0, col0+D).  pos[n] = smallest j with idx[j] == n (INT_MAX if dropped); for POOL it
// holds the multiplicity of n in idx instead.
template <bool BF16_G, bool BF16, int LPR, bool POOL>
__device__ __forceinline__ void
gather_mix_bwd_body(const void* __restrict__ grad_out, int ld_out, int col0,
                    const int64_t* __restrict__ idx, float alpha, int N, int K, int D,
                    void* __restrict__ grad_seq, const int b, const int slice) {
  constexpr int ROWS = GM_THREADS / LPR;
  constexpr int SLW = 8 * LPR;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  int* sidx = reinterpret_cast<int*>(smem_raw);
  int* pos = reinterpret_cast<int*>(smem_raw + (((size_t)K * 4 + 15) & ~(size_t)15));
  int* misc = reinterpret_cast<int*>(smem_raw + (((size_t)K * 4 + 15) & ~(size_t)15) +
                                     (((size_t)N * 4 + 15) & ~(size_t)15));

  const int tid = threadIdx.x;
  const int g = tid / LPR;
  const int li = tid % LPR;
  const int d0 = slice * SLW + li * 8;
  const bool active = d0 < D;
  const bool with_mix = alpha > 0.f;
  const int Kout = K + (with_mix ? 1 : 0);
  constexpr int NONE = POOL ? 0 : 0x7fffffff;
  constexpr int U = 8;
  constexpr int WG = BF16_G ? 1 : 2;  // 16-byte pieces per lane and grad_out row
  constexpr size_t gsz = BF16_G ? 2 : 4;

  // grad_out rows are read in memory order (j = 0, 1, ...), so the first batch does not depend
  // on the indices and is issued before the index staging.
  u32x4 cur[U][WG], nxt[U][WG];
  const unsigned char* grow0 = nullptr;
  if constexpr (!POOL)
    grow0 = reinterpret_cast<const unsigned char*>(grad_out) + ((size_t)b * Kout * D + min(d0, D - 8)) * gsz;
  auto load_batch = [&](u32x4 (&dst)[U][WG], int j0) {  // unconditional: rows clamped, extras unused
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = min(j0 + u * ROWS, K - 1);
      const u32x4* src = reinterpret_cast<const u32x4*>(grow0 + (size_t)j * D * gsz);
#pragma unroll
      for (int w = 0; w < WG; ++w) dst[u][w] = src[w];
    }
  };
  if constexpr (!POOL) load_batch(cur, g);

  for (int n = tid; n < N; n += GM_THREADS) pos[n] = NONE;
  if (tid < 2) misc[tid] = 0;
  __syncthreads();
  const int64_t* idx_b = idx + (size_t)b * K;
  for (int j = tid; j < K; j += GM_THREADS) {
    const int64_t n = idx_b[j];
    const bool ok = n >= 0 && n < (int64_t)N;
    sidx[j] = ok ? (int)n : -1;
    if (ok) {
      if constexpr (POOL) atomicAdd(&pos[n], 1);
      else atomicMin(&pos[n], j);
    }
  }
  __syncthreads();
  {  // count dropped rows; detect duplicate indices
    int local = 0;
    for (int n = tid; n < N; n += GM_THREADS) local += (pos[n] == NONE) ? 1 : 0;
    if (local) atomicAdd(&misc[0], local);
    if constexpr (!POOL) {
      int dup = 0;
      for (int j = tid; j < K; j += GM_THREADS) {
        const int n = sidx[j];
        if (n >= 0 && pos[n] != j) dup = 1;
      }
      if (dup) atomicOr(&misc[1], 1);
    }
  }
  __syncthreads();
  const int R = misc[0];
  const float cnt = (float)R + 1e-10f;
  const size_t seq_b = (size_t)b * N * D;

  if (active) {
    float gmix[8], gsel[8];
    zero8(gmix);
    zero8(gsel);
    if constexpr (POOL) {
      // pooled = (sum_sel + alpha * sum_drop / cnt) / Kout
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float go = load1<BF16_G>(grad_out, (size_t)b * ld_out + col0 + d0 + i) / (float)Kout;
        gsel[i] = go;
        gmix[i] = with_mix ? (alpha * go) / cnt : 0.f;
      }
    } else if (with_mix) {
      float t[8];
      load8<BF16_G>(grad_out, ((size_t)b * Kout + K) * D + d0, t);
#pragma unroll
      for (int i = 0; i < 8; ++i) gmix[i] = (alpha * t[i]) / cnt;
    }
    if constexpr (POOL && BF16) {
      // A row's gradient is one of two vectors (dropped row: the mix share; row kept once: the pooled
      // share): both are packed once, so the loop is a multiplicity lookup, a select and a 16-byte
      // store per row -- a fill with two patterns.  Rows kept more than once (an adversarial draw
      // that names a kept row again) scale the pooled share.
      u32x4 pk_mix, pk_sel;
      pk_mix.x = pack_bf16x2(gmix[0], gmix[1]); pk_mix.y = pack_bf16x2(gmix[2], gmix[3]);
      pk_mix.z = pack_bf16x2(gmix[4], gmix[5]); pk_mix.w = pack_bf16x2(gmix[6], gmix[7]);
      pk_sel.x = pack_bf16x2(gsel[0], gsel[1]); pk_sel.y = pack_bf16x2(gsel[2], gsel[3]);
      pk_sel.z = pack_bf16x2(gsel[4], gsel[5]); pk_sel.w = pack_bf16x2(gsel[6], gsel[7]);
      bf16_t* const rows = reinterpret_cast<bf16_t*>(grad_seq) + seq_b + d0;
      for (int n0 = g; n0 < N; n0 += GM_UNROLL * ROWS) {
        int pp[GM_UNROLL];
#pragma unroll
        for (int u = 0; u < GM_UNROLL; ++u) pp[u] = pos[min(n0 + u * ROWS, N - 1)];
#pragma unroll
        for (int u = 0; u < GM_UNROLL; ++u) {
          const int n = n0 + u * ROWS;
          if (n < N) {
            u32x4 v = pp[u] == 0 ? pk_mix : pk_sel;
            if (pp[u] > 1) {
              v.x = pack_bf16x2(gsel[0] * (float)pp[u], gsel[1] * (float)pp[u]);
              v.y = pack_bf16x2(gsel[2] * (float)pp[u], gsel[3] * (float)pp[u]);
              v.z = pack_bf16x2(gsel[4] * (float)pp[u], gsel[5] * (float)pp[u]);
              v.w = pack_bf16x2(gsel[6] * (float)pp[u], gsel[7] * (float)pp[u]);
            }
            // streamed past the caches: ordinary stores finish this kernel 10 us sooner (42 vs 52 us for
            // 238 MB) but leave the Infinity Cache full of dirty lines whose write-back then slows the
            // next kernels that stream from HBM by more than that (measured on the whole step)
            store_nt16(rows + (size_t)n * D, v);
          }
        }
      }
    } else if constexpr (POOL) {
      for (int n0 = g; n0 < N; n0 += GM_UNROLL * ROWS) {
#pragma unroll
        for (int u = 0; u < GM_UNROLL; ++u) {
          const int n = n0 + u * ROWS;
          if (n < N) {
            const int p = pos[n];
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (p == 0) ? gmix[i] : gsel[i] * (float)p;
            store8_nt<BF16>(grad_seq, seq_b + (size_t)n * D + d0, v);
          }
        }
      }
    } else {
      // kept rows: grad_out row j goes to token row idx[j] (the first j naming a row wins here,
      // later ones are added in the tail below)
      for (int j0 = g; j0 < K; j0 += U * ROWS) {
        const bool more = j0 + U * ROWS < K;
        load_batch(nxt, j0 + U * ROWS);  // unconditional (clamped): keeps the waits counted
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int j = j0 + u * ROWS;
          if (j < K) {
            const int n = sidx[j];
            if (n >= 0 && pos[n] == j) {
              if constexpr (BF16_G == BF16) {
                u32x4* dst = reinterpret_cast<u32x4*>(reinterpret_cast<unsigned char*>(grad_seq) +
                                                      (seq_b + (size_t)n * D + d0) * gsz);
#pragma unroll
                for (int w = 0; w < WG; ++w) store_nt16(dst + w, cur[u][w]);
              } else {
                float v[8];
                if constexpr (BF16_G) {
                  v[0] = bf16_lo(cur[u][0].x); v[1] = bf16_hi(cur[u][0].x);
                  v[2] = bf16_lo(cur[u][0].y); v[3] = bf16_hi(cur[u][0].y);
                  v[4] = bf16_lo(cur[u][0].z); v[5] = bf16_hi(cur[u][0].z);
                  v[6] = bf16_lo(cur[u][0].w); v[7] = bf16_hi(cur[u][0].w);
                } else {
                  v[0] = __uint_as_float(cur[u][0].x); v[1] = __uint_as_float(cur[u][0].y);
                  v[2] = __uint_as_float(cur[u][0].z); v[3] = __uint_as_float(cur[u][0].w);
                  v[4] = __uint_as_float(cur[u][WG - 1].x); v[5] = __uint_as_float(cur[u][WG - 1].y);
                  v[6] = __uint_as_float(cur[u][WG - 1].z); v[7] = __uint_as_float(cur[u][WG - 1].w);
                }
                store8_nt<BF16>(grad_seq, seq_b + (size_t)n * D + d0, v);
              }
            }
          }
        }
        if (more) {
#pragma unroll
          for (int u = 0; u < U; ++u)
#pragma unroll
            for (int w = 0; w < WG; ++w) cur[u][w] = nxt[u][w];
        }
      }
      // dropped rows: the broadcast mix gradient
      for (int n = g; n < N; n += ROWS)
        if (pos[n] == NONE) store8_nt<BF16>(grad_seq, seq_b + (size_t)n * D + d0, gmix);
    }
  }
  if constexpr (!POOL) {
    if (misc[1] != 0) {
      // Duplicate indices (never produced by the pruning module itself): add the rows that
      // lost the race, in j order, by the same lanes that wrote the row.
      __syncthreads();
      for (int j = 0; j < K; ++j) {
        const int n = sidx[j];
        if (n >= 0 && pos[n] != j && g == 0 && active) {
          float a[8], c[8];
          load8<BF16>(grad_seq, seq_b + (size_t)n * D + d0, a);
          load8<BF16_G>(grad_out, ((size_t)b * Kout + j) * D + d0, c);
#pragma unroll
          for (int i = 0; i < 8; ++i) a[i] += c[i];
          store8<BF16>(grad_seq, seq_b + (size_t)n * D + d0, a);
        }
      }
    }
  }
}

template <bool BF16_G, bool BF16, int LPR, bool POOL>
__global__ void __launch_bounds__(GM_THREADS)
gather_mix_bwd_kernel(const void* __restrict__ grad_out, int ld_out, int col0,
                      const int64_t* __restrict__ idx, float alpha, int N, int K, int D,
                      void* __restrict__ grad_seq) {
  gather_mix_bwd_body<BF16_G, BF16, LPR, POOL>(grad_out, ld_out, col0, idx, alpha, N, K, D, grad_seq,
                                               blockIdx.y, blockIdx.x);
}

struct PoolBwdMod {
  const int64_t* idx;
  void* grad_seq;  // NULL: this modality needs no gradient
  float alpha;
  int N, K, col0;
};
struct PoolBwdMods {
  PoolBwdMod m[GM_MAX_MODS];
};
template <bool BF16_G, bool BF16, int LPR>
__global__ void __launch_bounds__(GM_THREADS)
gather_mix_pool_bwd_multi_kernel(const void* __restrict__ grad_out, int ld_out, PoolBwdMods mods, int n_mod,
                                 int D) {
  const PoolBwdMod& m = mods.m[blockIdx.z];
  if (m.grad_seq == nullptr) return;
  gather_mix_bwd_body<BF16_G, BF16, LPR, true>(grad_out, ld_out, m.col0, m.idx, m.alpha, m.N, m.K, D,
                                               m.grad_seq, blockIdx.y, blockIdx.x);
}

// ------------------------------- plain mean pool ------------------------------------------
template <bool BF16, bool BF16_OUT, int LPR>
__global__ void __launch_bounds__(GM_THREADS)
mean_pool_kernel(const void* __restrict__ x, int M, int D, void* __restrict__ out, int ld_out,
                 int col0) {
  constexpr int ROWS = GM_THREADS / LPR;
  constexpr int SLW = 8 * LPR;
  __shared__ float red[GM_THREADS * 8];
  const int tid = threadIdx.x, b = blockIdx.y, g = tid / LPR, li = tid % LPR;
  const int d0 = blockIdx.x * SLW + li * 8;
  float acc[8];
  zero8(acc);
  if (d0 < D) {
    for (int j0 = g; j0 < M; j0 += GM_UNROLL * ROWS) {
      float v[GM_UNROLL][8];
#pragma unroll
      for (int u = 0; u < GM_UNROLL; ++u)  // unconditional loads (row clamped), masked when added
        load8<BF16>(x, ((size_t)b * M + min(j0 + u * ROWS, M - 1)) * D + d0, v[u]);
#pragma unroll
      for (int u = 0; u < GM_UNROLL; ++u) {
        const bool keep = j0 + u * ROWS < M;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += keep ? v[u][i] : 0.f;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) red[tid * 8 + i] = acc[i];
  __syncthreads();
  for (int c = tid; c < SLW; c += GM_THREADS) {
    float s = 0.f;
    for (int gg = 0; gg < ROWS; ++gg) s += red[(gg * LPR + c / 8) * 8 + (c & 7)];
    const int d = blockIdx.x * SLW + c;
    if (d < D) store1<BF16_OUT>(out, (size_t)b * ld_out + col0 + d, s / (float)M);
  }
}

template <bool BF16_G, bool BF16, int LPR>
__global__ void __launch_bounds__(GM_THREADS)
mean_pool_bwd_kernel(const void* __restrict__ grad_out, int ld_out, int col0, int M, int D,
                     void* __restrict__ grad_x) {
  constexpr int ROWS = GM_THREADS / LPR;
  constexpr int SLW = 8 * LPR;
  const int tid = threadIdx.x, b = blockIdx.y, g = tid / LPR, li = tid % LPR;
  const int d0 = blockIdx.x * SLW + li * 8;
  if (d0 >= D) return;
  float gv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
    gv[i] = load1<BF16_G>(grad_out, (size_t)b * ld_out + col0 + d0 + i) / (float)M;
  for (int j = g; j < M; j += ROWS) store8<BF16>(grad_x, ((size_t)b * M + j) * D + d0, gv);
}

// Picks the slice width so that small batches still launch >= ~1024 workgroups.
inline int pick_lpr(int B, int D) {
  static const int forced = [] {
    const char* e = getenv("VQAF_GM_LPR");  // tuning knob: 8, 16, 32 or 64
    return e ? atoi(e) : 0;
  }();
  if (forced == 8 || forced == 16 || forced == 32 || forced == 64) return forced;
  // measured on MI355X (bf16, D = 512): 256-column slices once they still give >= 2048
  // workgroups, 128-column slices down to 256 workgroups, 64-column slices for tiny batches
  if ((long)B * ((D + 255) / 256) >= 2048) return 32;
  if ((long)B * ((D + 127) / 128) >= 256) return 16;
  return 8;
}

template <bool POOL>
int launch_fwd(const void* seq, const int64_t* idx, const float* noise, int noise_kind,
               float noise_scale, const uint64_t* rng_state, float alpha, int B, int N, int K,
               int D, int dtype, int out_dtype, void* out, int ld_out, int col0,
               hipStream_t st) {
  const int lpr = pick_lpr(B, D);
  const dim3 grid((D + 8 * lpr - 1) / (8 * lpr), B), block(GM_THREADS);
  const size_t smem = fwd_smem_bytes(N, K);
#define VQAF_GM_LAUNCH(BF, BFO, L)                                                              \
  hipLaunchKernelGGL((gather_mix_kernel<BF, BFO, L, POOL>), grid, block, smem, st, seq, idx,     \
                     noise, noise_kind, noise_scale, rng_state, alpha, N, K, D, out, ld_out, col0)
  const bool bf = dtype == VQAF_BF16, bfo = out_dtype == VQAF_BF16;
#define VQAF_GM_BY_DT(L)                                   \
  do {                                                     \
    if (bf && bfo) VQAF_GM_LAUNCH(true, true, L);          \
    else if (bf) VQAF_GM_LAUNCH(true, false, L);           \
    else if (!bfo) VQAF_GM_LAUNCH(false, false, L);        \
    else VQAF_GM_LAUNCH(false, true, L);                   \
  } while (0)
  if (lpr == 64) VQAF_GM_BY_DT(64);
  else if (lpr == 32) VQAF_GM_BY_DT(32);
  else if (lpr == 16) VQAF_GM_BY_DT(16);
  else VQAF_GM_BY_DT(8);
#undef VQAF_GM_BY_DT
#undef VQAF_GM_LAUNCH
  VQAF_LAUNCH_CHECK();
}

template <bool POOL>
int launch_bwd(const void* grad_out, int ld_out, int col0, int g_dtype, const int64_t* idx,
               float alpha, int B, int N, int K, int D, int dtype, void* grad_seq,
               hipStream_t st) {
  const int lpr = pick_lpr(B, D);
  const dim3 grid((D + 8 * lpr - 1) / (8 * lpr), B), block(GM_THREADS);
  const size_t smem = bwd_smem_bytes(N, K);
#define VQAF_GMB_LAUNCH(BFG, BF, L)                                                         \
  hipLaunchKernelGGL((gather_mix_bwd_kernel<BFG, BF, L, POOL>), grid, block, smem, st,       \
                     grad_out, ld_out, col0, idx, alpha, N, K, D, grad_seq)
  const bool bfg = g_dtype == VQAF_BF16, bf = dtype == VQAF_BF16;
#define VQAF_GMB_BY_DT(L)                                  \
  do {                                                     \
    if (bfg && bf) VQAF_GMB_LAUNCH(true, true, L);         \
    else if (bfg) VQAF_GMB_LAUNCH(true, false, L);         \
    else if (bf) VQAF_GMB_LAUNCH(false, true, L);          \
    else VQAF_GMB_LAUNCH(false, false, L);                 \
  } while (0)
  if (lpr == 64) VQAF_GMB_BY_DT(64);
  else if (lpr == 32) VQAF_GMB_BY_DT(32);
  else if (lpr == 16) VQAF_GMB_BY_DT(16);
  else VQAF_GMB_BY_DT(8);
#undef VQAF_GMB_BY_DT
#undef VQAF_GMB_LAUNCH
  VQAF_LAUNCH_CHECK();
}

int check_common(const void* seq, const int64_t* idx, int B, int N, int K, int D, int dtype,
                 int out_dtype) {
  VQAF_REQUIRE(seq != nullptr && idx != nullptr, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_dtype_ok(dtype) && vqaf_dtype_ok(out_dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(B > 0 && N > 0 && K > 0 && K <= N && D > 0, VQAF_E_SHAPE);
  VQAF_REQUIRE(N <= 16384 && B <= 65535, VQAF_E_SHAPE);
  VQAF_REQUIRE(D % 8 == 0, VQAF_E_ALIGN);
  VQAF_REQUIRE(vqaf_aligned16(seq), VQAF_E_ALIGN);
  return VQAF_OK;
}

}  // namespace

extern "C" int vqaf_gather_mix(const void* seq, const int64_t* idx, const float* noise,
                               int noise_kind, float noise_scale, const uint64_t* rng_state,
                               float alpha, int B, int N, int K, int D, int dtype, int out_dtype,
                               void* out, vqaf_stream_t stream) {
  VQAF_ENTRY();
  const int rc = check_common(seq, idx, B, N, K, D, dtype, out_dtype);
  if (rc != VQAF_OK) return rc;
  VQAF_REQUIRE(out != nullptr, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_aligned16(out) && (noise == nullptr || vqaf_aligned16(noise)), VQAF_E_ALIGN);
  VQAF_REQUIRE(noise_kind >= VQAF_NOISE_NONE && noise_kind <= VQAF_NOISE_UNIFORM, VQAF_E_SHAPE);
  VQAF_REQUIRE(noise != nullptr || noise_kind == VQAF_NOISE_NONE || rng_state != nullptr, VQAF_E_NULL);
  VQAF_REQUIRE(!(out_dtype == VQAF_BF16 && dtype == VQAF_F32), VQAF_E_DTYPE);
  return launch_fwd<false>(seq, idx, noise, noise_kind, noise_scale, rng_state, alpha, B, N, K, D,
                           dtype, out_dtype, out, 0, 0, (hipStream_t)stream);
}

extern "C" int vqaf_gather_mix_pool(const void* seq, const int64_t* idx, const float* noise,
                                    int noise_kind, float noise_scale, const uint64_t* rng_state,
                                    float alpha, int B, int N, int K, int D, int dtype, void* out,
                                    int ld_out, int col0, int out_dtype, vqaf_stream_t stream) {
  VQAF_ENTRY();
  const int rc = check_common(seq, idx, B, N, K, D, dtype, out_dtype);
  if (rc != VQAF_OK) return rc;
  VQAF_REQUIRE(out != nullptr, VQAF_E_NULL);
  VQAF_REQUIRE(noise == nullptr || vqaf_aligned16(noise), VQAF_E_ALIGN);
  VQAF_REQUIRE(noise_kind >= VQAF_NOISE_NONE && noise_kind <= VQAF_NOISE_UNIFORM, VQAF_E_SHAPE);
  VQAF_REQUIRE(noise != nullptr || noise_kind == VQAF_NOISE_NONE || rng_state != nullptr, VQAF_E_NULL);
  VQAF_REQUIRE(col0 >= 0 && ld_out >= col0 + D, VQAF_E_SHAPE);
  return launch_fwd<true>(seq, idx, noise, noise_kind, noise_scale, rng_state, alpha, B, N, K, D,
                          dtype, out_dtype, out, ld_out, col0, (hipStream_t)stream);
}

extern "C" int vqaf_gather_mix_bwd(const void* grad_out, const int64_t* idx, float alpha, int B,
                                   int N, int K, int D, int g_dtype, int dtype, void* grad_seq,
                                   vqaf_stream_t stream) {
  VQAF_ENTRY();
  const int rc = check_common(grad_out, idx, B, N, K, D, g_dtype, dtype);
  if (rc != VQAF_OK) return rc;
  VQAF_REQUIRE(grad_seq != nullptr, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_aligned16(grad_seq), VQAF_E_ALIGN);
  return launch_bwd<false>(grad_out, 0, 0, g_dtype, idx, alpha, B, N, K, D, dtype, grad_seq,
                           (hipStream_t)stream);
}

extern "C" int vqaf_gather_mix_pool_bwd(const void* grad_out, int ld_out, int col0, int g_dtype,
                                        const int64_t* idx, float alpha, int B, int N, int K,
                                        int D, int dtype, void* grad_seq, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(grad_out != nullptr && idx != nullptr && grad_seq != nullptr, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_dtype_ok(dtype) && vqaf_dtype_ok(g_dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(B > 0 && N > 0 && K > 0 && K <= N && D > 0 && N <= 16384 && B <= 65535, VQAF_E_SHAPE);
  VQAF_REQUIRE(col0 >= 0 && ld_out >= col0 + D, VQAF_E_SHAPE);
  VQAF_REQUIRE(D % 8 == 0 && vqaf_aligned16(grad_seq), VQAF_E_ALIGN);
  return launch_bwd<true>(grad_out, ld_out, col0, g_dtype, idx, alpha, B, N, K, D, dtype, grad_seq,
                          (hipStream_t)stream);
}

extern "C" int vqaf_gather_mix_pool_multi(const vqaf_pool_mod* mods_host, int n_mod, int B, int D, int dtype,
                                          void* out, int ld_out, int out_dtype, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(mods_host != nullptr && out != nullptr, VQAF_E_NULL);
  VQAF_REQUIRE(n_mod > 0 && n_mod <= GM_MAX_MODS && B > 0 && B <= 65535 && D > 0, VQAF_E_SHAPE);
  VQAF_REQUIRE(vqaf_dtype_ok(dtype) && vqaf_dtype_ok(out_dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(D % 8 == 0, VQAF_E_ALIGN);
  PoolMods mods;
  size_t smem = 0;
  for (int i = 0; i < n_mod; ++i) {
    const vqaf_pool_mod& h = mods_host[i];
    VQAF_REQUIRE(h.seq != nullptr && h.idx != nullptr, VQAF_E_NULL);
    VQAF_REQUIRE(h.N > 0 && h.K > 0 && h.K <= h.N && h.N <= 16384, VQAF_E_SHAPE);
    VQAF_REQUIRE(h.col0 >= 0 && ld_out >= h.col0 + D, VQAF_E_SHAPE);
    VQAF_REQUIRE(vqaf_aligned16(h.seq) && (h.noise == nullptr || vqaf_aligned16(h.noise)), VQAF_E_ALIGN);
    VQAF_REQUIRE(h.noise_kind >= VQAF_NOISE_NONE && h.noise_kind <= VQAF_NOISE_UNIFORM, VQAF_E_SHAPE);
    VQAF_REQUIRE(h.noise != nullptr || h.noise_kind == VQAF_NOISE_NONE || h.rng_state != nullptr, VQAF_E_NULL);
    mods.m[i] = PoolMod{h.seq, h.idx, h.noise, h.rng_state, h.noise_scale, h.alpha, h.noise_kind, h.N, h.K, h.col0};
    const size_t need = fwd_smem_bytes(h.N, h.K);
    smem = need > smem ? need : smem;
  }
  const int lpr = pick_lpr(B, D);
  const dim3 grid((D + 8 * lpr - 1) / (8 * lpr), B, n_mod), block(GM_THREADS);
  hipStream_t st = (hipStream_t)stream;
  const bool bf = dtype == VQAF_BF16, bfo = out_dtype == VQAF_BF16;
#define VQAF_GMM_LAUNCH(BF, BFO, L) \
  hipLaunchKernelGGL((gather_mix_pool_multi_kernel<BF, BFO, L>), grid, block, smem, st, mods, n_mod, D, out, ld_out)
#define VQAF_GMM_BY_DT(L)                                  \
  do {                                                     \
    if (bf && bfo) VQAF_GMM_LAUNCH(true, true, L);         \
    else if (bf) VQAF_GMM_LAUNCH(true, false, L);          \
    else if (!bfo) VQAF_GMM_LAUNCH(false, false, L);       \
    else VQAF_GMM_LAUNCH(false, true, L);                  \
  } while (0)
  if (lpr == 64) VQAF_GMM_BY_DT(64);
  else if (lpr == 32) VQAF_GMM_BY_DT(32);
  else if (lpr == 16) VQAF_GMM_BY_DT(16);
  else VQAF_GMM_BY_DT(8);
#undef VQAF_GMM_BY_DT
#undef VQAF_GMM_LAUNCH
  VQAF_LAUNCH_CHECK();
}

extern "C" int vqaf_gather_mix_pool_bwd_multi(const void* grad_out, int ld_out, int g_dtype,
                                              const vqaf_pool_bwd_mod* mods_host, int n_mod, int B, int D,
                                              int dtype, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(grad_out != nullptr && mods_host != nullptr, VQAF_E_NULL);
  VQAF_REQUIRE(n_mod > 0 && n_mod <= GM_MAX_MODS && B > 0 && B <= 65535 && D > 0, VQAF_E_SHAPE);
  VQAF_REQUIRE(vqaf_dtype_ok(dtype) && vqaf_dtype_ok(g_dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(D % 8 == 0, VQAF_E_ALIGN);
  PoolBwdMods mods;
  size_t smem = 0;
  for (int i = 0; i < n_mod; ++i) {
    const vqaf_pool_bwd_mod& h = mods_host[i];
    VQAF_REQUIRE(h.idx != nullptr, VQAF_E_NULL);
    VQAF_REQUIRE(h.N > 0 && h.K > 0 && h.K <= h.N && h.N <= 16384, VQAF_E_SHAPE);
    VQAF_REQUIRE(h.col0 >= 0 && ld_out >= h.col0 + D, VQAF_E_SHAPE);
    VQAF_REQUIRE(h.grad_seq == nullptr || vqaf_aligned16(h.grad_seq), VQAF_E_ALIGN);
    mods.m[i] = PoolBwdMod{h.idx, h.grad_seq, h.alpha, h.N, h.K, h.col0};
    const size_t need = bwd_smem_bytes(h.N, h.K);
    smem = need > smem ? need : smem;
  }
  const int lpr = pick_lpr(B, D);
  const dim3 grid((D + 8 * lpr - 1) / (8 * lpr), B, n_mod), block(GM_THREADS);
  hipStream_t st = (hipStream_t)stream;
  const bool bfg = g_dtype == VQAF_BF16, bf = dtype == VQAF_BF16;
#define VQAF_GMBM_LAUNCH(BFG, BF, L)                                                              \
  hipLaunchKernelGGL((gather_mix_pool_bwd_multi_kernel<BFG, BF, L>), grid, block, smem, st, grad_out, \
                     ld_out, mods, n_mod, D)
#define VQAF_GMBM_BY_DT(L)                                 \
  do {                                                     \
    if (bfg && bf) VQAF_GMBM_LAUNCH(true, true, L);        \
    else if (bfg) VQAF_GMBM_LAUNCH(true, false, L);        \
    else if (bf) VQAF_GMBM_LAUNCH(false, true, L);         \
    else VQAF_GMBM_LAUNCH(false, false, L);                \
  } while (0)
  if (lpr == 64) VQAF_GMBM_BY_DT(64);
  else if (lpr == 32) VQAF_GMBM_BY_DT(32);
  else if (lpr == 16) VQAF_GMBM_BY_DT(16);
  else VQAF_GMBM_BY_DT(8);
#undef VQAF_GMBM_BY_DT
#undef VQAF_GMBM_LAUNCH
  VQAF_LAUNCH_CHECK();
}

extern "C" int vqaf_mean_pool(const void* x, int B, int M, int D, int dtype, void* out, int ld_out,
                              int col0, int out_dtype, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(x != nullptr && out != nullptr, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_dtype_ok(dtype) && vqaf_dtype_ok(out_dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(B > 0 && M > 0 && D > 0 && B <= 65535, VQAF_E_SHAPE);
  VQAF_REQUIRE(col0 >= 0 && ld_out >= col0 + D, VQAF_E_SHAPE);
  VQAF_REQUIRE(D % 8 == 0 && vqaf_aligned16(x), VQAF_E_ALIGN);
  const int lpr = pick_lpr(B, D) > 16 ? 16 : pick_lpr(B, D);
  const dim3 grid((D + 8 * lpr - 1) / (8 * lpr), B), block(GM_THREADS);
  hipStream_t st = (hipStream_t)stream;
#define VQAF_MP_LAUNCH(BF, BFO, L) \
  hipLaunchKernelGGL((mean_pool_kernel<BF, BFO, L>), grid, block, 0, st, x, M, D, out, ld_out, col0)
  const bool bf = dtype == VQAF_BF16, bfo = out_dtype == VQAF_BF16;
  if (lpr == 16) {
    if (bf && bfo) VQAF_MP_LAUNCH(true, true, 16);
    else if (bf) VQAF_MP_LAUNCH(true, false, 16);
    else if (bfo) VQAF_MP_LAUNCH(false, true, 16);
    else VQAF_MP_LAUNCH(false, false, 16);
  } else {
    if (bf && bfo) VQAF_MP_LAUNCH(true, true, 8);
    else if (bf) VQAF_MP_LAUNCH(true, false, 8);
    else if (bfo) VQAF_MP_LAUNCH(false, true, 8);
    else VQAF_MP_LAUNCH(false, false, 8);
  }
#undef VQAF_MP_LAUNCH
  VQAF_LAUNCH_CHECK();
}

extern "C" int vqaf_mean_pool_bwd(const void* grad_out, int ld_out, int col0, int g_dtype, int B,
                                  int M, int D, int dtype, void* grad_x, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(grad_out != nullptr && grad_x != nullptr, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_dtype_ok(dtype) && vqaf_dtype_ok(g_dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(B > 0 && M > 0 && D > 0 && B <= 65535, VQAF_E_SHAPE);
  VQAF_REQUIRE(col0 >= 0 && ld_out >= col0 + D, VQAF_E_SHAPE);
  VQAF_REQUIRE(D % 8 == 0 && vqaf_aligned16(grad_x), VQAF_E_ALIGN);
  const int lpr = pick_lpr(B, D) > 16 ? 16 : pick_lpr(B, D);
  const dim3 grid((D + 8 * lpr - 1) / (8 * lpr), B), block(GM_THREADS);
  hipStream_t st = (hipStream_t)stream;
#define VQAF_MPB_LAUNCH(BFG, BF, L)                                                            \
  hipLaunchKernelGGL((mean_pool_bwd_kernel<BFG, BF, L>), grid, block, 0, st, grad_out, ld_out, \
                     col0, M, D, grad_x)
  const bool bfg = g_dtype == VQAF_BF16, bf = dtype == VQAF_BF16;
  if (lpr == 16) {
    if (bfg && bf) VQAF_MPB_LAUNCH(true, true, 16);
    else if (bfg) VQAF_MPB_LAUNCH(true, false, 16);
    else if (bf) VQAF_MPB_LAUNCH(false, true, 16);
    else VQAF_MPB_LAUNCH(false, false, 16);
  } else {
    if (bfg && bf) VQAF_MPB_LAUNCH(true, true, 8);
    else if (bfg) VQAF_MPB_LAUNCH(true, false, 8);
    else if (bf) VQAF_MPB_LAUNCH(false, true, 8);
    else VQAF_MPB_LAUNCH(false, false, 8);
  }
#undef VQAF_MPB_LAUNCH
  VQAF_LAUNCH_CHECK();
}
