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

#ifdef OP_STAMPS  // developer build: timestamps of the selection's steps, kept in LDS until the kernel copies them out
__shared__ unsigned long long sel_stamps_lds[8];
#define SEL_STAMP(i) do { if (threadIdx.x == 0) sel_stamps_lds[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#endif
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
  c.keys = o;  o += fu_align16((size_t)(NP + 2) * 8);
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
  unsigned long long seed = 0, counter = 0;
  if (M.adv_mode == 2) {
    seed = sel_rng[0] + M.salt;
    // rng_advance < 0: the counter was advanced by -rng_advance BEFORE this launch (by the launch in front
    // of it); the draws use the position it had then
    counter = sel_rng[1] + (unsigned long long)(long long)(rng_advance < 0 ? rng_advance : 0);
  }

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
    load_half(ha, tile, 0);  // in flight while the draws are made and the operand below is staged
    load_half(hb, tile, HS);
    if (M.adv_mode == 2) select_draws(L, K, N - K, M.NPK, M.NPR, k_adv, seed, counter, (unsigned long long)b);
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
  select_core(L, H, N, K, NP, M.adv_mode, k_adv, M.NPK, M.NPR,
              M.pos ? M.pos + (size_t)b * k_adv : nullptr, M.src ? M.src + (size_t)b * k_adv : nullptr,
              seed, counter, (unsigned long long)b, M.w_out ? M.w_out + (size_t)b * N : nullptr,
              nullptr, nullptr, nullptr, true);
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
    const unsigned long long ncounter = M.noise_rng[1] + (unsigned long long)M.noise_ctr_off +
        (unsigned long long)(long long)((rng_advance < 0 && M.noise_rng == sel_rng) ? rng_advance : 0);
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
  if (rng_advance > 0)
    vqaf_rng_bump_by_last_workgroup(sel_rng, (unsigned long long)rng_advance, gridDim.x * gridDim.y);
}

// ---------------------------------------------------------------------------------------
// One-pass form for 512-wide tokens and samples of up to 208 tokens: the sample's rows are read
// from memory ONCE and stay on the compute unit between scoring and pooling.
//
// 256 threads; a wave keeps up to three 16-token tiles as raw bf16 in registers (the MFMA A-operand
// layout they were loaded in: lane = (token, 8-column group)); a thirteenth tile goes to LDS by
// LDS-DMA, row-major.  Two workgroups share a compute unit (256 registers per lane), so one
// sample's selection runs while the other sample's rows arrive; the adversarial draws, which do
// not depend on the scores, are made while the rows are in flight.
//
// Pooling needs sums over TOKENS, which lie across lanes in that layout.  A wave therefore passes
// its tiles through a small LDS image, 32 columns at a time (rows of 96 bytes: eight consecutive
// rows fall on disjoint banks), and reads them back with the transposing LDS read as the B operand
// of   acc = W (16 x 32: one row = keep multiplicity, one = dropped flag, k = token of two tiles)
//            . X (32 tokens x 16 columns),
// one MFMA per 16 columns and tile pair.  Every product is a bf16 value times 0, 1 or a small
// integer, exact in f32, and the sums accumulate in f32 as in the two-pass kernel.  A non-finite
// value in a DROPPED row makes that column's kept sum NaN as well (0 x inf), as in the two-pass kernel.
// ---------------------------------------------------------------------------------------
constexpr int OP_THREADS = 256;
constexpr int OP_WAVES = OP_THREADS / VQAF_WAVE;
constexpr int OP_D = 512;
constexpr int OP_STEPS = OP_D / 32;
constexpr int OP_REG_TILES = 3 * OP_WAVES;     // tiles held in registers
constexpr int OP_MAX_TILES = OP_REG_TILES + 1; // + one tile held in LDS (wave 0)
constexpr int OP_LROW = OP_D * 2 + 32;         // row pitch of the LDS tile (bytes): 8 banks past a multiple of 64
constexpr int OP_SROW = 96;                    // row pitch of a staging image: 32 columns + 32 bytes
constexpr int OP_STAGE_BYTES = 48 * OP_SROW;   // three tiles x 32 columns, per wave
constexpr size_t OP_LDS_LIMIT = 80 * 1024;     // two workgroups per compute unit

struct OnePassCarve {
  size_t ub, ltile, keys, sl, rank, s_pos, wl, mult, stat, misc, total;
};
__host__ __device__ inline OnePassCarve onepass_carve(int H, int N, int NP, int k_adv, bool noise) {
  OnePassCarve c;
  const int T = (N + 15) / 16;
  size_t o = 0;
  c.ub = o;  // B operand of the scoring pass, then the waves' staging images
  const size_t ubb = fu_align16((size_t)16 * (OP_D + 8) * 2), stg = (size_t)OP_WAVES * OP_STAGE_BYTES;
  o += ubb > stg ? ubb : stg;
  c.ltile = o;
  o += (size_t)(T > OP_REG_TILES ? 16 * OP_LROW : 0);
  c.keys = o;  // keys and head scores are dead after the selection: partial sums overlay them
  const size_t ks = fu_align16((size_t)(NP + 2) * 8) + fu_align16((size_t)N * H * 4);
  const size_t ps = (size_t)OP_WAVES * 2 * OP_D * 4 + (noise ? (size_t)OP_WAVES * OP_D * 4 : 0);
  c.sl = o + fu_align16((size_t)(NP + 2) * 8);
  o += ks > ps ? ks : ps;
  c.rank = o;  o += fu_align16((size_t)N * 4);
  c.s_pos = o; o += fu_align16((size_t)(k_adv > 0 ? 2 * k_adv : 1) * 4);
  c.wl = o;    o += fu_align16((size_t)N * 4);
  c.mult = o;  o += (size_t)T * 16 * 4;
  c.stat = o;  o += fu_align16((size_t)2 * SEL_MAX_H * 4);
  c.misc = o;  o += 16;
  c.total = o;
  return c;
}

__device__ __forceinline__ void op_glds16(const void* src, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

typedef unsigned int op_u32x4 __attribute__((ext_vector_type(4)));
typedef short op_s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ fu_bf16x8 op_as_frag(op_u32x4 v) {
  union { op_u32x4 u; fu_bf16x8 f; } c;
  c.u = v;
  return c.f;
}
// Transposing LDS read: the 16 lanes of a group name 4 rows x 16 columns (lane i: row i / 4, columns
// 4 * (i % 4) .. + 3) and lane i receives column i of the four rows.
__device__ __forceinline__ u32x2 op_read_tr(const unsigned char* p) {
  const op_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) op_s16x4*)(const_cast<unsigned char*>(p)));
  union { op_s16x4 s; u32x2 u; } c;
  c.s = v;
  return c.u;
}

#ifdef OP_STAMPS
#define OP_STAMP(i) do { if (tid == 0 && M.w_out) reinterpret_cast<unsigned long long*>(M.w_out)[(size_t)b * 24 + i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define OP_STAMP(i) do {} while (0)
#endif

__global__ void __launch_bounds__(OP_THREADS, 2)
fused_onepass_kernel(FusedMods mods, int n_mod, int H, uint64_t* sel_rng, void* __restrict__ out, int ld_out,
                     int out_bf16, int rng_advance) {
  // One workgroup per unit of the list [modality 0 samples, modality 1 samples, ...] (the long samples
  // are dispatched first).  (A persistent form that re-requested each register fragment for the
  // workgroup's next unit as soon as the pooling stage had staged it was built and measured: with the
  // three tiles live across the whole loop the compiler spills some two hundred registers and the
  // launch took twice as long.)
  constexpr int D = OP_D, STEPS = OP_STEPS, PITCH = D + 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char fu_lds[];
  const int b = blockIdx.x;
  const FusedMod& M = mods.m[blockIdx.y];
  const int N = M.N, K = M.K, NP = M.NP, k_adv = M.k_adv;
  const int T = (N + 15) >> 4;
  const bool noise_on = M.noise != nullptr || M.noise_kind != VQAF_NOISE_NONE;
  const OnePassCarve cv = onepass_carve(H, N, NP, k_adv, noise_on);
  bf16_t* ub = reinterpret_cast<bf16_t*>(fu_lds + cv.ub);
  float* part = reinterpret_cast<float*>(fu_lds + cv.keys);  // [wave][2][D]; overlays keys / sl after the selection
  float* nzp = part + OP_WAVES * 2 * D;                      // [wave][D] noise sums
  unsigned char* ltile = fu_lds + cv.ltile;
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
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, kg = lane >> 4;
  const op_u32x4 none = {0u, 0u, 0u, 0u};
  const bool have0 = 16 * wave < N, have1 = 16 * (wave + OP_WAVES) < N, have2 = 16 * (wave + 2 * OP_WAVES) < N;
  const bool have_l = wave == 0 && T > OP_REG_TILES;
  const bf16_t* __restrict__ seq_b = M.seq + (size_t)b * N * D;

  fu_bf16x8 r0[STEPS], r1[STEPS], r2[STEPS];
  auto load_tile = [&](fu_bf16x8 (&dst)[STEPS], int tile, bool have) {
    if (have) {
      const int row = 16 * tile + col;
      const bool real = row < N;  // rows past the end: zero (only the last tile has any)
      const bf16_t* xr = seq_b + (size_t)min(row, N - 1) * D + 8 * kg;
#pragma unroll
      for (int st = 0; st < STEPS; ++st) dst[st] = *reinterpret_cast<const fu_bf16x8*>(xr + 32 * st);
      if (16 * tile + 16 > N) {
#pragma unroll
        for (int st = 0; st < STEPS; ++st)
          if (!real) dst[st] = op_as_frag(none);
      }
    } else {
#pragma unroll
      for (int st = 0; st < STEPS; ++st) dst[st] = op_as_frag(none);
    }
  };
  {
    OP_STAMP(0);
    unsigned long long seed = 0, counter = 0;
    if (M.adv_mode == 2) {
      seed = sel_rng[0] + M.salt;
      // rng_advance < 0: the counter was advanced by -rng_advance BEFORE this launch (by the launch in front
      // of it); the draws use the position it had then
      counter = sel_rng[1] + (unsigned long long)(long long)(rng_advance < 0 ? rng_advance : 0);
    }
    // folded queries first: their data is wanted first and loads return in issue order
    f32x4 uv[4];
    {
      const float* ug = M.u + (size_t)b * H * D;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int h = (tid >> 7) + 2 * i, k4 = tid & 127;
        uv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (h < H) uv[i] = *reinterpret_cast<const f32x4*>(ug + (size_t)h * D + 4 * k4);
      }
    }
    // thirteenth tile: wave 0 brings it into LDS row by row (one whole row per LDS-DMA instruction);
    // rows past the sample are zeros
    if (have_l) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 16 * OP_REG_TILES + r;
        if (row < N) op_glds16(seq_b + (size_t)row * D + 8 * lane, ltile + r * OP_LROW);
        else *reinterpret_cast<op_u32x4*>(ltile + r * OP_LROW + 16 * lane) = none;
      }
    }
    load_tile(r0, wave, have0);
    // while the first rows are in flight: the adversarial draws (they depend on the generator state
    // only).  The other two register tiles are requested after them: the draws need registers of
    // their own, and with all three tiles live the compiler spills loaded rows (and waits for them).
    if (tid < 4) misc[tid] = 0;
    for (int n = tid; n < 16 * T; n += OP_THREADS) mult[n] = 0;
    if (M.adv_mode == 2) select_draws(L, K, N - K, M.NPK, M.NPR, k_adv, seed, counter, (unsigned long long)b);
    load_tile(r1, wave + OP_WAVES, have1);
    load_tile(r2, wave + 2 * OP_WAVES, have2);
    // B operand of the scoring pass: [bf16(u) | bf16(u - bf16(u))] rows h and h + 8
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int h = (tid >> 7) + 2 * i, k4 = tid & 127;
      const f32x4 v = uv[i];
      const bf16_t h0 = f32_to_bf16(v.x), h1 = f32_to_bf16(v.y), h2 = f32_to_bf16(v.z), h3 = f32_to_bf16(v.w);
      const u32x2 hi = {(unsigned int)h0 | ((unsigned int)h1 << 16), (unsigned int)h2 | ((unsigned int)h3 << 16)};
      const u32x2 lo = {pack_bf16x2(v.x - bf16_to_f32(h0), v.y - bf16_to_f32(h1)),
                        pack_bf16x2(v.z - bf16_to_f32(h2), v.w - bf16_to_f32(h3))};
      *reinterpret_cast<u32x2*>(&ub[h * PITCH + 4 * k4]) = hi;
      *reinterpret_cast<u32x2*>(&ub[(h + 8) * PITCH + 4 * k4]) = lo;
    }
    OP_STAMP(1);
    __syncthreads();
    OP_STAMP(2);

    // ------------------------------- head scores --------------------------------------------
    // One operand fragment serves the wave's three register tiles; each tile's products accumulate in
    // step order (the order of the separate scoring kernel: identical scores, identical ranking).
    {
      const bf16_t* ubl = &ub[col * PITCH + 8 * kg];
      auto put_scores = [&](const f32x4 acc, int tile) {
        // C/D layout: column = lane & 15, row = 4 * (lane >> 4) + r; column h + 8 holds the low part
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = acc[r] + __shfl_xor(acc[r], 8, 64);
          const int tok = 16 * tile + 4 * kg + r;
          if (col < H && tok < N) L.sl[tok * H + col] = v;
        }
      };
      if (have0) {
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0;
        fu_bf16x8 bq[4];  // operand fragments four steps ahead of the products (two waves per SIMD hide little)
#pragma unroll
        for (int st = 0; st < 4; ++st) bq[st] = *reinterpret_cast<const fu_bf16x8*>(ubl + 32 * st);
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
          const fu_bf16x8 bf = bq[st & 3];
          if (st + 4 < STEPS) bq[st & 3] = *reinterpret_cast<const fu_bf16x8*>(ubl + 32 * (st + 4));
          a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(r0[st], bf, a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(r1[st], bf, a1, 0, 0, 0);
          a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(r2[st], bf, a2, 0, 0, 0);
        }
        put_scores(a0, wave);
        if (have1) put_scores(a1, wave + OP_WAVES);
        if (have2) put_scores(a2, wave + 2 * OP_WAVES);
      }
      if (have_l) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const unsigned char* arow = ltile + col * OP_LROW + 16 * kg;
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
          const fu_bf16x8 a = *reinterpret_cast<const fu_bf16x8*>(arow + 64 * st);
          const fu_bf16x8 bf = *reinterpret_cast<const fu_bf16x8*>(ubl + 32 * st);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bf, acc, 0, 0, 0);
        }
        put_scores(acc, OP_REG_TILES);
      }
    }
    OP_STAMP(3);
    __syncthreads();
    OP_STAMP(4);

    // ------------------------------- selection ----------------------------------------------
#ifdef OP_STAMPS
    float* const w_out_b = nullptr;
#else
    float* const w_out_b = M.w_out ? M.w_out + (size_t)b * N : nullptr;
#endif
    select_core(L, H, N, K, NP, M.adv_mode, k_adv, M.NPK, M.NPR,
                M.pos ? M.pos + (size_t)b * k_adv : nullptr, M.src ? M.src + (size_t)b * k_adv : nullptr,
                seed, counter, (unsigned long long)b, w_out_b, nullptr, nullptr, nullptr, true);
    OP_STAMP(5);
    for (int j = tid; j < K; j += OP_THREADS) atomicAdd(&mult[L.rank[j]], 1);
    __syncthreads();
    {
      // rows outside the keep list, counted once per wave
      int local = 0;
      for (int n = tid; n < 16 * T; n += OP_THREADS) local += __popcll(__ballot(n < N && mult[n] == 0));
      if (lane == 0 && local) atomicAdd(&misc[0], local);
    }

    // ------------------------------- pooled sums on the matrix cores ------------------------
    // Weights of a tile pair as the A operand of a 16x16x32 product: k = 8 * kg + e is token
    // 4 * kg + (e & 3) of the pair's first (e < 4) or second (e >= 4) tile.  Lanes whose operand row m
    // has m % 4 == 0 carry the keep multiplicities, m % 4 == 1 the dropped flags, the others zeros; a
    // product then keeps the rows of ONE group of four (m / 4 == q), so that four products -- four
    // 16-column blocks -- land in the four lane groups of one accumulator.
    auto weights_of = [&](int tile_a, bool have_a, int tile_b, bool have_b) -> op_u32x4 {
      unsigned int wq[4] = {0u, 0u, 0u, 0u};
      if ((col & 3) < 2) {
        auto four = [&](int tile, bool have, unsigned int& w0, unsigned int& w1) {
          if (have) {
            const int n0 = 16 * tile + 4 * kg;
            const int4 m = *reinterpret_cast<const int4*>(&mult[n0]);
            // small integers are exact in bf16: the upper half of their f32 image
            unsigned int f[4];
            const int mm[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
              f[i] = __float_as_uint((col & 3) == 0 ? (float)mm[i] : ((mm[i] == 0 && n0 + i < N) ? 1.f : 0.f)) >> 16;
            w0 = f[0] | (f[1] << 16);
            w1 = f[2] | (f[3] << 16);
          }
        };
        four(tile_a, have_a, wq[0], wq[1]);
        four(tile_b, have_b, wq[2], wq[3]);
      }
      return op_u32x4{wq[0], wq[1], wq[2], wq[3]};
    };
    const op_u32x4 u01 = weights_of(wave, have0, wave + OP_WAVES, have1);
    const op_u32x4 u2l = weights_of(wave + 2 * OP_WAVES, have2, OP_REG_TILES, have_l);

    OP_STAMP(6);
    {
      // (the staging images overlay the scoring operand, the partial sums the selection's keys and
      // scores: every wave passed a barrier after its last use of either)
      // A wave's image holds one 32-column step of its three register tiles: rows 0-15, 16-31, 32-47,
      // 96 bytes apart (eight consecutive rows fall on disjoint banks).
      unsigned char* stg = fu_lds + cv.ub + (size_t)wave * OP_STAGE_BYTES;
      float* mypart = part + (size_t)wave * 2 * D;
      unsigned char* wr = stg + col * OP_SROW + 16 * kg;  // this lane's 16 bytes of row `col`
      const unsigned char* rd = stg + (4 * kg + (col >> 2)) * OP_SROW + 8 * (col & 3);
      const unsigned char* rdl = ltile + (4 * kg + (col >> 2)) * OP_LROW + 8 * (col & 3);
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      const int mq = col >> 2;
#pragma unroll
      for (int sp = 0; sp < STEPS / 2; ++sp) {  // two steps = 64 columns = four blocks per accumulator
        f32x4 acc = zero;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const int st = 2 * sp + s2;
          if (have0) {
            *reinterpret_cast<fu_bf16x8*>(wr) = r0[st];
            *reinterpret_cast<fu_bf16x8*>(wr + 16 * OP_SROW) = r1[st];
            *reinterpret_cast<fu_bf16x8*>(wr + 32 * OP_SROW) = r2[st];
          }
          if (have0) {
            u32x2 a_lo[2], a_hi[2], b_lo[2], b_hi[2];
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
              a_lo[blk] = op_read_tr(rd + 32 * blk);
              a_hi[blk] = op_read_tr(rd + 16 * OP_SROW + 32 * blk);
              b_lo[blk] = op_read_tr(rd + 32 * OP_SROW + 32 * blk);
              b_hi[blk] = u32x2{0u, 0u};
              if (have_l) b_hi[blk] = op_read_tr(rdl + 64 * st + 32 * blk);
            }
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
              const int q = 2 * s2 + blk;
              const op_u32x4 wa = mq == q ? u01 : none, wb = mq == q ? u2l : none;
              acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                  op_as_frag(wa), op_as_frag(op_u32x4{a_lo[blk].x, a_lo[blk].y, a_hi[blk].x, a_hi[blk].y}), acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                  op_as_frag(wb), op_as_frag(op_u32x4{b_lo[blk].x, b_lo[blk].y, b_hi[blk].x, b_hi[blk].y}), acc, 0, 0, 0);
            }
          }
        }
        // lane group kg holds block kg of these 64 columns: register 0 = kept sum, 1 = dropped sum
        mypart[64 * sp + 16 * kg + col] = acc[0];
        mypart[D + 64 * sp + 16 * kg + col] = acc[1];
      }
    }
    OP_STAMP(7);
    for (int j = tid; j < K; j += OP_THREADS) M.idx_out[(size_t)b * K + j] = (int64_t)L.rank[j];
    // noise: one draw per keep slot and element, addressed exactly as the unfused kernels address it;
    // lane owns 8 columns, wave g takes slots g, g + 4, ...
    if (noise_on) {
      const int g = wave, d0 = lane * 8;
      float acck[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) acck[i] = 0.f;
      if (M.noise != nullptr) {
        for (int j = g; j < K; j += OP_WAVES) {
          const f32x4* p = reinterpret_cast<const f32x4*>(M.noise + ((size_t)b * K + j) * D + d0);
          const f32x4 a = p[0], c = p[1];
          acck[0] += a.x; acck[1] += a.y; acck[2] += a.z; acck[3] += a.w;
          acck[4] += c.x; acck[5] += c.y; acck[6] += c.z; acck[7] += c.w;
        }
      } else {
        const unsigned long long nseed = M.noise_rng[0];
        const unsigned long long ncounter = M.noise_rng[1] + (unsigned long long)M.noise_ctr_off +
        (unsigned long long)(long long)((rng_advance < 0 && M.noise_rng == sel_rng) ? rng_advance : 0);
        for (int j = g; j < K; j += OP_WAVES)
          vqaf_token_noise8(acck, nullptr, M.noise_kind, M.noise_scale, nseed, ncounter, ((size_t)b * K + j) * D + d0);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) nzp[g * D + d0 + i] = acck[i];
    }
    OP_STAMP(8);
    __syncthreads();
    const int R = misc[0];
    const bool with_mix = M.alpha > 0.f;
    const int Kout = K + (with_mix ? 1 : 0);
    const float cnt = (float)R + 1e-10f;
    for (int c = tid; c < D; c += OP_THREADS) {
      float sk = 0.f, sd = 0.f;
#pragma unroll
      for (int w = 0; w < OP_WAVES; ++w) {
        sk += part[w * 2 * D + c];
        sd += part[w * 2 * D + D + c];
      }
      if (noise_on) {
        float nz = 0.f;
#pragma unroll
        for (int w = 0; w < OP_WAVES; ++w) nz += nzp[w * D + c];
        sk += nz;
      }
      float tot = sk;
      if (with_mix) tot += M.alpha * (sd / cnt);  // remaining_sum / (count + 1e-10), then * alpha
      const float r = tot / (float)Kout;
      const size_t o = (size_t)b * ld_out + M.col0 + c;
      if (out_bf16) reinterpret_cast<bf16_t*>(out)[o] = f32_to_bf16(r);
      else reinterpret_cast<float*>(out)[o] = r;
    }
    OP_STAMP(9);
#ifdef OP_STAMPS
    if (tid == 0 && M.w_out)
      for (int i = 0; i < 6; ++i) reinterpret_cast<unsigned long long*>(M.w_out)[(size_t)b * 24 + 10 + i] = sel_stamps_lds[i];
#endif
  }
  if (rng_advance > 0)
    vqaf_rng_bump_by_last_workgroup(sel_rng, (unsigned long long)rng_advance, gridDim.x * gridDim.y);
}

}  // namespace

static int fused_prune_pool_launch(const vqaf_fused_mod* mods_host, int n_mod, int B, int D, int H,
                                   int dtype, uint64_t* sel_rng_state, int rng_advance, void* out,
                                   int ld_out, int out_dtype, vqaf_stream_t stream, bool allow_one_pass) {
  VQAF_ENTRY();
  VQAF_REQUIRE(mods_host != nullptr && out != nullptr, VQAF_E_NULL);
  VQAF_REQUIRE(rng_advance == 0 || sel_rng_state != nullptr, VQAF_E_NULL);
  VQAF_REQUIRE(dtype == VQAF_BF16 && vqaf_dtype_ok(out_dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(n_mod > 0 && n_mod <= FU_MAX_MODS && B > 0 && H > 0 && H <= 8, VQAF_E_SHAPE);
  VQAF_REQUIRE(D == 64 || D == 128 || D == 256 || D == 512 || D == 1024, VQAF_E_SHAPE);
  FusedMods mods = {};
  size_t smem = 0;
  bool one_pass = allow_one_pass && D == OP_D, noise_any = false;
  int n_max = 1, kadv_max = 0;
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
    noise_any = noise_any || h.noise != nullptr || h.noise_kind != VQAF_NOISE_NONE;
    n_max = h.N > n_max ? h.N : n_max;
    kadv_max = k_adv > kadv_max ? k_adv : kadv_max;
    if (h.N > 16 * OP_MAX_TILES) one_pass = false;
  }
  hipStream_t st = (hipStream_t)stream;
  const int obf = out_dtype == VQAF_BF16 ? 1 : 0;
  // one LDS layout for every unit a persistent workgroup takes: sized for the longest modality
  const OnePassCarve cv_one = onepass_carve(H, n_max, sel_next_pow2(n_max), kadv_max, noise_any);
  if (one_pass && cv_one.total <= OP_LDS_LIMIT) {
    // more than 64 KiB of dynamic LDS needs the attribute once per device; setting it again is cheap
    if (cv_one.total > 64 * 1024)
      VQAF_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(fused_onepass_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)OP_LDS_LIMIT) == hipSuccess,
                   VQAF_E_LAUNCH);
    int cus = 256, dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    (void)cus;
    hipLaunchKernelGGL(fused_onepass_kernel, dim3(B, n_mod), dim3(OP_THREADS), cv_one.total, st, mods, n_mod, H,
                       sel_rng_state, out, ld_out, obf, rng_advance);
    VQAF_LAUNCH_CHECK();
  }
  VQAF_REQUIRE(smem <= 64 * 1024, VQAF_E_SHAPE);
  const dim3 grid(B, n_mod), block(FU_THREADS);
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

extern "C" int vqaf_fused_prune_pool_multi(const vqaf_fused_mod* mods_host, int n_mod, int B, int D, int H,
                                           int dtype, uint64_t* sel_rng_state, int rng_advance, void* out,
                                           int ld_out, int out_dtype, vqaf_stream_t stream) {
  return fused_prune_pool_launch(mods_host, n_mod, B, D, H, dtype, sel_rng_state, rng_advance, out, ld_out,
                                 out_dtype, stream, false);
}

extern "C" int vqaf_fused_prune_pool_onepass_supported(int n_mod, const int* N, int D, int H) {
  if (n_mod <= 0 || n_mod > FU_MAX_MODS || N == nullptr || D != OP_D || H <= 0 || H > 8) return 0;
  for (int i = 0; i < n_mod; ++i)
    if (N[i] <= 0 || N[i] > 16 * OP_MAX_TILES) return 0;
  return 1;
}

extern "C" int vqaf_fused_prune_pool_onepass_multi(const vqaf_fused_mod* mods_host, int n_mod, int B, int D,
                                                   int H, int dtype, uint64_t* sel_rng_state, int rng_advance,
                                                   void* out, int ld_out, int out_dtype, vqaf_stream_t stream) {
  if (mods_host != nullptr && n_mod > 0 && n_mod <= FU_MAX_MODS) {
    int n[FU_MAX_MODS];
    for (int i = 0; i < n_mod; ++i) n[i] = mods_host[i].N;
    if (!vqaf_fused_prune_pool_onepass_supported(n_mod, n, D, H)) return VQAF_E_SHAPE;
  }
  return fused_prune_pool_launch(mods_host, n_mod, B, D, H, dtype, sel_rng_state, rng_advance, out, ld_out,
                                 out_dtype, stream, true);
}
