// Keep-index selection building blocks shared by the selection kernels and the fused
// score -> select -> pool kernel: order-preserving keys, in-LDS ranking, random k-prefixes of a
// permutation, and the whole "weights -> ranked keep list -> adversarial swap" sequence on LDS arrays.
#pragma once
#include "common.h"

constexpr int TK_MAX_N = 4096;
constexpr int SEL_MAX_H = 16;

__device__ __forceinline__ unsigned int order_key(float w) {
  unsigned int u = __float_as_uint(w);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0xffffffffu;  // NaN ranks above +inf
  if (u == 0x80000000u) u = 0u;                             // -0 ties with +0
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// Bitonic network through LDS, one barrier per step: any NP, any block size.
__device__ inline void bitonic_sort_desc_lds(unsigned long long* keys, int NP) {
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int k = 2; k <= NP; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int p = tid; p < (NP >> 1); p += nt) {
        const int t = ((p & ~(j - 1)) << 1) | (p & (j - 1));
        const int q = t | j;
        const unsigned long long a = keys[t], c = keys[q];
        const bool desc = (t & k) == 0;
        if (desc ? (a < c) : (a > c)) {
          keys[t] = c;
          keys[q] = a;
        }
      }
      __syncthreads();
    }
  }
}

// The same network with the keys in registers.  Thread t holds elements t*E .. t*E+E-1, so a step
// of distance j is an exchange between registers (j < E), between lanes of a wave (j < 64*E: DPP
// moves within a row of 16 lanes, a cross-lane permute beyond) or between waves (through LDS, the
// only steps that need a barrier: 3 of the 55 for 1024 keys on 256 threads).  A step through LDS
// costs a wave two dependent LDS round trips; the register steps cost a handful of ALU
// instructions, which is what makes ranking 577 weights take microseconds rather than tens.
template <int CTRL>
__device__ __forceinline__ unsigned int sel_dpp_all(unsigned int src) {  // every lane receives a value
  return (unsigned int)__builtin_amdgcn_mov_dpp((int)src, CTRL, 0xF, 0xF, false);
}
template <int CTRL, int BANK>
__device__ __forceinline__ unsigned int sel_dpp_banks(unsigned int old, unsigned int src) {
  return (unsigned int)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, 0xF, BANK, false);
}
template <int D>
__device__ __forceinline__ unsigned int sel_lane_xor32(unsigned int x) {
  static_assert(D == 1 || D == 2 || D == 4 || D == 8, "DPP reaches lanes of the same row of 16");
  if constexpr (D == 1) {
    return sel_dpp_all<0xB1>(x);  // quad_perm [1,0,3,2]
  } else if constexpr (D == 2) {
    return sel_dpp_all<0x4E>(x);  // quad_perm [2,3,0,1]
  } else if constexpr (D == 4) {
    const unsigned int t = sel_dpp_banks<0x104, 0x5>(x, x);  // row_shl:4 into banks 0,2 (lane <- lane + 4)
    return sel_dpp_banks<0x114, 0xA>(t, x);                  // row_shr:4 into banks 1,3 (lane <- lane - 4)
  } else {
    return sel_dpp_all<0x128>(x);  // row_ror:8
  }
}
// One step between lanes D apart.  take_max: this lane keeps the larger key of the pair.
template <int E, int D>
__device__ __forceinline__ void sel_lane_step(unsigned long long (&v)[E], bool take_max) {
#pragma unroll
  for (int r = 0; r < E; ++r) {
    if constexpr (D < 16) {
      const unsigned int lo = sel_lane_xor32<D>((unsigned int)v[r]), hi = sel_lane_xor32<D>((unsigned int)(v[r] >> 32));
      const unsigned long long o = ((unsigned long long)hi << 32) | lo;
      v[r] = ((o > v[r]) == take_max) ? o : v[r];
    } else {
      // Rows of 16 lanes (D = 16) or wave halves (D = 32) trade places between two copies of the
      // key: afterwards the first copy holds the pair's member from the even row (lower half) and
      // the second the one from the odd row (upper half), on both lanes of the pair; each lane then
      // picks the larger or the smaller of the two.
      const unsigned int xl = (unsigned int)v[r], xh = (unsigned int)(v[r] >> 32);
      unsigned int al, bl, ah, bh;
      if constexpr (D == 16) {
        const auto l = __builtin_amdgcn_permlane16_swap(xl, xl, false, false);
        const auto h = __builtin_amdgcn_permlane16_swap(xh, xh, false, false);
        al = l[0]; bl = l[1]; ah = h[0]; bh = h[1];
      } else {
        static_assert(D == 32, "lane distance");
        const auto l = __builtin_amdgcn_permlane32_swap(xl, xl, false, false);
        const auto h = __builtin_amdgcn_permlane32_swap(xh, xh, false, false);
        al = l[0]; bl = l[1]; ah = h[0]; bh = h[1];
      }
      const unsigned long long a = ((unsigned long long)ah << 32) | al, c = ((unsigned long long)bh << 32) | bl;
      v[r] = ((a > c) == take_max) ? a : c;
    }
  }
}

template <int E>
__device__ inline void bitonic_sort_desc_regs(unsigned long long* keys, int NP) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int e0 = tid * E;
  const bool mine = e0 < NP;  // NP is a multiple of E; threads beyond the keys carry padding
  unsigned long long v[E];
#pragma unroll
  for (int r = 0; r < E; ++r) v[r] = mine ? keys[e0 + r] : 0ull;
  for (int k = 2; k <= NP; k <<= 1) {
    const bool desc = (e0 & k) == 0;  // for k >= E; the register steps below handle k < E themselves
    for (int j = k >> 1; j >= 64 * E; j >>= 1) {  // partner in another wave
      __syncthreads();
      if (mine) {
#pragma unroll
        for (int r = 0; r < E; ++r) keys[e0 + r] = v[r];
      }
      __syncthreads();
      if (mine) {
        const bool take_max = ((e0 & j) == 0) == desc;
#pragma unroll
        for (int r = 0; r < E; ++r) {
          const unsigned long long o = keys[(e0 + r) ^ j];
          v[r] = ((o > v[r]) == take_max) ? o : v[r];
        }
      }
    }
    if (k >= 64 * E) sel_lane_step<E, 32>(v, ((lane & 32) == 0) == desc);
    if (k >= 32 * E) sel_lane_step<E, 16>(v, ((lane & 16) == 0) == desc);
    if (k >= 16 * E) sel_lane_step<E, 8>(v, ((lane & 8) == 0) == desc);
    if (k >= 8 * E) sel_lane_step<E, 4>(v, ((lane & 4) == 0) == desc);
    if (k >= 4 * E) sel_lane_step<E, 2>(v, ((lane & 2) == 0) == desc);
    if (k >= 2 * E) sel_lane_step<E, 1>(v, ((lane & 1) == 0) == desc);
#pragma unroll
    for (int j = E >> 1; j >= 1; j >>= 1) {
      if (k >= 2 * j) {
#pragma unroll
        for (int r = 0; r < E; ++r) {
          if ((r & j) == 0) {
            const bool d = ((e0 + r) & k) == 0;
            const unsigned long long a = v[r], c = v[r | j];
            const bool swap = d ? (a < c) : (a > c);
            v[r] = swap ? c : a;
            v[r | j] = swap ? a : c;
          }
        }
      }
    }
  }
  __syncthreads();  // cross-wave partner reads of the last pass are done
  if (mine) {
#pragma unroll
    for (int r = 0; r < E; ++r) keys[e0 + r] = v[r];
  }
  __syncthreads();
}

// Sorts keys[0..NP) descending; NP a power of two; all threads of the block take part; the keys
// are visible to the block on entry (a barrier has been passed) and again on return.
__device__ inline void bitonic_sort_desc(unsigned long long* keys, int NP) {
  const int nt = blockDim.x;
  const int E = NP > nt ? NP / nt : 1;
  if ((nt & 63) != 0 || E * nt < NP) return bitonic_sort_desc_lds(keys, NP);
  switch (E) {
    case 1: return bitonic_sort_desc_regs<1>(keys, NP);
    case 2: return bitonic_sort_desc_regs<2>(keys, NP);
    case 4: return bitonic_sort_desc_regs<4>(keys, NP);
    case 8: return bitonic_sort_desc_regs<8>(keys, NP);
    case 16: return bitonic_sort_desc_regs<16>(keys, NP);
    default: return bitonic_sort_desc_lds(keys, NP);
  }
}

// Ranking by counting: rank_i = #{j : key_j > key_i}.  Keys are distinct, so the ranks are a
// permutation.  Every thread streams the whole key array from LDS with broadcast 16-byte reads
// (all lanes read the same address: one LDS cycle for two keys) and needs no barrier, which
// makes one sample's ranking quicker than the 21-36 barrier-separated steps of the bitonic network
// for rows of up to 256 weights (measured: N = 30: 4.0 vs 5.2 us, N = 197: 7.3 vs 8.3 us per
// launch); its N^2 work loses beyond that (N = 577: 24 vs 14 us), where the network is kept.
constexpr int RANK_COUNT_MAX_N = 256;
// `keys` is 16-byte aligned; M keys are compared (no padding is read).
__device__ __forceinline__ int count_greater(const unsigned long long* keys, int M, unsigned long long ki) {
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  int rank = 0;
  const int even = M & ~1;
#pragma unroll 8
  for (int j = 0; j < even; j += 2) {
    const u64x2 kk = *reinterpret_cast<const u64x2*>(keys + j);
    rank += (kk.x > ki) ? 1 : 0;
    rank += (kk.y > ki) ? 1 : 0;
  }
  if (M & 1) rank += (keys[even] > ki) ? 1 : 0;
  return rank;
}

// Wave-wide reductions whose result reaches every lane with the same bits: exchanges between
// lanes 1, 2, 4, 8 apart by DPP, between rows of 16 lanes and wave halves by the swap
// instructions; the order of the additions depends on the lane distance only.
__device__ __forceinline__ float sel_readlane(float v, int l) {
  return __uint_as_float((unsigned int)__builtin_amdgcn_readlane((int)__float_as_uint(v), l));
}
__device__ __forceinline__ float sel_wave_allmax(float v) {
  v = fmaxf(v, __uint_as_float(sel_lane_xor32<1>(__float_as_uint(v))));
  v = fmaxf(v, __uint_as_float(sel_lane_xor32<2>(__float_as_uint(v))));
  v = fmaxf(v, __uint_as_float(sel_lane_xor32<4>(__float_as_uint(v))));
  v = fmaxf(v, __uint_as_float(sel_lane_xor32<8>(__float_as_uint(v))));
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const auto c = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(c[0]), __uint_as_float(c[1]));
}
__device__ __forceinline__ float sel_wave_allsum(float v) {
  v += __uint_as_float(sel_lane_xor32<1>(__float_as_uint(v)));
  v += __uint_as_float(sel_lane_xor32<2>(__float_as_uint(v)));
  v += __uint_as_float(sel_lane_xor32<4>(__float_as_uint(v)));
  v += __uint_as_float(sel_lane_xor32<8>(__float_as_uint(v)));
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);  // (even row) + (odd row) on both lanes
  const auto c = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(c[0]) + __uint_as_float(c[1]);
}

// Random k_adv-prefix of a uniform permutation of [0,M): rank M random 51-bit keys (ties are
// broken by position, probability ~ M^2 / 2^52) and keep the first k_adv positions.
__device__ inline void random_prefix(unsigned long long* keys, int M, int NP, unsigned long long seed,
                                     unsigned long long counter, unsigned long long stream_id) {
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int i = tid; i < NP; i += nt) {
    unsigned long long key = 0ull;
    if (i < M) {
      const Philox4 r = philox4x32_10(seed, stream_id * (unsigned long long)TK_MAX_N + i, counter);
      const unsigned long long r51 = (((unsigned long long)r.x << 32) | r.y) >> 13;
      key = (1ull << 63) | (r51 << 12) | (unsigned long long)(TK_MAX_N - 1 - i);  // above padding
    }
    keys[i] = key;
  }
  __syncthreads();
  bitonic_sort_desc(keys, NP);
}

// Same draw as random_prefix, delivered as positions: out[r] = position with rank r, r < take.
// Uses counting when M is small (identical result: the ranking of the same keys).
__device__ inline void random_prefix_positions(unsigned long long* keys, int M, int NP, unsigned long long seed,
                                               unsigned long long counter, unsigned long long stream_id,
                                               int take, int* out) {
  const int tid = threadIdx.x, nt = blockDim.x;
  if (M > RANK_COUNT_MAX_N) {
    random_prefix(keys, M, NP, seed, counter, stream_id);
    for (int j = tid; j < take; j += nt) out[j] = TK_MAX_N - 1 - (int)(keys[j] & (TK_MAX_N - 1));
    __syncthreads();
    return;
  }
  for (int i = tid; i < NP; i += nt) {
    unsigned long long key = 0ull;
    if (i < M) {
      const Philox4 r = philox4x32_10(seed, stream_id * (unsigned long long)TK_MAX_N + i, counter);
      const unsigned long long r51 = (((unsigned long long)r.x << 32) | r.y) >> 13;
      key = (1ull << 63) | (r51 << 12) | (unsigned long long)(TK_MAX_N - 1 - i);
    }
    keys[i] = key;
  }
  __syncthreads();
  for (int i = tid; i < M; i += nt) {
    const int r = count_greater(keys, M, keys[i]);
    if (r < take) out[r] = i;
  }
  __syncthreads();
}

__device__ __forceinline__ unsigned long long random_prefix_key(unsigned long long seed, unsigned long long counter,
                                                                unsigned long long stream_id, int i) {
  const Philox4 r = philox4x32_10(seed, stream_id * (unsigned long long)TK_MAX_N + i, counter);
  const unsigned long long r51 = (((unsigned long long)r.x << 32) | r.y) >> 13;
  return (1ull << 63) | (r51 << 12) | (unsigned long long)(TK_MAX_N - 1 - i);
}

// The same reductions over HH independent values at once (instruction-level parallelism: with few
// waves per SIMD a single dependent chain of cross-lane steps is all latency).  Per value the
// operations and their order are exactly those of sel_wave_allmax / sel_wave_allsum.
template <int HH, bool IS_MAX>
__device__ __forceinline__ void sel_wave_allreduce_n(float (&v)[HH]) {
  auto op = [](float a, float b) { return IS_MAX ? fmaxf(a, b) : a + b; };
#pragma unroll
  for (int h = 0; h < HH; ++h) v[h] = op(v[h], __uint_as_float(sel_lane_xor32<1>(__float_as_uint(v[h]))));
#pragma unroll
  for (int h = 0; h < HH; ++h) v[h] = op(v[h], __uint_as_float(sel_lane_xor32<2>(__float_as_uint(v[h]))));
#pragma unroll
  for (int h = 0; h < HH; ++h) v[h] = op(v[h], __uint_as_float(sel_lane_xor32<4>(__float_as_uint(v[h]))));
#pragma unroll
  for (int h = 0; h < HH; ++h) v[h] = op(v[h], __uint_as_float(sel_lane_xor32<8>(__float_as_uint(v[h]))));
#pragma unroll
  for (int h = 0; h < HH; ++h) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[h]), __float_as_uint(v[h]), false, false);
    v[h] = op(__uint_as_float(a[0]), __uint_as_float(a[1]));
  }
#pragma unroll
  for (int h = 0; h < HH; ++h) {
    const auto c = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[h]), __float_as_uint(v[h]), false, false);
    v[h] = op(__uint_as_float(c[0]), __uint_as_float(c[1]));
  }
}

#ifndef SEL_STAMP
#define SEL_STAMP(i) do {} while (0)
#endif

// LDS arrays of one sample's selection.
struct SelectLds {
  unsigned long long* keys;  // [NP]
  float* sl;                 // [N*H] head scores (16-byte aligned), unused when the weights are given
  int* rank;                 // [N]   full ranking; the keep list is rank[0..K) on return
  int* s_pos;                // [2*k_adv] drawn positions, then drawn sources
  float* wl;                 // [N]   head-averaged weights
  float* stat;               // [2*SEL_MAX_H]
};

// The two adversarial draws of one sample (adv_mode 2): a random k_adv-prefix of a permutation of the
// K keep-list positions -> L.s_pos[0..k_adv), and of the R bottom-list positions ->
// L.s_pos[k_adv..2*k_adv).  They depend on (seed, counter, stream_b) and the list lengths only, not
// on the weights, so a kernel may call this before its scores exist (while its rows are still in
// flight) and pass draws_done = true to select_core.  Uses L.keys (NP + 2 entries) as scratch.
// When both lists are short enough for the counting rank, one half of the workgroup draws the
// positions while the other draws the sources.  All threads call it; ends with a barrier.
__device__ inline void select_draws(const SelectLds& L, int K, int R, int NPK, int NPR, int k_adv,
                                    unsigned long long seed, unsigned long long counter,
                                    unsigned long long stream_b) {
  const int tid = threadIdx.x, nt = blockDim.x;
  int* s_src = L.s_pos + k_adv;
  if (K <= RANK_COUNT_MAX_N && R <= RANK_COUNT_MAX_N && (nt & 127) == 0) {
    const int hn = nt >> 1;
    const int which = tid >= hn ? 1 : 0, t = tid - which * hn;
    const int M = which ? R : K;
    unsigned long long* keys = L.keys + (which ? ((K + 1) & ~1) : 0);
    int* out = which ? s_src : L.s_pos;
    const unsigned long long sid = 2ull * stream_b + (unsigned long long)which;
    for (int i = t; i < M; i += hn) keys[i] = random_prefix_key(seed, counter, sid, i);
    __syncthreads();
    for (int i = t; i < M; i += hn) {
      const int r = count_greater(keys, M, keys[i]);
      if (r < k_adv) out[r] = i;
    }
    __syncthreads();
  } else {
    random_prefix_positions(L.keys, K, NPK, seed, counter, 2ull * stream_b, k_adv, L.s_pos);
    random_prefix_positions(L.keys, R, NPR, seed, counter, 2ull * stream_b + 1ull, k_adv, s_src);
  }
}

// Whole keep-index selection of one sample by one workgroup, on LDS arrays:
//   (H > 0) softmax per head over sl -> head mean -> wl;   keys -> ranking -> rank[];
//   bottom list reported (before the swap, as the reference keeps it);
//   adversarial swap with injected (adv_mode 1) or in-kernel (adv_mode 2) draws.
// On entry sl (H > 0) or wl (H == 0) is filled and visible to the workgroup; on return rank[0..K)
// is the keep list and all threads have passed a barrier.  pos_b / src_b / the *_out_b pointers are
// this sample's rows (already offset by the sample index).  draws_done: select_draws has run.
//
// The softmax statistics are formed per chunk of 64 tokens (one wave-wide reduction each) and the
// chunks are then combined in index order, so the weights do not depend on the workgroup's size.
__device__ inline void select_core(const SelectLds& L, int H, int N, int K, int NP, int adv_mode, int k_adv,
                                   int NPK, int NPR, const int64_t* pos_b, const int64_t* src_b,
                                   unsigned long long seed, unsigned long long counter, unsigned long long stream_b,
                                   float* w_out_b, int64_t* bottom_b, int64_t* pos_out_b, int64_t* src_out_b,
                                   bool draws_done = false) {
  const int tid = threadIdx.x, nt = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
  const int R = N - K;
  if (H == 8 && ((N + 63) >> 6) <= nw && (reinterpret_cast<size_t>(L.sl) & 15) == 0) {
    // Eight heads, at most one chunk per wave: thread n owns token n, its eight scores stay in
    // registers, the eight reductions run side by side.  Same arithmetic, value for value, as the
    // general form below.
    constexpr int HH = 8;
    const int nchunks = (N + 63) >> 6;
    float* cm = reinterpret_cast<float*>(L.keys);
    float* cz = cm + nchunks * HH;
    const bool mine = wave < nchunks, real = tid < N;
    float v[HH], e[HH];
    if (mine) {
      if (real) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(L.sl + tid * HH), c = *reinterpret_cast<const f32x4*>(L.sl + tid * HH + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
      } else {
#pragma unroll
        for (int h = 0; h < HH; ++h) v[h] = -INFINITY;
      }
      float mx[HH];
#pragma unroll
      for (int h = 0; h < HH; ++h) mx[h] = v[h];
      sel_wave_allreduce_n<HH, true>(mx);
      float pick = mx[0];
#pragma unroll
      for (int h = 1; h < HH; ++h) pick = lane == h ? mx[h] : pick;
      if (lane < HH) cm[wave * HH + lane] = pick;
    }
    __syncthreads();
    float mv = -INFINITY;
    if (lane < HH)
      for (int c = 0; c < nchunks; ++c) mv = fmaxf(mv, cm[c * HH + lane]);
    if (mine) {
      float zs[HH];
#pragma unroll
      for (int h = 0; h < HH; ++h) {
        e[h] = real ? expf(v[h] - sel_readlane(mv, h)) : 0.f;
        zs[h] = e[h];
      }
      sel_wave_allreduce_n<HH, false>(zs);
      float pick = zs[0];
#pragma unroll
      for (int h = 1; h < HH; ++h) pick = lane == h ? zs[h] : pick;
      if (lane < HH) cz[wave * HH + lane] = pick;
    }
    SEL_STAMP(0);
    __syncthreads();
    float zv = 0.f;
    if (lane < HH)
      for (int c = 0; c < nchunks; ++c) zv += cz[c * HH + lane];
    if (mine && real) {
      float a = 0.f;
#pragma unroll
      for (int h = 0; h < HH; ++h) a += e[h] / sel_readlane(zv, h);
      L.wl[tid] = a / (float)HH;
    }
    SEL_STAMP(1);
    __syncthreads();
  } else if (H > 0) {
    const int nchunks = (N + 63) >> 6;
    float* cm = reinterpret_cast<float*>(L.keys);  // [nchunks][H] chunk maxima (the keys come later)
    float* cz = cm + nchunks * H;                  // [nchunks][H] chunk sums
    for (int c = wave; c < nchunks; c += nw) {
      const int n = 64 * c + lane;
      for (int h = 0; h < H; ++h) {
        const float v = sel_wave_allmax(n < N ? L.sl[n * H + h] : -INFINITY);
        if (lane == h) cm[c * H + h] = v;
      }
    }
    __syncthreads();
    float mv = -INFINITY;  // lane h: maximum of head h
    if (lane < H)
      for (int c = 0; c < nchunks; ++c) mv = fmaxf(mv, cm[c * H + lane]);
    for (int c = wave; c < nchunks; c += nw) {
      const int n = 64 * c + lane;
      for (int h = 0; h < H; ++h) {
        const float m = sel_readlane(mv, h);
        const float z = sel_wave_allsum(n < N ? expf(L.sl[n * H + h] - m) : 0.f);
        if (lane == h) cz[c * H + h] = z;
      }
    }
    SEL_STAMP(0);
    __syncthreads();
    float zv = 0.f;  // lane h: sum of head h, chunks added in index order
    if (lane < H)
      for (int c = 0; c < nchunks; ++c) zv += cz[c * H + lane];
    for (int n = tid; n < N; n += nt) {
      float a = 0.f;
      for (int h = 0; h < H; ++h)
        a += expf(L.sl[n * H + h] - sel_readlane(mv, h)) / sel_readlane(zv, h);
      L.wl[n] = a / (float)H;
    }
    SEL_STAMP(1);
    __syncthreads();
  }
  if (w_out_b != nullptr)
    for (int n = tid; n < N; n += nt) w_out_b[n] = L.wl[n];
  for (int i = tid; i < NP; i += nt)
    L.keys[i] = (i < N) ? (((unsigned long long)order_key(L.wl[i]) << 32) |
                           (unsigned long long)(0xffffffffu - (unsigned int)i))
                        : 0ull;
  __syncthreads();
  SEL_STAMP(2);
  if (N <= RANK_COUNT_MAX_N) {
    for (int i = tid; i < N; i += nt) L.rank[count_greater(L.keys, N, L.keys[i])] = i;
  } else {
    bitonic_sort_desc(L.keys, NP);
    for (int r = tid; r < N; r += nt) L.rank[r] = (int)(0xffffffffu - (unsigned int)L.keys[r]);
  }
  SEL_STAMP(3);
  __syncthreads();
  if (bottom_b != nullptr)
    for (int i = tid; i < R; i += nt) bottom_b[i] = (int64_t)L.rank[N - 1 - i];

  if (adv_mode == 1) {
    for (int j = tid; j < k_adv; j += nt) {
      const int64_t p = pos_b[j], q = src_b[j];
      if (p >= 0 && p < K && q >= 0 && q < R) L.rank[p] = L.rank[N - 1 - (int)q];
    }
    __syncthreads();
  } else if (adv_mode == 2) {
    int* s_src = L.s_pos + k_adv;
    if (!draws_done) select_draws(L, K, R, NPK, NPR, k_adv, seed, counter, stream_b);
    SEL_STAMP(5);
    for (int j = tid; j < k_adv; j += nt) {
      const int p = L.s_pos[j];
      const int q = s_src[j];
      L.rank[p] = L.rank[N - 1 - q];  // positions distinct, sources from the untouched tail: no hazard
      if (pos_out_b != nullptr) pos_out_b[j] = p;
      if (src_out_b != nullptr) src_out_b[j] = q;
    }
    __syncthreads();
  }
}

inline int sel_next_pow2(int n) {
  int p = 64;
  while (p < n) p <<= 1;
  return p;
}
