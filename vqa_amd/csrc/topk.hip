// Keep-index selection: full per-row ranking by a bitonic network in LDS, adversarial index
// mixing, and the device-side RNG counter.
//
// One workgroup ranks one row.  Every element becomes a 64-bit key
//     (order-preserving image of the f32 weight) << 32 | (0xFFFFFFFF - index)
// so keys are distinct and a descending sort yields (weight desc, index asc); rows are padded
// to a power of two with key 0, which sorts below every real key.  The sorted prefix is the
// keep list, the reversed tail the "bottom" list, so the two always partition [0,N).
#include "select.h"

namespace {

__global__ void topk_kernel(const float* __restrict__ w, int N, int K, int NP,
                            int64_t* __restrict__ top, int64_t* __restrict__ bottom) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const float* wr = w + (size_t)b * N;
  for (int i = tid; i < NP; i += nt)
    keys[i] = (i < N) ? (((unsigned long long)order_key(wr[i]) << 32) |
                         (unsigned long long)(0xffffffffu - (unsigned int)i))
                      : 0ull;
  __syncthreads();
  const int R = N - K;
  if (N <= RANK_COUNT_MAX_N) {
    for (int i = tid; i < N; i += nt) {
      const int r = count_greater(keys, N, keys[i]);
      if (r < K) top[(size_t)b * K + r] = (int64_t)i;
      else if (bottom != nullptr) bottom[(size_t)b * R + (N - 1 - r)] = (int64_t)i;
    }
    return;
  }
  bitonic_sort_desc(keys, NP);
  for (int r = tid; r < N; r += nt) {
    const int64_t i = (int64_t)(0xffffffffu - (unsigned int)keys[r]);
    if (r < K) top[(size_t)b * K + r] = i;
    else if (bottom != nullptr) bottom[(size_t)b * R + (N - 1 - r)] = i;
  }
}

__global__ void adv_swap_kernel(int64_t* __restrict__ top, const int64_t* __restrict__ bottom,
                                const int64_t* __restrict__ pos, const int64_t* __restrict__ src,
                                int K, int R, int k_adv) {
  const int b = blockIdx.x;
  for (int j = threadIdx.x; j < k_adv; j += blockDim.x) {
    const int64_t p = pos[(size_t)b * k_adv + j], s = src[(size_t)b * k_adv + j];
    if (p >= 0 && p < K && s >= 0 && s < R) top[(size_t)b * K + p] = bottom[(size_t)b * R + s];
  }
}

__global__ void adv_swap_random_kernel(int64_t* __restrict__ top, const int64_t* __restrict__ bottom,
                                       int K, int R, int k_adv, int NPK, int NPR,
                                       const uint64_t* __restrict__ rng_state,
                                       int64_t* __restrict__ pos_out, int64_t* __restrict__ src_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
  __shared__ int s_pos[TK_MAX_N / 2];  // k_adv <= K/1 but bounded by host check
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const unsigned long long seed = rng_state[0], counter = rng_state[1];
  __shared__ int s_src[TK_MAX_N / 2];
  // positions inside the keep list, then sources inside the bottom list
  random_prefix_positions(keys, K, NPK, seed, counter, 2ull * b, k_adv, s_pos);
  random_prefix_positions(keys, R, NPR, seed, counter, 2ull * b + 1ull, k_adv, s_src);
  for (int j = tid; j < k_adv; j += nt) {
    const int p = s_pos[j];
    const int s = s_src[j];
    top[(size_t)b * K + p] = bottom[(size_t)b * R + s];
    if (pos_out != nullptr) pos_out[(size_t)b * k_adv + j] = p;
    if (src_out != nullptr) src_out[(size_t)b * k_adv + j] = s;
  }
}

// Whole keep-index selection of one sample in one workgroup: (optionally) head scores -> softmax
// per head -> head mean, ranking, keep / bottom lists, adversarial swap with injected or in-kernel
// draws.  Replaces four launches (weights, top-k, swap, counter bump is still separate) and keeps
// the weights and both index lists in LDS in between.
constexpr int SEL_MAX_MODS = 4;
struct SelectMod {
  const float* s;
  const float* w_in;
  const int64_t* pos;
  const int64_t* src;
  float* w_out;
  int64_t* top;
  int64_t* bottom;
  int64_t* pos_out;
  int64_t* src_out;
  unsigned long long salt;
  int N, K, NP, adv_mode, k_adv, NPK, NPR;
};
struct SelectMods {
  SelectMod m[SEL_MAX_MODS];
};
__global__ void select_kernel(SelectMods mods, int H, const uint64_t* __restrict__ rng_state) {
  const SelectMod& M_ = mods.m[blockIdx.y];
  const float* __restrict__ s = M_.s;
  const int N = M_.N, K = M_.K, NP = M_.NP, k_adv = M_.k_adv;
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];  // [NP + 2]
  __shared__ float stat[2 * SEL_MAX_H];
  SelectLds L;
  L.keys = keys;                                                        // [NP + 2]
  L.sl = reinterpret_cast<float*>(keys + NP + 2);                       // [N*H] head scores (16-B aligned)
  const int slots = (s != nullptr) ? ((N * H + 3) & ~3) : 0;
  L.rank = reinterpret_cast<int*>(L.sl + slots);                        // [N] full ranking
  L.s_pos = L.rank + N;                                                 // [2*k_adv]: positions, sources
  L.wl = reinterpret_cast<float*>(L.s_pos + (k_adv > 0 ? 2 * k_adv : 1)); // [N]
  L.stat = stat;
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int R = N - K;
  if (s != nullptr) {
    // one coalesced sweep brings the sample's N x H scores into LDS; every later pass reads LDS
    const float* sb = s + (size_t)b * N * H;
    if (((N * H) & 3) == 0) {  // rows of H floats: 16-byte loads whenever N*H is a multiple of 4
      const f32x4* sb4 = reinterpret_cast<const f32x4*>(sb);
      f32x4* sl4 = reinterpret_cast<f32x4*>(L.sl);
      for (int i = tid; i < (N * H) >> 2; i += nt) sl4[i] = sb4[i];
    } else {
      for (int i = tid; i < N * H; i += nt) L.sl[i] = sb[i];
    }
  } else {
    for (int n = tid; n < N; n += nt) L.wl[n] = M_.w_in[(size_t)b * N + n];
  }
  __syncthreads();
  unsigned long long seed = 0, counter = 0;
  if (M_.adv_mode == 2) {
    seed = rng_state[0] + M_.salt;
    counter = rng_state[1];
  }
  select_core(L, s != nullptr ? H : 0, N, K, NP, M_.adv_mode, k_adv, M_.NPK, M_.NPR,
              M_.pos ? M_.pos + (size_t)b * k_adv : nullptr, M_.src ? M_.src + (size_t)b * k_adv : nullptr,
              seed, counter, (unsigned long long)b,
              M_.w_out ? M_.w_out + (size_t)b * N : nullptr, M_.bottom ? M_.bottom + (size_t)b * R : nullptr,
              M_.pos_out ? M_.pos_out + (size_t)b * k_adv : nullptr,
              M_.src_out ? M_.src_out + (size_t)b * k_adv : nullptr);
  for (int j = tid; j < K; j += nt) M_.top[(size_t)b * K + j] = (int64_t)L.rank[j];
}

__global__ void rng_advance_kernel(uint64_t* state, uint64_t delta) {
  if (threadIdx.x == 0 && blockIdx.x == 0) state[1] += delta;
}

inline int next_pow2(int n) { return sel_next_pow2(n); }
inline int sort_threads(int NP) {
  int t = NP / 2;
  if (t < 64) t = 64;
  if (t > 1024) t = 1024;
  return t;
}

}  // namespace

extern "C" int vqaf_topk(const float* w, int B, int N, int K, int64_t* top, int64_t* bottom,
                         vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(w != nullptr && top != nullptr, VQAF_E_NULL);
  VQAF_REQUIRE(B > 0 && N > 0 && K > 0 && K <= N && N <= TK_MAX_N, VQAF_E_SHAPE);
  const int NP = next_pow2(N);
  hipLaunchKernelGGL(topk_kernel, dim3(B), dim3(sort_threads(NP)), (size_t)NP * 8,
                     (hipStream_t)stream, w, N, K, NP, top, bottom);
  VQAF_LAUNCH_CHECK();
}

extern "C" int vqaf_adv_swap(int64_t* top, const int64_t* bottom, const int64_t* pos,
                             const int64_t* src, int B, int K, int R, int k_adv,
                             vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(top != nullptr && bottom != nullptr && pos != nullptr && src != nullptr, VQAF_E_NULL);
  VQAF_REQUIRE(B > 0 && K > 0 && R > 0 && k_adv > 0 && k_adv <= K && k_adv <= R, VQAF_E_SHAPE);
  hipLaunchKernelGGL(adv_swap_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, top, bottom, pos,
                     src, K, R, k_adv);
  VQAF_LAUNCH_CHECK();
}

extern "C" int vqaf_adv_swap_random(int64_t* top, const int64_t* bottom, int B, int K, int R,
                                    int k_adv, const uint64_t* rng_state, int64_t* pos_out,
                                    int64_t* src_out, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(top != nullptr && bottom != nullptr && rng_state != nullptr, VQAF_E_NULL);
  VQAF_REQUIRE(B > 0 && K > 0 && R > 0 && k_adv > 0 && k_adv <= K && k_adv <= R, VQAF_E_SHAPE);
  VQAF_REQUIRE(K <= TK_MAX_N && R <= TK_MAX_N && k_adv <= TK_MAX_N / 2, VQAF_E_SHAPE);
  const int NPK = next_pow2(K), NPR = next_pow2(R);
  const int NP = NPK > NPR ? NPK : NPR;
  hipLaunchKernelGGL(adv_swap_random_kernel, dim3(B), dim3(sort_threads(NP)), (size_t)NP * 8,
                     (hipStream_t)stream, top, bottom, K, R, k_adv, NPK, NPR, rng_state, pos_out,
                     src_out);
  VQAF_LAUNCH_CHECK();
}

static int select_fill(SelectMod& m, const float* s, const float* w_in, int N, int H, int K, int adv_mode,
                       int k_adv, const int64_t* pos, const int64_t* src, const uint64_t* rng_state,
                       uint64_t salt, float* w_out, int64_t* top, int64_t* bottom, int64_t* pos_out,
                       int64_t* src_out, size_t* smem, int* threads) {
  VQAF_REQUIRE((s != nullptr) != (w_in != nullptr), VQAF_E_NULL);  // exactly one source of weights
  VQAF_REQUIRE(top != nullptr, VQAF_E_NULL);
  VQAF_REQUIRE(N > 0 && K > 0 && K <= N && N <= TK_MAX_N, VQAF_E_SHAPE);
  VQAF_REQUIRE(s == nullptr || (H > 0 && H <= SEL_MAX_H), VQAF_E_SHAPE);
  VQAF_REQUIRE(adv_mode >= 0 && adv_mode <= 2, VQAF_E_SHAPE);
  const int R = N - K;
  if (adv_mode != 0) {
    VQAF_REQUIRE(k_adv > 0 && k_adv <= K && k_adv <= R, VQAF_E_SHAPE);
    if (adv_mode == 1) VQAF_REQUIRE(pos != nullptr && src != nullptr, VQAF_E_NULL);
    if (adv_mode == 2) VQAF_REQUIRE(rng_state != nullptr, VQAF_E_NULL);
  } else {
    k_adv = 0;
  }
  const int NP = next_pow2(N);
  m = SelectMod{s, w_in, pos, src, w_out, top, bottom, pos_out, src_out, (unsigned long long)salt,
                N, K, NP, adv_mode, k_adv, next_pow2(K), next_pow2(R > 0 ? R : 1)};
  const size_t need = (size_t)(NP + 2) * 8 + (size_t)N * 4 + (size_t)(k_adv > 0 ? 2 * k_adv : 1) * 4 + (size_t)N * 4 +
                      (s != nullptr ? (size_t)((N * H + 3) & ~3) * 4 : 0);
  VQAF_REQUIRE(need <= 160 * 1024, VQAF_E_SHAPE);  // one workgroup may take the whole LDS of a compute unit
  *smem = need > *smem ? need : *smem;
  int t = sort_threads(NP);
  if (t < 256) t = 256;  // the score / weight passes want more lanes than the sort
  *threads = t > *threads ? t : *threads;
  return VQAF_OK;
}

extern "C" int vqaf_select(const float* s, const float* w_in, int B, int N, int H, int K, int adv_mode,
                           int k_adv, const int64_t* pos, const int64_t* src,
                           const uint64_t* rng_state, uint64_t salt, float* w_out, int64_t* top,
                           int64_t* bottom, int64_t* pos_out, int64_t* src_out, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(B > 0, VQAF_E_SHAPE);
  SelectMods mods = {};
  size_t smem = 0;
  int threads = 0;
  const int rc = select_fill(mods.m[0], s, w_in, N, H, K, adv_mode, k_adv, pos, src, rng_state, salt, w_out, top,
                             bottom, pos_out, src_out, &smem, &threads);
  if (rc != VQAF_OK) return rc;
  if (smem > 64 * 1024)  // beyond the default limit of dynamic LDS: raise it for this kernel (cheap when already set)
    VQAF_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(select_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess, VQAF_E_LAUNCH);
  hipLaunchKernelGGL(select_kernel, dim3(B, 1), dim3(threads), smem, (hipStream_t)stream, mods, H, rng_state);
  VQAF_LAUNCH_CHECK();
}

extern "C" int vqaf_select_multi(const vqaf_select_desc* descs_host, int n, int B, int H,
                                 const uint64_t* rng_state, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(descs_host != nullptr, VQAF_E_NULL);
  VQAF_REQUIRE(n > 0 && n <= SEL_MAX_MODS && B > 0, VQAF_E_SHAPE);
  SelectMods mods = {};
  size_t smem = 0;
  int threads = 0;
  for (int i = 0; i < n; ++i) {
    const vqaf_select_desc& d = descs_host[i];
    const int rc = select_fill(mods.m[i], d.s, d.w_in, d.N, H, d.K, d.adv_mode, d.k_adv, d.pos, d.src, rng_state,
                               d.salt, d.w_out, d.top, d.bottom, d.pos_out, d.src_out, &smem, &threads);
    if (rc != VQAF_OK) return rc;
  }
  if (smem > 64 * 1024)  // beyond the default limit of dynamic LDS: raise it for this kernel (cheap when already set)
    VQAF_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(select_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess, VQAF_E_LAUNCH);
  hipLaunchKernelGGL(select_kernel, dim3(B, n), dim3(threads), smem, (hipStream_t)stream, mods, H, rng_state);
  VQAF_LAUNCH_CHECK();
}

extern "C" int vqaf_rng_advance(uint64_t* rng_state, uint64_t delta, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(rng_state != nullptr, VQAF_E_NULL);
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, rng_state, delta);
  VQAF_LAUNCH_CHECK();
}
