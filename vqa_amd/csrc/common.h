// Shared device/host helpers for the libvqafuse kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vqaf.h"

#define VQAF_WAVE 64

#define VQAF_REQUIRE(cond, code) \
  do {                           \
    if (!(cond)) return (code);  \
  } while (0)

// hipGetLastError() is sticky per thread: an error left behind by an unrelated earlier runtime
// call (for example an aborted stream capture) would otherwise be blamed on the next launch.
// Every entry point therefore clears it first and reports only what its own launches raise.
struct VqafEntryScope {
  VqafEntryScope() { (void)hipGetLastError(); }
};
#define VQAF_ENTRY() VqafEntryScope vqaf_entry_scope_

#define VQAF_LAUNCH_CHECK() \
  do { return hipGetLastError() == hipSuccess ? VQAF_OK : VQAF_E_LAUNCH; } while (0)

static inline bool vqaf_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline bool vqaf_dtype_ok(int d) { return d == VQAF_F32 || d == VQAF_BF16; }
static inline size_t vqaf_esize(int d) { return d == VQAF_BF16 ? 2 : 4; }

typedef unsigned short bf16_t;  // raw bf16 bits

// Device address of 16 bytes of zeros on the current device (the source of out-of-range operand chunks of
// the LDS-DMA GEMM kernels); defined in gemm_bf16.hip.  Not part of the C ABI.
__attribute__((visibility("hidden"))) int vqaf_internal_zero_block(const unsigned int** out);
// The 256 x 256-tile weight-gradient kernel (wgrad256.hip) behind vqaf_wgrad_tn: how it cuts the token range, and
// the launch of its partial products into slabs (splits x M*N f32; cs_slabs: splits x M column sums, or NULL).
__attribute__((visibility("hidden"))) void vqaf_internal_wgrad256_plan(int M, int N, int K, int* k_per_split, int* splits);
__attribute__((visibility("hidden"))) int vqaf_internal_wgrad256_launch(const void* gy, long ldg, const void* x, long ldx,
                                                                        int M, int N, int K, float* slabs,
                                                                        float* cs_slabs, hipStream_t st);

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float bf16_lo(unsigned int w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned int w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ float bf16_to_f32(bf16_t h) { return __uint_as_float(((unsigned int)h) << 16); }

// Round-to-nearest-even f32 -> bf16; NaN stays NaN (quiet).  One hardware conversion (v_cvt_pk_bf16_f32).
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  return (bf16_t)__builtin_bit_cast(unsigned short, (__bf16)f);
}
// Two values in one instruction (v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays a quiet NaN): the same
// results as f32_to_bf16 for every non-NaN input at an eighth of the vector instructions.
typedef float vqaf_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 vqaf_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(vqaf_f32x2{lo, hi}, vqaf_bf16x2));
}

// Loads 8 consecutive elements starting at element offset `e` (multiple of 8) as f32.
template <bool BF16>
__device__ __forceinline__ void load8(const void* base, size_t e, float (&v)[8]) {
  if constexpr (BF16) {
    const u32x4 r = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(base) + e);
    v[0] = bf16_lo(r.x); v[1] = bf16_hi(r.x); v[2] = bf16_lo(r.y); v[3] = bf16_hi(r.y);
    v[4] = bf16_lo(r.z); v[5] = bf16_hi(r.z); v[6] = bf16_lo(r.w); v[7] = bf16_hi(r.w);
  } else {
    const f32x4* p = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + e);
    const f32x4 a = p[0], b = p[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
}
template <bool BF16>
__device__ __forceinline__ void store8(void* base, size_t e, const float (&v)[8]) {
  if constexpr (BF16) {
    u32x4 r;
    r.x = pack_bf16x2(v[0], v[1]); r.y = pack_bf16x2(v[2], v[3]);
    r.z = pack_bf16x2(v[4], v[5]); r.w = pack_bf16x2(v[6], v[7]);
    *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(base) + e) = r;
  } else {
    f32x4* p = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + e);
    f32x4 a, b;
    a.x = v[0]; a.y = v[1]; a.z = v[2]; a.w = v[3]; b.x = v[4]; b.y = v[5]; b.z = v[6]; b.w = v[7];
    p[0] = a; p[1] = b;
  }
}
// Streaming (non-temporal) stores for large outputs that nothing re-reads soon: they go to HBM
// without displacing re-used data from the L2 / Infinity Cache and without leaving hundreds of
// megabytes of dirty lines for the next kernel's reads to flush.
__device__ __forceinline__ void store_nt16(void* p, u32x4 v) {
  __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
}
template <bool BF16>
__device__ __forceinline__ void store8_nt(void* base, size_t e, const float (&v)[8]) {
  if constexpr (BF16) {
    u32x4 r;
    r.x = pack_bf16x2(v[0], v[1]); r.y = pack_bf16x2(v[2], v[3]);
    r.z = pack_bf16x2(v[4], v[5]); r.w = pack_bf16x2(v[6], v[7]);
    store_nt16(reinterpret_cast<bf16_t*>(base) + e, r);
  } else {
    // f32: a lane's 32 bytes go out as two 16-byte stores, each covering only every other
    // 16 bytes of a line across the wave; streamed (uncached) that is two partial-line writes
    // (measured 2.2 TB/s), through the cache the halves merge -- so f32 uses ordinary stores
    store8<false>(base, e, v);
  }
}

template <bool BF16>
__device__ __forceinline__ float load1(const void* base, size_t e) {
  if constexpr (BF16) return bf16_to_f32(reinterpret_cast<const bf16_t*>(base)[e]);
  else return reinterpret_cast<const float*>(base)[e];
}
template <bool BF16>
__device__ __forceinline__ void store1(void* base, size_t e, float v) {
  if constexpr (BF16) reinterpret_cast<bf16_t*>(base)[e] = f32_to_bf16(v);
  else reinterpret_cast<float*>(base)[e] = v;
}

// GroupNorm backward of eight consecutive channels of one row whose group spans `lpg` adjacent lanes
// (gs = 8 * lpg channels): dx = rstd * (g - mean(g) - xhat * mean(g * xhat)) with g = gy * gamma.
// Every operation is pinned (no contraction left to the compiler), so the kernels that share this
// function produce the same bits.
__device__ __forceinline__ void vqaf_gn_bwd8(const float (&gy)[8], const float (&xh)[8], const float (&ga)[8],
                                             int lpg, int gs, float rs, float (&dx)[8]) {
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    dx[i] = __fmul_rn(gy[i], ga[i]);
    s1 = __fadd_rn(s1, dx[i]);
    s2 = fmaf(dx[i], xh[i], s2);
  }
  for (int o = lpg >> 1; o > 0; o >>= 1) {
    s1 = __fadd_rn(s1, __shfl_xor(s1, o, 64));
    s2 = __fadd_rn(s2, __shfl_xor(s2, o, 64));
  }
  const float m1 = __fdiv_rn(s1, (float)gs), m2 = __fdiv_rn(s2, (float)gs);
#pragma unroll
  for (int i = 0; i < 8; ++i) dx[i] = __fmul_rn(rs, fmaf(-xh[i], m2, __fsub_rn(dx[i], m1)));
}

// GroupNorm forward of eight consecutive channels (same layout): v <- (v - mean) * rstd, returns rstd.
__device__ __forceinline__ float vqaf_gn_fwd8(float (&v)[8], int lpg, int gs, float eps) {
  float sm = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) sm = __fadd_rn(sm, v[i]);
  for (int o = lpg >> 1; o > 0; o >>= 1) sm = __fadd_rn(sm, __shfl_xor(sm, o, 64));
  const float mean = __fdiv_rn(sm, (float)gs);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    v[i] = __fsub_rn(v[i], mean);
    q = fmaf(v[i], v[i], q);
  }
  for (int o = lpg >> 1; o > 0; o >>= 1) q = __fadd_rn(q, __shfl_xor(q, o, 64));
  const float rs = rsqrtf(__fadd_rn(__fdiv_rn(q, (float)gs), eps));
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = __fmul_rn(v[i], rs);
  return rs;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---------------------------------------------------------------------------------------
// Counter-based RNG: Philox4x32-10 (Salmon et al., SC'11).  A draw is addressed by
// (seed, counter, subsequence, index): results depend on those four numbers only, never on
// the launch geometry, so every kernel that consumes randomness is reproducible and can be
// re-checked on the host.
// ---------------------------------------------------------------------------------------
struct Philox4 {
  unsigned int x, y, z, w;
};
__host__ __device__ __forceinline__ unsigned int vqaf_mulhi32(unsigned int a, unsigned int b) {
  return (unsigned int)(((unsigned long long)a * (unsigned long long)b) >> 32);
}
__host__ __device__ __forceinline__ Philox4 philox4x32_10(unsigned long long seed,
                                                          unsigned long long ctr_lo,
                                                          unsigned long long ctr_hi) {
  unsigned int k0 = (unsigned int)seed, k1 = (unsigned int)(seed >> 32);
  Philox4 c = {(unsigned int)ctr_lo, (unsigned int)(ctr_lo >> 32), (unsigned int)ctr_hi,
               (unsigned int)(ctr_hi >> 32)};
#pragma unroll
  for (int r = 0; r < 10; ++r) {
#if defined(__HIP_DEVICE_COMPILE__)
    // One v_mad_u64_u32 yields both halves of a 32x32 product; left alone the compiler issues a
    // v_mul_hi_u32 and a v_mul_lo_u32 (each a quarter-rate instruction) per product, and the forty
    // of them are most of the generator's cost.
    unsigned long long p0, p1, cy0, cy1;
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(p0), "=s"(cy0) : "s"(0xD2511F53u), "v"(c.x));
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(p1), "=s"(cy1) : "s"(0xCD9E8D57u), "v"(c.z));
    const unsigned int hi0 = (unsigned int)(p0 >> 32), lo0 = (unsigned int)p0;
    const unsigned int hi1 = (unsigned int)(p1 >> 32), lo1 = (unsigned int)p1;
#else
    const unsigned int hi0 = vqaf_mulhi32(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    const unsigned int hi1 = vqaf_mulhi32(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
#endif
    Philox4 n = {hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
    c = n;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}
// (0,1] uniform from 32 random bits (never 0, so log() is finite): (r + 0.5) * 2^-32.
__host__ __device__ __forceinline__ float vqaf_u01(unsigned int r) {
  return ((float)r + 0.5f) * 2.3283064365386963e-10f;
}
// Two independent N(0,1) from two 32-bit words (Box-Muller).
__device__ __forceinline__ void vqaf_box_muller(unsigned int a, unsigned int b, float& n0, float& n1) {
  const float u = vqaf_u01(a), v = vqaf_u01(b);
  const float r = sqrtf(-2.0f * logf(u));
  float s, c;
  sincospif(2.0f * v, &s, &c);
  n0 = r * c;
  n1 = r * s;
}
// Draw `call` of stream (seed, counter) -> four N(0,1) values z[0..3].
__device__ __forceinline__ void vqaf_normal4(unsigned long long seed, unsigned long long counter,
                                             unsigned long long call, float (&z)[4]) {
  const Philox4 r = philox4x32_10(seed, call, counter);
  vqaf_box_muller(r.x, r.y, z[0], z[1]);
  vqaf_box_muller(r.z, r.w, z[2], z[3]);
}
// One AdamW update of the DP step (torch.optim.AdamW's arithmetic on g = (S + noise_std z) scale), with
// every operation pinned so that the kernels which share it produce the same bits.
struct VqafAdamConsts {
  float noise_std, scale, decay, one_m_b1, beta2, one_m_b2, bc2_sqrt, step_size, eps;
};
__device__ __forceinline__ VqafAdamConsts vqaf_adam_consts(unsigned long long steps_taken, float noise_std, float scale,
                                                           float lr, float beta1, float beta2, float eps,
                                                           float weight_decay) {
  const float t = (float)(steps_taken + 1ull);
  const float bc1 = 1.f - powf(beta1, t);
  VqafAdamConsts c;
  c.noise_std = noise_std;
  c.scale = scale;
  c.decay = 1.f - lr * weight_decay;
  c.one_m_b1 = 1.f - beta1;
  c.beta2 = beta2;
  c.one_m_b2 = 1.f - beta2;
  c.bc2_sqrt = sqrtf(1.f - powf(beta2, t));
  c.step_size = lr / bc1;
  c.eps = eps;
  return c;
}
__device__ __forceinline__ void vqaf_adamw1(const VqafAdamConsts& c, float s, float z, float& p, float& m, float& v,
                                            float& g) {
  g = __fmul_rn(fmaf(c.noise_std, z, s), c.scale);
  p = __fmul_rn(p, c.decay);
  m = fmaf(c.one_m_b1, __fsub_rn(g, m), m);                        // lerp, as torch
  v = fmaf(__fmul_rn(c.one_m_b2, g), g, __fmul_rn(c.beta2, v));
  const float denom = __fadd_rn(__fdiv_rn(sqrtf(v), c.bc2_sqrt), c.eps);
  p = fmaf(-c.step_size, __fdiv_rn(m, denom), p);
}

// In-kernel counter bump for a FOUR-word generator state {seed, counter, ticket, reserved}: every
// workgroup calls this once, after its last read of the state; the last one to arrive adds `delta`
// to the counter and clears the ticket (which starts at zero and is zero between launches).  Saves the
// separate one-thread launch that would otherwise follow every kernel that draws random numbers.
// All threads of the workgroup must call it (it contains a barrier).
__device__ __forceinline__ void vqaf_rng_bump_by_last_workgroup(uint64_t* state, unsigned long long delta,
                                                                 unsigned int total_workgroups) {
  __syncthreads();
  if (threadIdx.x == 0 && threadIdx.y == 0 && threadIdx.z == 0) {
    __threadfence();
    unsigned int* ticket = reinterpret_cast<unsigned int*>(state + 2);
    if (atomicAdd(ticket, 1u) == total_workgroups - 1u) {
      state[1] += delta;
      atomicExch(ticket, 0u);
    }
  }
}

// Token noise (utils/fusion.py:96-116: a regulariser added to kept tokens, not the privacy
// mechanism; the DP noise keeps vqaf_normal4).  One Philox call serves EIGHT consecutive elements:
// each 32-bit word is split into two 16-bit uniforms ((h + 0.5) / 65536) that give one Box-Muller
// pair through the hardware transcendentals (v_log_f32, v_sqrt_f32, v_sin/v_cos_f32, whose argument
// is in revolutions, so the angle goes in unscaled), or two U(-1,1) values.  Word i of the call
// serves elements 2i and 2i+1.  16 bits per uniform are far finer than the bf16 tokens the noise
// is added to and cap |z| at 4.85 sigma; the generator, not the transform, is what the noise costs
// (twenty 32x32-bit products per call), so halving the calls halves it.
__device__ __forceinline__ void vqaf_token_normal8(unsigned long long seed, unsigned long long counter,
                                                   unsigned long long call, float (&z)[8]) {
  const Philox4 r = philox4x32_10(seed, call, counter);
  const unsigned int w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float u = ((float)(w[i] & 0xffffu) + 0.5f) * (1.0f / 65536.0f);
    const float v = ((float)(w[i] >> 16) + 0.5f) * (1.0f / 65536.0f);
    const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u));  // -2 ln u
    z[2 * i] = rad * __builtin_amdgcn_cosf(v);
    z[2 * i + 1] = rad * __builtin_amdgcn_sinf(v);
  }
}
__device__ __forceinline__ void vqaf_token_uniform8(unsigned long long seed, unsigned long long counter,
                                                    unsigned long long call, float (&z)[8]) {
  const Philox4 r = philox4x32_10(seed, call, counter);
  const unsigned int w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    z[2 * i] = ((float)(w[i] & 0xffffu) + 0.5f) * (2.0f / 65536.0f) - 1.0f;
    z[2 * i + 1] = ((float)(w[i] >> 16) + 0.5f) * (2.0f / 65536.0f) - 1.0f;
  }
}
// Adds the noise of elements [e, e + 8) (e a multiple of 8) to v: injected values, or draw e / 8 of
// stream (seed, counter).
__device__ __forceinline__ void vqaf_token_noise8(float (&v)[8], const float* noise, int noise_kind,
                                                  float noise_scale, unsigned long long seed,
                                                  unsigned long long counter, size_t e) {
  if (noise != nullptr) {
    const f32x4* p = reinterpret_cast<const f32x4*>(noise + e);
    const f32x4 a = p[0], b = p[1];
    v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
    v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
  } else if (noise_kind != VQAF_NOISE_NONE) {
    float z[8];
    if (noise_kind == VQAF_NOISE_GAUSSIAN) vqaf_token_normal8(seed, counter, e >> 3, z);
    else vqaf_token_uniform8(seed, counter, e >> 3, z);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = fmaf(noise_scale, z[i], v[i]);
  }
}
// Same draw as four U(-1,1) values.
__device__ __forceinline__ void vqaf_uniform4(unsigned long long seed, unsigned long long counter,
                                              unsigned long long call, float (&z)[4]) {
  const Philox4 r = philox4x32_10(seed, call, counter);
  z[0] = 2.0f * vqaf_u01(r.x) - 1.0f; z[1] = 2.0f * vqaf_u01(r.y) - 1.0f;
  z[2] = 2.0f * vqaf_u01(r.z) - 1.0f; z[3] = 2.0f * vqaf_u01(r.w) - 1.0f;
}
