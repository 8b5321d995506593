// DP-SGD step kernels for the classifier: per-sample ("ghost") gradient norms, clip
// coefficients, clipped gradient sums, Gaussian noise + scaling, GroupNorm forward/backward.
//
// Per-sample gradients are never materialised.  For y = x W^T + b on 2-D input the
// per-sample weight gradient is the rank-1 matrix g_i a_i^T (g_i = dL_i/dy_i, a_i = x_i), so
//   ||grad_i W||_F^2 = ||a_i||^2 ||g_i||^2,       sum_i c_i grad_i W = G^T diag(c) A,
// i.e. two row-norm passes plus one (Out x B) x (B x In) GEMM whose contraction index is the
// batch.  That GEMM runs on the matrix cores: operands are converted to f32 while they are
// staged, so the f32 path is exact f32 (v_mfma_f32_32x32x2_f32 is a k-ordered fmaf chain).
#include <mutex>

#include "common.h"

namespace {

constexpr int DP_THREADS = 256;

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
  return t;
}

template <bool BF16>
__global__ void __launch_bounds__(DP_THREADS)
ghost_sqnorm_linear_kernel(const void* __restrict__ a, const void* __restrict__ g, int In, int Out,
                           int has_bias, int accumulate, float* __restrict__ sq) {
  __shared__ float red[DP_THREADS / 64];
  const int b = blockIdx.x;
  float na = 0.f, ng = 0.f;
  for (int i = threadIdx.x; i < In; i += DP_THREADS) {
    const float v = load1<BF16>(a, (size_t)b * In + i);
    na = fmaf(v, v, na);
  }
  for (int i = threadIdx.x; i < Out; i += DP_THREADS) {
    const float v = load1<BF16>(g, (size_t)b * Out + i);
    ng = fmaf(v, v, ng);
  }
  na = block_sum(na, red);
  ng = block_sum(ng, red);
  if (threadIdx.x == 0) {
    const float v = na * ng + (has_bias ? ng : 0.f);
    sq[b] = accumulate ? sq[b] + v : v;
  }
}

template <bool BF16>
__global__ void __launch_bounds__(DP_THREADS)
ghost_sqnorm_affine_kernel(const void* __restrict__ xhat, const void* __restrict__ g, int C,
                           int accumulate, float* __restrict__ sq) {
  __shared__ float red[DP_THREADS / 64];
  const int b = blockIdx.x;
  float s = 0.f;
  for (int i = threadIdx.x; i < C; i += DP_THREADS) {
    const float gv = load1<BF16>(g, (size_t)b * C + i);
    const float xg = load1<BF16>(xhat, (size_t)b * C + i) * gv;
    s = fmaf(xg, xg, s);
    s = fmaf(gv, gv, s);
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) sq[b] = accumulate ? sq[b] + s : s;
}

// All layers' ghost norms and the clip coefficient of one sample in one workgroup:
//   sq = sum_layers { ||a||^2 ||g||^2 (+ ||g||^2 with a bias)   for a Linear
//                   { sum (xhat g)^2 + sum g^2                   for an affine normalisation
//   n = scale sqrt(sq);  coef = scale min(1, C / (n + 1e-6))
constexpr int GC_MAX_LAYERS = 8;
__device__ __forceinline__ bool vqaf_dev_aligned16(const void* p) {
  return (reinterpret_cast<uintptr_t>(p) & 15u) == 0;
}
// Ranges of f32 that a launch clears on the side (sums that later kernels build with atomic adds).
constexpr int GC_MAX_ZERO = 8;
struct GhostZero {
  float* p[GC_MAX_ZERO];
  unsigned int n[GC_MAX_ZERO];  // elements
  int count;
  unsigned long long* bump;  // {unused, count} pair whose count one thread of the launch advances by 1, or NULL
};
__device__ __forceinline__ void ghost_zero_ranges(const GhostZero& Z) {
  const unsigned int stride = gridDim.x * blockDim.x, first = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int r = 0; r < GC_MAX_ZERO; ++r)
    if (r < Z.count)
      for (unsigned int i = first; i < Z.n[r]; i += stride) Z.p[r][i] = 0.f;
  // nothing in this launch reads the count, and launches on one stream do not overlap: a plain store
  if (Z.bump != nullptr && first == 0) Z.bump[1] = Z.bump[1] + 1ull;
}
struct GhostLayers {
  const void* a[GC_MAX_LAYERS];
  const void* g[GC_MAX_LAYERS];
  int in_dim[GC_MAX_LAYERS];
  int out_dim[GC_MAX_LAYERS];
  int kind[GC_MAX_LAYERS];  // 0 Linear without bias, 1 Linear with bias, 2 affine
  int n;
};
// Shared tail of the two ghost_clip kernels: one block reduction for all layers, the clip coefficient.
__device__ __forceinline__ float ghost_clip_coef(const GhostLayers& L, const float (&part)[2 * GC_MAX_LAYERS],
                                                 float (*red)[2 * GC_MAX_LAYERS], float* s_coef, int b, float max_norm,
                                                 float scale, float* coef, float* norms, float* sq_out) {
  constexpr int NW = DP_THREADS / 64;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int j = 0; j < 2 * GC_MAX_LAYERS; ++j) {
    if (j < 2 * L.n) {
      const float w = wave_sum(part[j]);
      if (lane == 0) red[wave][j] = w;
    }
  }
  __syncthreads();
  if (tid == 0) {
    float total = 0.f;
    for (int l = 0; l < L.n; ++l) {
      float s0 = 0.f, s1 = 0.f;
      for (int w = 0; w < NW; ++w) {
        s0 += red[w][2 * l];
        s1 += red[w][2 * l + 1];
      }
      total += L.kind[l] == 2 ? s0 : s0 * s1 + (L.kind[l] == 1 ? s1 : 0.f);
    }
    const float n = scale * sqrtf(total);
    const float c = scale * fminf(1.0f, max_norm / (n + 1e-6f));
    coef[b] = c;
    *s_coef = c;
    if (norms != nullptr) norms[b] = n;
    if (sq_out != nullptr) sq_out[b] = total;
  }
  __syncthreads();
  return *s_coef;
}

__device__ __forceinline__ float sumsq_bf16x8(u32x4 r) {
  float s = 0.f;
  s = fmaf(bf16_lo(r.x), bf16_lo(r.x), s); s = fmaf(bf16_hi(r.x), bf16_hi(r.x), s);
  s = fmaf(bf16_lo(r.y), bf16_lo(r.y), s); s = fmaf(bf16_hi(r.y), bf16_hi(r.y), s);
  s = fmaf(bf16_lo(r.z), bf16_lo(r.z), s); s = fmaf(bf16_hi(r.z), bf16_hi(r.z), s);
  s = fmaf(bf16_lo(r.w), bf16_lo(r.w), s); s = fmaf(bf16_hi(r.w), bf16_hi(r.w), s);
  return s;
}
__device__ __forceinline__ float affine_term_bf16x2(unsigned int x, unsigned int g, float s) {
  const float g0 = bf16_lo(g), g1 = bf16_hi(g), p0 = bf16_lo(x) * g0, p1 = bf16_hi(x) * g1;
  s = fmaf(p0, p0, s); s = fmaf(g0, g0, s);
  s = fmaf(p1, p1, s); s = fmaf(g1, g1, s);
  return s;
}
__device__ __forceinline__ unsigned int scale_bf16x2(unsigned int w, float c) {
  return pack_bf16x2(c * bf16_lo(w), c * bf16_hi(w));
}

// bf16 tapes whose rows are 16-byte aligned and at most 2048 elements wide (the host checks): EVERY row of
// the sample is requested before anything is summed -- a thread holds its eight elements of each row as
// four raw registers, 64 for eight layers -- so the workgroup pays one memory latency instead of one per
// layer, and the in-place scaling g <- c g reuses the registers instead of reading the rows again.
__global__ void __launch_bounds__(DP_THREADS, 4)
ghost_clip_rows_kernel(GhostLayers L, float max_norm, float scale, int scale_g, float* __restrict__ coef,
                       float* __restrict__ norms, float* __restrict__ sq_out, GhostZero Z) {
  __shared__ float red[DP_THREADS / 64][2 * GC_MAX_LAYERS];
  __shared__ float s_coef;
  const int b = blockIdx.x, i0 = threadIdx.x * 8;
  u32x4 ra[GC_MAX_LAYERS], rg[GC_MAX_LAYERS];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int l = 0; l < GC_MAX_LAYERS; ++l) {
    ra[l] = zero4;
    rg[l] = zero4;
    if (l < L.n) {
      const int Out = L.out_dim[l], In = L.kind[l] == 2 ? Out : L.in_dim[l];
      if (i0 < In) ra[l] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(L.a[l]) + (size_t)b * In + i0);
      if (i0 < Out) rg[l] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(L.g[l]) + (size_t)b * Out + i0);
    }
  }
  ghost_zero_ranges(Z);  // while the rows are on their way
  float part[2 * GC_MAX_LAYERS];
#pragma unroll
  for (int l = 0; l < GC_MAX_LAYERS; ++l) {
    float s0 = 0.f, s1 = 0.f;
    if (l < L.n) {
      if (L.kind[l] == 2) {
        s0 = affine_term_bf16x2(ra[l].x, rg[l].x, s0);
        s0 = affine_term_bf16x2(ra[l].y, rg[l].y, s0);
        s0 = affine_term_bf16x2(ra[l].z, rg[l].z, s0);
        s0 = affine_term_bf16x2(ra[l].w, rg[l].w, s0);
      } else {
        s0 = sumsq_bf16x8(ra[l]);
        s1 = sumsq_bf16x8(rg[l]);
      }
    }
    part[2 * l] = s0;
    part[2 * l + 1] = s1;
  }
  const float c = ghost_clip_coef(L, part, red, &s_coef, b, max_norm, scale, coef, norms, sq_out);
  if (!scale_g) return;
#pragma unroll
  for (int l = 0; l < GC_MAX_LAYERS; ++l) {
    if (l < L.n && i0 < L.out_dim[l]) {
      u32x4 v;
      v.x = scale_bf16x2(rg[l].x, c); v.y = scale_bf16x2(rg[l].y, c);
      v.z = scale_bf16x2(rg[l].z, c); v.w = scale_bf16x2(rg[l].w, c);
      *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(const_cast<void*>(L.g[l])) + (size_t)b * L.out_dim[l] + i0) = v;
    }
  }
}

// The general form (f32 tapes, unaligned or wider rows): the layers' partial sums stay in registers until
// ONE block reduction at the end (no barrier between layers), and the in-place scaling moves 16 bytes per
// lane where the row pitch allows it.
template <bool BF16>
__global__ void __launch_bounds__(DP_THREADS)
ghost_clip_kernel(GhostLayers L, float max_norm, float scale, int scale_g, float* __restrict__ coef,
                  float* __restrict__ norms, float* __restrict__ sq_out, GhostZero Z) {
  constexpr int NW = DP_THREADS / 64, SPAN = DP_THREADS * 8;
  __shared__ float red[NW][2 * GC_MAX_LAYERS];
  __shared__ float s_coef;
  const int b = blockIdx.x, tid = threadIdx.x, i0 = tid * 8;
  ghost_zero_ranges(Z);
  // per-thread partial sums: [2l] = ||a||^2 part (Linear) or the whole affine term, [2l+1] = ||g||^2 part
  float part[2 * GC_MAX_LAYERS];
#pragma unroll
  for (int l = 0; l < GC_MAX_LAYERS; ++l) {
    float s0 = 0.f, s1 = 0.f;
    if (l < L.n) {
      const int Out = L.out_dim[l];
      const bool vg = Out % 8 == 0 && vqaf_dev_aligned16(L.g[l]);
      if (L.kind[l] == 2) {
        if (vg && vqaf_dev_aligned16(L.a[l])) {
          for (int i = i0; i < Out; i += SPAN) {
            float gv[8], xv[8];
            load8<BF16>(L.g[l], (size_t)b * Out + i, gv);
            load8<BF16>(L.a[l], (size_t)b * Out + i, xv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float xg = xv[e] * gv[e];
              s0 = fmaf(xg, xg, s0);
              s0 = fmaf(gv[e], gv[e], s0);
            }
          }
        } else {
          for (int i = tid; i < Out; i += DP_THREADS) {
            const float gv = load1<BF16>(L.g[l], (size_t)b * Out + i);
            const float xg = load1<BF16>(L.a[l], (size_t)b * Out + i) * gv;
            s0 = fmaf(xg, xg, s0);
            s0 = fmaf(gv, gv, s0);
          }
        }
      } else {
        const int In = L.in_dim[l];
        if (In % 8 == 0 && vqaf_dev_aligned16(L.a[l])) {
          for (int i = i0; i < In; i += SPAN) {
            float v[8];
            load8<BF16>(L.a[l], (size_t)b * In + i, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) s0 = fmaf(v[e], v[e], s0);
          }
        } else {
          for (int i = tid; i < In; i += DP_THREADS) {
            const float v = load1<BF16>(L.a[l], (size_t)b * In + i);
            s0 = fmaf(v, v, s0);
          }
        }
        if (vg) {
          for (int i = i0; i < Out; i += SPAN) {
            float v[8];
            load8<BF16>(L.g[l], (size_t)b * Out + i, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) s1 = fmaf(v[e], v[e], s1);
          }
        } else {
          for (int i = tid; i < Out; i += DP_THREADS) {
            const float v = load1<BF16>(L.g[l], (size_t)b * Out + i);
            s1 = fmaf(v, v, s1);
          }
        }
      }
    }
    part[2 * l] = s0;
    part[2 * l + 1] = s1;
  }
  const float c = ghost_clip_coef(L, part, red, &s_coef, b, max_norm, scale, coef, norms, sq_out);
  if (!scale_g) return;
  // g_b <- coef_b * g_b for every layer, in place (the rows were just read, so they are cache hot):
  // the clipped-sum kernels then run with coef == NULL, as plain batch contractions.
  for (int l = 0; l < L.n; ++l) {
    const int Out = L.out_dim[l];
    void* gl = const_cast<void*>(L.g[l]);
    if (Out % 8 == 0 && vqaf_dev_aligned16(gl)) {
      for (int i = i0; i < Out; i += SPAN) {
        float v[8];
        load8<BF16>(gl, (size_t)b * Out + i, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= c;
        store8<BF16>(gl, (size_t)b * Out + i, v);
      }
    } else {
      for (int i = tid; i < Out; i += DP_THREADS)
        store1<BF16>(gl, (size_t)b * Out + i, c * load1<BF16>(gl, (size_t)b * Out + i));
    }
  }
}

__global__ void clip_coef_kernel(const float* __restrict__ sq, int B, float max_norm, float scale,
                                 float* __restrict__ coef, float* __restrict__ norms) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float n = scale * sqrtf(sq[b]);
  coef[b] = scale * fminf(1.0f, max_norm / (n + 1e-6f));
  if (norms != nullptr) norms[b] = n;
}

// SW (Out,In) (+)= sum_b coef[b] g[b,:]^T a[b,:]  on f32 MFMA 32x32x2.
// Workgroup tile 64 (Out) x 64 (In), four waves of 32x32; batch rows staged 16 at a time.
constexpr int CS_T = 64;
constexpr int CS_KB = 16;
template <bool BF16>
__global__ void __launch_bounds__(DP_THREADS)
clipped_sum_linear_kernel(const void* __restrict__ a, const void* __restrict__ g,
                          const float* __restrict__ coef, int B, int In, int Out, int accumulate,
                          float* __restrict__ SW, float* __restrict__ Sb) {
  __shared__ float Gs[CS_KB][CS_T];
  __shared__ float As[CS_KB][CS_T];
  __shared__ float bred[4][CS_T];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * CS_T, n0 = blockIdx.x * CS_T;
  const int col = tid & 63, rg = tid >> 6;
  const bool do_bias = (Sb != nullptr) && (blockIdx.x == 0);
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float bsum = 0.f;
  for (int k0 = 0; k0 < B; k0 += CS_KB) {
#pragma unroll
    for (int i = 0; i < CS_KB / 4; ++i) {
      const int r = rg + 4 * i, bb = k0 + r;
      float gv = 0.f, av = 0.f;
      if (bb < B) {
        if (m0 + col < Out) gv = (coef != nullptr ? coef[bb] : 1.f) * load1<BF16>(g, (size_t)bb * Out + m0 + col);
        if (n0 + col < In) av = load1<BF16>(a, (size_t)bb * In + n0 + col);
      }
      Gs[r][col] = gv;
      As[r][col] = av;
      bsum += gv;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < CS_KB; kk += 2) {
      const float af = Gs[kk + (lane >> 5)][wm * 32 + (lane & 31)];
      const float bf = As[kk + (lane >> 5)][wn * 32 + (lane & 31)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  // C/D layout of the 32x32 tile: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const int n = n0 + wn * 32 + (lane & 31);
    if (m < Out && n < In) {
      float* p = SW + (size_t)m * In + n;
      *p = accumulate ? *p + acc[r] : acc[r];
    }
  }
  if (do_bias) {
    bred[rg][col] = bsum;
    __syncthreads();
    if (tid < CS_T && m0 + tid < Out) {
      const float v = (bred[0][tid] + bred[1][tid]) + (bred[2][tid] + bred[3][tid]);
      Sb[m0 + tid] = accumulate ? Sb[m0 + tid] + v : v;
    }
  }
}

// Sb[o] += sum_b g[b,o]: the bias part of a clipped sum whose rows are already scaled.  A
// workgroup owns 64 columns of 256 rows (its four waves take every fourth row and are folded in a
// fixed order) and adds its partial sum atomically; the caller zeroes Sb first unless accumulating.
constexpr int COLSUM_ROWS = 256;
__global__ void __launch_bounds__(DP_THREADS)
colsum_f32_kernel(const float* __restrict__ g, int B, int Out, float* __restrict__ Sb) {
  __shared__ float red[4][64];
  const int col = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int o = blockIdx.x * 64 + col;
  const int b1 = min(B, (int)(blockIdx.y + 1) * COLSUM_ROWS);
  float s0 = 0.f, s1 = 0.f;
  if (o < Out) {
    int b = blockIdx.y * COLSUM_ROWS + rg;
    for (; b + 4 < b1; b += 8) {
      s0 += g[(size_t)b * Out + o];
      s1 += g[(size_t)(b + 4) * Out + o];
    }
    if (b < b1) s0 += g[(size_t)b * Out + o];
  }
  red[rg][col] = s0 + s1;
  __syncthreads();
  if (rg == 0 && o < Out) atomicAdd(&Sb[o], (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]));
}

// bf16 operands: the same batch contraction on v_mfma_f32_32x32x16_bf16.
// Both operands are stored batch-major ((B,Out) and (B,In)), i.e. the contraction index is the
// SLOW index of both, while an MFMA lane needs 8 consecutive k for one row/column.  The tiles are
// therefore staged into LDS exactly as they lie in memory ([k][column], 16-byte row chunks, rows
// padded by 64 B so that the four k-rows of a transposed read fall on disjoint bank groups) and
// the fragments are fetched with ds_read_b64_tr_b16, the hardware transposing read: a 16-lane
// group reads a 4(k) x 16(column) block and each lane receives one column's four k values.
// Workgroup tile 128 (Out) x 128 (In), four waves of 64 x 64 (2 x 2 MFMA tiles), 64 batch rows per
// stage (16 MFMAs per wave and stage, so the staging overhead is amortised; LDS 80 KiB double
// buffered, one workgroup per CU).  Staging goes through registers two stages ahead (loads of stage s+2 are in flight while
// stage s is multiplied), the coefficient is applied while stage s+1 moves from registers to LDS
// (the product is re-rounded to bf16 by v_cvt_pk_bf16_f32).
constexpr int CB_TM = 128;
constexpr int CB_TN = 128;
constexpr int CB_KB = 64;
constexpr int CB_PS = CB_KB / 16;   // 16-row passes per stage and operand
constexpr int CB_PG = CB_TM + 32;  // LDS row pitch of the g tile (elements)
constexpr int CB_PA = CB_TN + 32;  // LDS row pitch of the a tile
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

template <int PITCH>
__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* tile, int kblock, int col0, int lane) {
  // lane 4q+p of each 16-lane group addresses row kblock+q, columns col0 + 4p .. 4p+3
  const int i = lane & 15;
  const bf16_t* p0 = tile + (kblock + (i >> 2)) * PITCH + col0 + 4 * (i & 3);
  typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)p0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p0 + 4 * PITCH));
  union { s16x8 s; bf16x8 b; } u;
  u.s = s16x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  return u.b;
}

// 8 consecutive bf16 of row `row` from column `col`, as raw bits.  Loads are unconditional
// (indices outside the matrix are clamped to a valid element) so that the loads of a stage issue
// back to back; mask8() zeroes what was out of range when the registers are consumed.
// VEC = elements per load instruction the row pitch allows: 8 (16 B), 2 (4 B) or 1.
template <int VEC>
__device__ __forceinline__ u32x4 load_row8(const bf16_t* base, int row, int nrows, int col, int ncols) {
  const bf16_t* p = base + (size_t)min(row, nrows - 1) * ncols;
  if constexpr (VEC == 8) {
    return *reinterpret_cast<const u32x4*>(p + min(col, ncols - 8));
  } else if constexpr (VEC == 2) {
    const unsigned int* q = reinterpret_cast<const unsigned int*>(p);
    const int last = ncols / 2 - 1, c2 = col / 2;
    return u32x4{q[min(c2, last)], q[min(c2 + 1, last)], q[min(c2 + 2, last)], q[min(c2 + 3, last)]};
  } else {
    unsigned int w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
      w[e] = (unsigned int)p[min(col + 2 * e, ncols - 1)] | ((unsigned int)p[min(col + 2 * e + 1, ncols - 1)] << 16);
    return u32x4{w[0], w[1], w[2], w[3]};
  }
}
__device__ __forceinline__ u32x4 mask8(u32x4 v, int row, int nrows, int col, int ncols) {
  const bool r = row < nrows;
  unsigned int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const unsigned int lo = (r && col + 2 * e < ncols) ? 0xffffu : 0u;
    const unsigned int hi = (r && col + 2 * e + 1 < ncols) ? 0xffff0000u : 0u;
    w[e] &= (lo | hi);
  }
  return u32x4{w[0], w[1], w[2], w[3]};
}

__device__ __forceinline__ unsigned int scale_pack2(unsigned int w, float c, float& s0, float& s1) {
  const float lo = bf16_lo(w) * c, hi = bf16_hi(w) * c;
  s0 += lo;
  s1 += hi;
  union { bf16x2 b; unsigned int u; } cv;
  cv.b = bf16x2{(__bf16)lo, (__bf16)hi};
  return cv.u;
}

template <int VEC_G, int VEC_A>
__global__ void __launch_bounds__(DP_THREADS)
clipped_sum_linear_bf16_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ g,
                               const float* __restrict__ coef, int Btot, int In, int Out,
                               int accumulate, float* __restrict__ SW, float* __restrict__ Sb,
                               int rows_per_split) {
  // blockIdx.z owns the batch rows [b_lo, b_hi): with more than one split the partial tiles are
  // added to the (pre-zeroed) output with f32 atomics -- used only when the output has too few
  // tiles to occupy the chip (the 430 x 1024 head).
  const int b_lo = blockIdx.z * rows_per_split;
  const int B = min(Btot, b_lo + rows_per_split) - b_lo;
  const bool split = gridDim.z > 1;
  if (B <= 0) return;
  a += (size_t)b_lo * In;
  g += (size_t)b_lo * Out;
  if (coef != nullptr) coef += b_lo;
  extern __shared__ __attribute__((aligned(16))) bf16_t cb_lds[];
  bf16_t (*ldg)[CB_KB][CB_PG] = reinterpret_cast<bf16_t (*)[CB_KB][CB_PG]>(cb_lds);
  bf16_t (*lda)[CB_KB][CB_PA] = reinterpret_cast<bf16_t (*)[CB_KB][CB_PA]>(cb_lds + 2 * CB_KB * CB_PG);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * CB_TM, n0 = blockIdx.x * CB_TN;
  const int gr = tid >> 4, gc = (tid & 15) * 8;  // rows gr + 16*ps, 8 columns from gc (both tiles)
  const bool do_bias = (Sb != nullptr) && (blockIdx.x == 0);
  const bool interior = (m0 + CB_TM <= Out) && (n0 + CB_TN <= In);  // no column masking needed
  const int g16 = lane >> 4;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float bsum[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bsum[e] = 0.f;

  // two register sets; stage t lives in set t & 1
  u32x4 rg[2][CB_PS], ra[2][CB_PS];
  float rc[2][CB_PS];
  const int stages = (B + CB_KB - 1) / CB_KB;

#define CB_FETCH(SET, T)                                                              \
  do {                                                                                \
    const int k0_ = (T) * CB_KB;                                                      \
    _Pragma("unroll") for (int ps = 0; ps < CB_PS; ++ps) {                            \
      const int row = k0_ + gr + 16 * ps;                                             \
      rg[SET][ps] = load_row8<VEC_G>(g, row, B, m0 + gc, Out);                        \
      ra[SET][ps] = load_row8<VEC_A>(a, row, B, n0 + gc, In);                         \
      rc[SET][ps] = coef != nullptr ? coef[min(row, B - 1)] : 1.f;                    \
    }                                                                                 \
  } while (0)
#define CB_STASH(SET, BUF, T)                                                         \
  do {                                                                                \
    const int k0_ = (T) * CB_KB;                                                      \
    _Pragma("unroll") for (int ps = 0; ps < CB_PS; ++ps) {                            \
      const int row = k0_ + gr + 16 * ps;                                             \
      const float c = rc[SET][ps];                                                    \
      u32x4 v = rg[SET][ps], w = ra[SET][ps];                                         \
      if (!interior || k0_ + CB_KB > B) {                                             \
        v = mask8(v, row, B, m0 + gc, Out);                                           \
        w = mask8(w, row, B, n0 + gc, In);                                            \
      }                                                                               \
      if (coef != nullptr) {                                                          \
        v.x = scale_pack2(v.x, c, bsum[0], bsum[1]);                                  \
        v.y = scale_pack2(v.y, c, bsum[2], bsum[3]);                                  \
        v.z = scale_pack2(v.z, c, bsum[4], bsum[5]);                                  \
        v.w = scale_pack2(v.w, c, bsum[6], bsum[7]);                                  \
      } else if (do_bias) {                                                           \
        bsum[0] += bf16_lo(v.x); bsum[1] += bf16_hi(v.x); bsum[2] += bf16_lo(v.y);    \
        bsum[3] += bf16_hi(v.y); bsum[4] += bf16_lo(v.z); bsum[5] += bf16_hi(v.z);    \
        bsum[6] += bf16_lo(v.w); bsum[7] += bf16_hi(v.w);                             \
      }                                                                               \
      *reinterpret_cast<u32x4*>(&ldg[BUF][gr + 16 * ps][gc]) = v;                     \
      *reinterpret_cast<u32x4*>(&lda[BUF][gr + 16 * ps][gc]) = w;                     \
    }                                                                                 \
  } while (0)

  CB_FETCH(0, 0);
  CB_FETCH(1, 1);
  CB_STASH(0, 0, 0);
  __syncthreads();
  // Branch-free loop body (an odd stage count is padded with one all-zero stage: rows past the
  // batch are clamped on load and masked on the way to LDS), so the waits stay counted.
  const int stages2 = (stages + 1) & ~1;
#pragma unroll 1
  for (int s = 0; s < stages2; s += 2) {
#pragma unroll
    for (int P = 0; P < 2; ++P) {
      const int t = s + P;
      CB_FETCH(P, t + 2);  // set P is free: stage t already sits in LDS
      const bf16_t* tg = &ldg[P][0][0];
      const bf16_t* ta = &lda[P][0][0];
#pragma unroll
      for (int ks = 0; ks < CB_KB / 16; ++ks) {
        const int kblock = 16 * ks + 8 * (g16 >> 1);
        bf16x8 afv[2], bfv[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          afv[i] = tr_frag<CB_PG>(tg, kblock, wm * 64 + i * 32 + 16 * (g16 & 1), lane);
          bfv[i] = tr_frag<CB_PA>(ta, kblock, wn * 64 + i * 32 + 16 * (g16 & 1), lane);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afv[i], bfv[j], acc[i][j], 0, 0, 0);
      }
      CB_STASH(P ^ 1, P ^ 1, t + 1);
      __syncthreads();
    }
  }
#undef CB_FETCH
#undef CB_STASH
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int n = n0 + wn * 64 + j * 32 + (lane & 31);
        if (m < Out && n < In) {
          float* p = SW + (size_t)m * In + n;
          if (split) atomicAdd(p, acc[i][j][r]);
          else *p = accumulate ? *p + acc[i][j][r] : acc[i][j][r];
        }
      }
  if (do_bias) {
    float* red = reinterpret_cast<float*>(cb_lds);  // 16 x 128 f32 = 8 KiB; the loop is done
#pragma unroll
    for (int e = 0; e < 8; ++e) red[gr * CB_TM + gc + e] = bsum[e];
    __syncthreads();
    if (tid < CB_TM && m0 + tid < Out) {
      float v = 0.f;
      for (int r = 0; r < 16; ++r) v += red[r * CB_TM + tid];
      if (split) atomicAdd(Sb + m0 + tid, v);
      else Sb[m0 + tid] = accumulate ? Sb[m0 + tid] + v : v;
    }
  }
}

// Sgamma[c] (+)= sum_b coef[b] xhat[b,c] g[b,c];  Sbeta[c] (+)= sum_b coef[b] g[b,c].
// Grid (column blocks of 128, batch chunks of AF_ROWS): a workgroup sums its chunk with 16-byte
// loads (16 lanes per row, 16 rows in flight), folds the 16 row groups in LDS in a fixed order and
// adds its 128 partial sums to the output with one f32 atomic per column.  The outputs are zeroed
// first when accumulate == 0.  (The order of the few atomic adds per column is not fixed, so the
// last bits of these sums may differ from run to run.)
constexpr int AF_ROWS = 64;
template <bool BF16>
__device__ __forceinline__ void clipped_sum_affine_body(const void* __restrict__ xhat, const void* __restrict__ g,
                                                        const float* __restrict__ coef, int B, int C,
                                                        float* __restrict__ Sgamma, float* __restrict__ Sbeta) {
  __shared__ float r0[16][128], r1[16][128];
  const int tid = threadIdx.x, r16 = tid >> 4, cc = (tid & 15) * 8;
  const int c0 = blockIdx.x * 128 + cc;
  const int b0 = blockIdx.y * AF_ROWS;
  float sg[8], sb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) sg[e] = sb[e] = 0.f;
  if (c0 < C) {
#pragma unroll
    for (int i = 0; i < AF_ROWS / 16; ++i) {
      const int b = b0 + r16 + 16 * i;
      if (b < B) {
        float gv[8], xv[8];
        load8<BF16>(g, (size_t)b * C + c0, gv);
        load8<BF16>(xhat, (size_t)b * C + c0, xv);
        const float c = coef != nullptr ? coef[b] : 1.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float t = c * gv[e];
          sg[e] = fmaf(xv[e], t, sg[e]);
          sb[e] += t;
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    r0[r16][cc + e] = sg[e];
    r1[r16][cc + e] = sb[e];
  }
  __syncthreads();
  if (tid < 128 && blockIdx.x * 128 + tid < C) {
    float vg = 0.f, vb = 0.f;
    for (int r = 0; r < 16; ++r) {
      vg += r0[r][tid];
      vb += r1[r][tid];
    }
    const int c = blockIdx.x * 128 + tid;
    if (Sgamma != nullptr) atomicAdd(Sgamma + c, vg);
    if (Sbeta != nullptr) atomicAdd(Sbeta + c, vb);
  }
}

template <bool BF16>
__global__ void __launch_bounds__(DP_THREADS)
clipped_sum_affine_kernel(const void* __restrict__ xhat, const void* __restrict__ g,
                          const float* __restrict__ coef, int B, int C,
                          float* __restrict__ Sgamma, float* __restrict__ Sbeta) {
  clipped_sum_affine_body<BF16>(xhat, g, coef, B, C, Sgamma, Sbeta);
}

// The same sums for up to AF_MAX affine layers in one launch (blockIdx.z picks the layer).
constexpr int AF_MAX = 8;
struct AffineLayers {
  const void* xhat[AF_MAX];
  const void* g[AF_MAX];
  float* Sgamma[AF_MAX];
  float* Sbeta[AF_MAX];
  int C[AF_MAX];
};
template <bool BF16>
__global__ void __launch_bounds__(DP_THREADS)
clipped_sum_affine_multi_kernel(AffineLayers L, const float* __restrict__ coef, int B) {
  const void* xhat = nullptr;
  const void* g = nullptr;
  float* sg = nullptr;
  float* sb = nullptr;
  int C = 0;
#pragma unroll
  for (int i = 0; i < AF_MAX; ++i)
    if (i == (int)blockIdx.z) { xhat = L.xhat[i]; g = L.g[i]; sg = L.Sgamma[i]; sb = L.Sbeta[i]; C = L.C[i]; }
  if ((int)blockIdx.x * 128 >= C) return;  // the grid is as wide as the widest layer (uniform per workgroup)
  clipped_sum_affine_body<BF16>(xhat, g, coef, B, C, sg, sb);
}

// grad[i] = (S[i] + std * z_i) * scale; z_{4c..4c+3} = draw c of stream (seed, counter).
__global__ void __launch_bounds__(DP_THREADS)
noise_scale_kernel(const float* __restrict__ S, const float* __restrict__ noise,
                   const uint64_t* __restrict__ rng_state, float noise_std, float scale, size_t n,
                   float* __restrict__ grad) {
  unsigned long long seed = 0, counter = 0;
  const bool draw = (noise == nullptr) && (noise_std != 0.f);
  if (draw) {
    seed = rng_state[0];
    counter = rng_state[1];
  }
  const size_t ncall = (n + 3) >> 2;
  for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < ncall;
       c += (size_t)gridDim.x * blockDim.x) {
    float z[4] = {0.f, 0.f, 0.f, 0.f};
    if (draw) vqaf_normal4(seed, counter, c, z);
    const size_t i0 = c << 2;
    if (i0 + 3 < n) {
      const f32x4 s4 = *reinterpret_cast<const f32x4*>(S + i0);
      if (noise != nullptr) {
        const f32x4 n4 = *reinterpret_cast<const f32x4*>(noise + i0);
        z[0] = n4.x; z[1] = n4.y; z[2] = n4.z; z[3] = n4.w;
      }
      f32x4 o;
      o.x = (s4.x + noise_std * z[0]) * scale;
      o.y = (s4.y + noise_std * z[1]) * scale;
      o.z = (s4.z + noise_std * z[2]) * scale;
      o.w = (s4.w + noise_std * z[3]) * scale;
      *reinterpret_cast<f32x4*>(grad + i0) = o;
    } else {
      for (int j = 0; j < 4 && i0 + j < n; ++j) {
        const float zz = (noise != nullptr) ? noise[i0 + j] : z[j];
        grad[i0 + j] = (S[i0 + j] + noise_std * zz) * scale;
      }
    }
  }
}

// Noise + scale + AdamW in one pass over the flat buckets (what DPOptimizer.step() does when it
// wraps torch.optim.AdamW): g = (S + std z) * scale, then the decoupled-weight-decay Adam update
// exactly as torch applies it (p *= 1 - lr wd; m, v moments; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)),
// optionally refreshing a bf16 copy of the parameters for the next forward.  `step` holds the
// number of updates already applied (device memory, so a replayed HIP graph keeps counting).
__global__ void __launch_bounds__(DP_THREADS)
dp_adamw_kernel(const float* __restrict__ S, const float* __restrict__ noise,
                const uint64_t* rng_state, float noise_std, float scale,
                float* __restrict__ param, float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq,
                bf16_t* __restrict__ shadow, float* __restrict__ grad_out, size_t n, float lr, float beta1,
                float beta2, float eps, float weight_decay, uint64_t* step, int advance) {
  unsigned long long seed = 0, counter = 0;
  const bool draw = (noise == nullptr) && (noise_std != 0.f);
  // advance < 0: the count was already advanced by one for this step before the launch
  const unsigned long long pre = advance < 0 ? 1ull : 0ull;
  if (draw) {
    seed = rng_state[0];
    counter = rng_state[1] - (rng_state == step ? pre : 0ull);
  }
  const VqafAdamConsts ac = vqaf_adam_consts(step[1] - pre, noise_std, scale, lr, beta1, beta2, eps, weight_decay);
  const size_t ncall = (n + 3) >> 2;
  for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < ncall;
       c += (size_t)gridDim.x * blockDim.x) {
    float z[4] = {0.f, 0.f, 0.f, 0.f};
    if (draw) vqaf_normal4(seed, counter, c, z);
    const size_t i0 = c << 2;
    const int cnt = (i0 + 3 < n) ? 4 : (int)(n - i0);
    float sv[4], pv[4], mv[4], vv[4];
    if (cnt == 4) {
      const f32x4 s4 = *reinterpret_cast<const f32x4*>(S + i0);
      const f32x4 p4 = *reinterpret_cast<const f32x4*>(param + i0);
      const f32x4 m4 = *reinterpret_cast<const f32x4*>(exp_avg + i0);
      const f32x4 v4 = *reinterpret_cast<const f32x4*>(exp_avg_sq + i0);
      sv[0] = s4.x; sv[1] = s4.y; sv[2] = s4.z; sv[3] = s4.w;
      pv[0] = p4.x; pv[1] = p4.y; pv[2] = p4.z; pv[3] = p4.w;
      mv[0] = m4.x; mv[1] = m4.y; mv[2] = m4.z; mv[3] = m4.w;
      vv[0] = v4.x; vv[1] = v4.y; vv[2] = v4.z; vv[3] = v4.w;
      if (noise != nullptr) {
        const f32x4 n4 = *reinterpret_cast<const f32x4*>(noise + i0);
        z[0] = n4.x; z[1] = n4.y; z[2] = n4.z; z[3] = n4.w;
      }
    } else {
      for (int j = 0; j < 4; ++j) {
        const bool ok = j < cnt;
        sv[j] = ok ? S[i0 + j] : 0.f;
        pv[j] = ok ? param[i0 + j] : 0.f;
        mv[j] = ok ? exp_avg[i0 + j] : 0.f;
        vv[j] = ok ? exp_avg_sq[i0 + j] : 0.f;
        if (noise != nullptr && ok) z[j] = noise[i0 + j];
      }
    }
    float gv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) vqaf_adamw1(ac, sv[j], z[j], pv[j], mv[j], vv[j], gv[j]);
    if (cnt == 4) {
      // parameters, moments and gradient are next touched one whole step later: streaming stores
      // keep them from displacing (and dirtying) the caches the next forward reads through
      store_nt16(param + i0, u32x4{__float_as_uint(pv[0]), __float_as_uint(pv[1]), __float_as_uint(pv[2]), __float_as_uint(pv[3])});
      store_nt16(exp_avg + i0, u32x4{__float_as_uint(mv[0]), __float_as_uint(mv[1]), __float_as_uint(mv[2]), __float_as_uint(mv[3])});
      store_nt16(exp_avg_sq + i0, u32x4{__float_as_uint(vv[0]), __float_as_uint(vv[1]), __float_as_uint(vv[2]), __float_as_uint(vv[3])});
      if (grad_out != nullptr)
        store_nt16(grad_out + i0, u32x4{__float_as_uint(gv[0]), __float_as_uint(gv[1]), __float_as_uint(gv[2]), __float_as_uint(gv[3])});
      if (shadow != nullptr)
        *reinterpret_cast<u32x2*>(shadow + i0) = u32x2{pack_bf16x2(pv[0], pv[1]), pack_bf16x2(pv[2], pv[3])};
    } else {
      for (int j = 0; j < cnt; ++j) {
        param[i0 + j] = pv[j];
        exp_avg[i0 + j] = mv[j];
        exp_avg_sq[i0 + j] = vv[j];
        if (grad_out != nullptr) grad_out[i0 + j] = gv[j];
        if (shadow != nullptr) shadow[i0 + j] = f32_to_bf16(pv[j]);
      }
    }
  }
  // `step` (which may also be the noise state) was read before the loop by every thread
  if (advance > 0) vqaf_rng_bump_by_last_workgroup(step, (unsigned long long)advance, gridDim.x);
}

// The same update over up to eight RANGES of the flat buckets in one launch (blockIdx.y picks the range).
// Element i of the buckets draws value (i - noise_lo) & 3 of generator call (i - noise_lo) >> 2: the
// addressing of one dp_adamw_kernel launch over [noise_lo, ...), whichever launch updates the element.
constexpr int DA_MAX_RANGES = 8;
struct AdamRanges {
  long off[DA_MAX_RANGES];            // first element (a multiple of 4 past noise_lo)
  unsigned int n[DA_MAX_RANGES];      // elements
  int count;
};
__global__ void __launch_bounds__(DP_THREADS)
dp_adamw_ranges_kernel(AdamRanges R, const float* __restrict__ S, const uint64_t* rng_state, long noise_lo,
                       float noise_std, float scale, float* __restrict__ param, float* __restrict__ exp_avg,
                       float* __restrict__ exp_avg_sq, bf16_t* __restrict__ shadow, float* __restrict__ grad_out,
                       float lr, float beta1, float beta2, float eps, float weight_decay, const uint64_t* step) {
  long off = 0;
  unsigned int n = 0;
#pragma unroll
  for (int i = 0; i < DA_MAX_RANGES; ++i)
    if (i == (int)blockIdx.y) { off = R.off[i]; n = R.n[i]; }
  const bool draw = noise_std != 0.f;
  const unsigned long long seed = draw ? rng_state[0] : 0ull, counter = draw ? rng_state[1] : 0ull;
  const VqafAdamConsts ac = vqaf_adam_consts(step[1], noise_std, scale, lr, beta1, beta2, eps, weight_decay);
  const unsigned int ncall = (n + 3) >> 2;
  for (unsigned int c = blockIdx.x * blockDim.x + threadIdx.x; c < ncall; c += gridDim.x * blockDim.x) {
    float z[4] = {0.f, 0.f, 0.f, 0.f};
    const long i0 = off + ((long)c << 2);
    if (draw) vqaf_normal4(seed, counter, (unsigned long long)((i0 - noise_lo) >> 2), z);
    const int cnt = min(4, (int)(n - (c << 2)));
    for (int j = 0; j < cnt; ++j) {
      float p = param[i0 + j], m = exp_avg[i0 + j], v = exp_avg_sq[i0 + j], g;
      vqaf_adamw1(ac, S[i0 + j], z[j], p, m, v, g);
      param[i0 + j] = p;
      exp_avg[i0 + j] = m;
      exp_avg_sq[i0 + j] = v;
      if (grad_out != nullptr) grad_out[i0 + j] = g;
      if (shadow != nullptr) shadow[i0 + j] = f32_to_bf16(p);
    }
  }
}

// GroupNorm on (B,C): one workgroup per sample, a lane owns 8 consecutive channels, a group of
// gs = C/groups channels spans gs/8 adjacent lanes (a power of two <= 64).
template <bool BF16>
__global__ void __launch_bounds__(DP_THREADS)
groupnorm_fwd_kernel(const void* __restrict__ x, const void* __restrict__ gamma,
                     const void* __restrict__ beta, int C, int groups, float eps, int relu,
                     void* __restrict__ y, void* __restrict__ xhat, float* __restrict__ rstd,
                     float drop_p, unsigned long long drop_seed, const uint64_t* __restrict__ drop_counter) {
  const int b = blockIdx.x;
  const int gs = C / groups, lpg = gs >= 8 ? gs / 8 : 1;
  // inverted dropout on the output (nn.Dropout after the ReLU, model/model.py:57-58): element e is
  // kept when its 16-bit uniform (two per generator word, eight elements per call) is >= p
  const bool drop = drop_counter != nullptr && drop_p > 0.f;
  const unsigned long long dseed = drop_seed, dcounter = drop ? drop_counter[1] : 0ull;
  const unsigned int drop_thr = (unsigned int)(drop_p * 65536.0f);
  const float keep_scale = drop ? 1.0f / (1.0f - drop_p) : 1.0f;
  for (int c0 = threadIdx.x * 8; c0 < C; c0 += DP_THREADS * 8) {
    float v[8], rsv[8];
    load8<BF16>(x, (size_t)b * C + c0, v);
    if (gs >= 8) {
      const float rs = vqaf_gn_fwd8(v, lpg, gs, eps);  // v is normalised on return
#pragma unroll
      for (int i = 0; i < 8; ++i) rsv[i] = 1.f;
      if ((threadIdx.x % lpg) == 0) rstd[(size_t)b * groups + c0 / gs] = rs;
    } else {
      // several groups inside one lane's 8 channels (gs in {1,2,4}): segmented butterflies
      float s[8], q[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) s[i] = v[i];
      for (int o = 1; o < gs; o <<= 1) {
        float t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = s[i ^ o];
#pragma unroll
        for (int i = 0; i < 8; ++i) s[i] += t[i];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        v[i] -= s[i] / (float)gs;
        q[i] = v[i] * v[i];
      }
      for (int o = 1; o < gs; o <<= 1) {
        float t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = q[i ^ o];
#pragma unroll
        for (int i = 0; i < 8; ++i) q[i] += t[i];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        rsv[i] = rsqrtf(q[i] / (float)gs + eps);
        if ((i % gs) == 0) rstd[(size_t)b * groups + (c0 + i) / gs] = rsv[i];
      }
    }
    float ga[8], be[8], yo[8];
    load8<BF16>(gamma, (size_t)c0, ga);
    load8<BF16>(beta, (size_t)c0, be);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v[i] *= rsv[i];
      yo[i] = fmaf(v[i], ga[i], be[i]);
      if (relu) yo[i] = fmaxf(yo[i], 0.f);
    }
    if (drop) {
      const Philox4 r = philox4x32_10(dseed, ((size_t)b * C + c0) >> 3, dcounter);
      const unsigned int w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        yo[2 * i] = (w[i] & 0xffffu) >= drop_thr ? yo[2 * i] * keep_scale : 0.f;
        yo[2 * i + 1] = (w[i] >> 16) >= drop_thr ? yo[2 * i + 1] * keep_scale : 0.f;
      }
    }
    store8<BF16>(y, (size_t)b * C + c0, yo);
    store8<BF16>(xhat, (size_t)b * C + c0, v);
  }
}

template <bool BF16>
__global__ void __launch_bounds__(DP_THREADS)
groupnorm_bwd_kernel(const void* __restrict__ grad_y, const void* __restrict__ y,
                     const void* __restrict__ xhat, const float* __restrict__ rstd,
                     const void* __restrict__ gamma, int C, int groups, int relu, float out_scale,
                     void* __restrict__ g_pre, void* __restrict__ grad_x) {
  const int b = blockIdx.x;
  const int gs = C / groups, lpg = gs >= 8 ? gs / 8 : 1;
  for (int c0 = threadIdx.x * 8; c0 < C; c0 += DP_THREADS * 8) {
    float gy[8], xh[8], ga[8], yo[8];
    load8<BF16>(grad_y, (size_t)b * C + c0, gy);
    load8<BF16>(xhat, (size_t)b * C + c0, xh);
    load8<BF16>(gamma, (size_t)c0, ga);
    if (relu) {
      load8<BF16>(y, (size_t)b * C + c0, yo);
#pragma unroll
      // y > 0 is the ReLU gate AND the dropout keep mask (dropped outputs are exactly zero);
      // out_scale = 1 / (1 - p) of the inverted dropout, 1 without it
      for (int i = 0; i < 8; ++i) gy[i] = yo[i] > 0.f ? gy[i] * out_scale : 0.f;
    }
    store8<BF16>(g_pre, (size_t)b * C + c0, gy);
    float dx[8];
    if (gs >= 8) {
      vqaf_gn_bwd8(gy, xh, ga, lpg, gs, rstd[(size_t)b * groups + c0 / gs], dx);
    } else {
      float s1[8], s2[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        dx[i] = gy[i] * ga[i];
        s1[i] = dx[i];
        s2[i] = dx[i] * xh[i];
      }
      for (int o = 1; o < gs; o <<= 1) {
        float t1[8], t2[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { t1[i] = s1[i ^ o]; t2[i] = s2[i ^ o]; }
#pragma unroll
        for (int i = 0; i < 8; ++i) { s1[i] += t1[i]; s2[i] += t2[i]; }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float rs = rstd[(size_t)b * groups + (c0 + i) / gs];
        dx[i] = rs * (dx[i] - s1[i] / (float)gs - xh[i] * (s2[i] / (float)gs));
      }
    }
    store8<BF16>(grad_x, (size_t)b * C + c0, dx);
  }
}

// One wave per row of logits.  Rows of up to 512 classes are read once and stay in registers for the
// three uses (maximum, normaliser, gradient).  With loss_mean the launch also returns the mean of the
// per-row losses: the last workgroup to finish (a ticket counter that it resets) adds the B values in
// a fixed order, so the mean does not depend on which workgroup that was.
constexpr int CE_REG = 8;
template <bool BF16, bool GBF16>
__global__ void __launch_bounds__(DP_THREADS)
ce_conf_kernel(const void* __restrict__ logits, const int64_t* __restrict__ targets, int B, int C,
               float* __restrict__ loss, float* __restrict__ conf, int64_t* __restrict__ pred,
               void* __restrict__ grad_logits, long ldg, float grad_scale, float* __restrict__ loss_mean,
               unsigned int* __restrict__ ticket) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * (DP_THREADS / 64) + (threadIdx.x >> 6);
  const bool row = b < B, cached = C <= 64 * CE_REG;
  int64_t t = -1;
  float v[CE_REG];
  float m = -INFINITY, z = 0.f;
  if (row) {
    t = targets != nullptr ? targets[b] : -1;  // requested first: nothing below waits on it alone
    float xt = 0.f;  // the target's logit (cached rows: picked out of the registers)
    int am = 0x7fffffff;  // lowest index wins among equal maxima
    if (cached) {
#pragma unroll
      for (int i = 0; i < CE_REG; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < C ? load1<BF16>(logits, (size_t)b * C + c) : -INFINITY;
      }
#pragma unroll
      for (int i = 0; i < CE_REG; ++i) {
        if (v[i] > m) { m = v[i]; am = lane + 64 * i; }
        if ((int64_t)(lane + 64 * i) == t) xt = v[i];
      }
      xt = wave_sum(xt);  // one lane holds it, the others hold zero
    } else {
      for (int c = lane; c < C; c += 64) {
        const float x = load1<BF16>(logits, (size_t)b * C + c);
        if (x > m) { m = x; am = c; }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float om = __shfl_xor(m, o, 64);
      const int oa = __shfl_xor(am, o, 64);
      if (om > m || (om == m && oa < am)) { m = om; am = oa; }
    }
    if (cached) {
#pragma unroll
      for (int i = 0; i < CE_REG; ++i) {
        v[i] = expf(v[i] - m);  // exp(-inf) = 0 for the columns past C
        z += v[i];
      }
    } else {
      for (int c = lane; c < C; c += 64) z += expf(load1<BF16>(logits, (size_t)b * C + c) - m);
    }
    z = wave_sum(z);
    if (lane == 0) {
      if (loss != nullptr) {
        if (t >= 0 && t < C) loss[b] = (m + logf(z)) - (cached ? xt : load1<BF16>(logits, (size_t)b * C + t));
        else if (loss_mean != nullptr) loss[b] = NAN;
      }
      if (conf != nullptr) conf[b] = 1.0f / z;
      if (pred != nullptr) pred[b] = (int64_t)(am == 0x7fffffff ? 0 : am);
    }
  }
  // The ticket is taken as soon as the workgroup's losses are written, BEFORE its gradient rows: the fence
  // then waits for four floats, not for the rows, and the last workgroup adds up while rows are still landing.
  __shared__ int s_last;
  if (loss_mean != nullptr) {  // uniform
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      s_last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
  }
  if (row && grad_logits != nullptr) {
    if (cached) {
#pragma unroll
      for (int i = 0; i < CE_REG; ++i) {
        const int c = lane + 64 * i;
        if (c < C) store1<GBF16>(grad_logits, (size_t)b * ldg + c, grad_scale * (v[i] / z - ((int64_t)c == t ? 1.f : 0.f)));
      }
    } else {
      for (int c = lane; c < C; c += 64) {
        const float p = expf(load1<BF16>(logits, (size_t)b * C + c) - m) / z;
        store1<GBF16>(grad_logits, (size_t)b * ldg + c, grad_scale * (p - ((int64_t)c == t ? 1.f : 0.f)));
      }
    }
    for (int c = C + lane; c < ldg; c += 64) store1<GBF16>(grad_logits, (size_t)b * ldg + c, 0.f);  // row padding
  }
  if (loss_mean == nullptr) return;  // uniform
  if (s_last) {
    // thread t adds rows t, t + 256, ... (four requests in flight at a time), waves and threads are
    // combined in index order
    __shared__ float s_part[DP_THREADS / 64];
    __threadfence();
    float s = 0.f;
    int i = threadIdx.x;
    for (; i + 3 * DP_THREADS < B; i += 4 * DP_THREADS) {
      const float x0 = __hip_atomic_load(loss + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float x1 = __hip_atomic_load(loss + i + DP_THREADS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float x2 = __hip_atomic_load(loss + i + 2 * DP_THREADS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float x3 = __hip_atomic_load(loss + i + 3 * DP_THREADS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s += (x0 + x1) + (x2 + x3);
    }
    for (; i < B; i += DP_THREADS) s += __hip_atomic_load(loss + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s = wave_sum(s);
    if (lane == 0) s_part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int w = 0; w < DP_THREADS / 64; ++w) t += s_part[w];
      *loss_mean = t / (float)B;
      *ticket = 0u;
    }
  }
}

// out[i] = in[i] * scale[0] in the requested dtype: the chain-rule factor of a loss whose logits
// gradient the forward already produced (one pass, eight elements per lane).
template <bool BF16>
__global__ void __launch_bounds__(DP_THREADS)
scale_cast_kernel(const float* __restrict__ in, const float* __restrict__ scale, size_t n, void* __restrict__ out) {
  const float s = scale[0];
  const size_t n8 = n / 8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    float v[8];
    load8<false>(in, i * 8, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] *= s;
    store8<BF16>(out, i * 8, v);
  }
  for (size_t i = n8 * 8 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    store1<BF16>(out, i, in[i] * s);
}

bool gn_shape_ok(int C, int groups) {
  if (groups <= 0 || C % groups != 0) return false;
  const int gs = C / groups;
  if (C % 8 != 0) return false;
  if (gs < 8) return gs == 1 || gs == 2 || gs == 4;
  if (gs % 8 != 0) return false;
  const int lpg = gs / 8;
  return lpg <= 64 && (lpg & (lpg - 1)) == 0;
}

}  // namespace

#define VQAF_BY_DTYPE(kernel, grid, block, st, ...)                                         \
  do {                                                                                      \
    if (dtype == VQAF_BF16) hipLaunchKernelGGL(kernel<true>, grid, block, 0, st, __VA_ARGS__); \
    else hipLaunchKernelGGL(kernel<false>, grid, block, 0, st, __VA_ARGS__);                   \
  } while (0)

extern "C" int vqaf_ghost_sqnorm_linear(const void* a, const void* g, int B, int In, int Out,
                                        int dtype, int has_bias, int accumulate, float* sq,
                                        vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(a && g && sq, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_dtype_ok(dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(B > 0 && In > 0 && Out > 0, VQAF_E_SHAPE);
  VQAF_BY_DTYPE(ghost_sqnorm_linear_kernel, dim3(B), dim3(DP_THREADS), (hipStream_t)stream, a, g,
                In, Out, has_bias, accumulate, sq);
  VQAF_LAUNCH_CHECK();
}

extern "C" int vqaf_ghost_sqnorm_affine(const void* xhat, const void* g, int B, int C, int dtype,
                                        int accumulate, float* sq, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(xhat && g && sq, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_dtype_ok(dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(B > 0 && C > 0, VQAF_E_SHAPE);
  VQAF_BY_DTYPE(ghost_sqnorm_affine_kernel, dim3(B), dim3(DP_THREADS), (hipStream_t)stream, xhat,
                g, C, accumulate, sq);
  VQAF_LAUNCH_CHECK();
}

extern "C" int vqaf_clip_coef(const float* sq, int B, float max_norm, float scale, float* coef,
                              float* norms, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(sq && coef, VQAF_E_NULL);
  VQAF_REQUIRE(B > 0, VQAF_E_SHAPE);
  hipLaunchKernelGGL(clip_coef_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     sq, B, max_norm, scale, coef, norms);
  VQAF_LAUNCH_CHECK();
}

static int ghost_clip_launch(int n_layers, const void* const* a_host, const void* const* g_host,
                             const int* in_dim_host, const int* out_dim_host, const int* kind_host,
                             int B, int dtype, float max_norm, float scale, int scale_g_inplace,
                             float* coef, float* norms, float* sq, const GhostZero& Z, vqaf_stream_t stream) {
  VQAF_REQUIRE(a_host && g_host && in_dim_host && out_dim_host && kind_host && coef, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_dtype_ok(dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(B > 0 && n_layers > 0 && n_layers <= GC_MAX_LAYERS, VQAF_E_SHAPE);
  GhostLayers L;
  L.n = n_layers;
  for (int l = 0; l < n_layers; ++l) {
    VQAF_REQUIRE(a_host[l] && g_host[l], VQAF_E_NULL);
    VQAF_REQUIRE(in_dim_host[l] > 0 && out_dim_host[l] > 0 && kind_host[l] >= 0 && kind_host[l] <= 2,
                 VQAF_E_SHAPE);
    if (kind_host[l] == 2)
      VQAF_REQUIRE(out_dim_host[l] % 8 == 0 && vqaf_aligned16(a_host[l]) && vqaf_aligned16(g_host[l]),
                   VQAF_E_ALIGN);
    L.a[l] = a_host[l];
    L.g[l] = g_host[l];
    L.in_dim[l] = in_dim_host[l];
    L.out_dim[l] = out_dim_host[l];
    L.kind[l] = kind_host[l];
  }
  // bf16 rows that are 16-byte aligned and fit one 8-element piece per thread: the register-resident form
  bool rows_fit = dtype == VQAF_BF16;
  for (int l = 0; l < n_layers && rows_fit; ++l) {
    const int in = kind_host[l] == 2 ? out_dim_host[l] : in_dim_host[l], out = out_dim_host[l];
    rows_fit = in % 8 == 0 && out % 8 == 0 && in <= DP_THREADS * 8 && out <= DP_THREADS * 8 &&
               vqaf_aligned16(a_host[l]) && vqaf_aligned16(g_host[l]);
  }
  if (rows_fit)
    hipLaunchKernelGGL(ghost_clip_rows_kernel, dim3(B), dim3(DP_THREADS), 0, (hipStream_t)stream, L,
                       max_norm, scale, scale_g_inplace, coef, norms, sq, Z);
  else if (dtype == VQAF_BF16)
    hipLaunchKernelGGL(ghost_clip_kernel<true>, dim3(B), dim3(DP_THREADS), 0, (hipStream_t)stream, L,
                       max_norm, scale, scale_g_inplace, coef, norms, sq, Z);
  else
    hipLaunchKernelGGL(ghost_clip_kernel<false>, dim3(B), dim3(DP_THREADS), 0, (hipStream_t)stream, L,
                       max_norm, scale, scale_g_inplace, coef, norms, sq, Z);
  VQAF_LAUNCH_CHECK();
}

extern "C" int vqaf_ghost_clip(int n_layers, const void* const* a_host, const void* const* g_host,
                               const int* in_dim_host, const int* out_dim_host, const int* kind_host,
                               int B, int dtype, float max_norm, float scale, int scale_g_inplace,
                               float* coef, float* norms, float* sq, vqaf_stream_t stream) {
  VQAF_ENTRY();
  GhostZero Z = {};
  return ghost_clip_launch(n_layers, a_host, g_host, in_dim_host, out_dim_host, kind_host, B, dtype, max_norm, scale,
                           scale_g_inplace, coef, norms, sq, Z, stream);
}

extern "C" int vqaf_ghost_clip_zero(int n_layers, const void* const* a_host, const void* const* g_host,
                                    const int* in_dim_host, const int* out_dim_host, const int* kind_host,
                                    int B, int dtype, float max_norm, float scale, int scale_g_inplace,
                                    float* coef, float* norms, float* sq, float* const* zero_ptrs_host,
                                    const size_t* zero_counts_host, int n_zero, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(n_zero >= 0 && n_zero <= GC_MAX_ZERO, VQAF_E_SHAPE);
  VQAF_REQUIRE(n_zero == 0 || (zero_ptrs_host && zero_counts_host), VQAF_E_NULL);
  GhostZero Z = {};
  for (int i = 0; i < n_zero; ++i) {
    VQAF_REQUIRE(zero_ptrs_host[i] != nullptr, VQAF_E_NULL);
    VQAF_REQUIRE(zero_counts_host[i] < (1ull << 32), VQAF_E_SHAPE);
    Z.p[i] = zero_ptrs_host[i];
    Z.n[i] = (unsigned int)zero_counts_host[i];
  }
  Z.count = n_zero;
  return ghost_clip_launch(n_layers, a_host, g_host, in_dim_host, out_dim_host, kind_host, B, dtype, max_norm, scale,
                           scale_g_inplace, coef, norms, sq, Z, stream);
}

extern "C" int vqaf_ghost_clip_zero_bump(int n_layers, const void* const* a_host, const void* const* g_host,
                                         const int* in_dim_host, const int* out_dim_host, const int* kind_host,
                                         int B, int dtype, float max_norm, float scale, int scale_g_inplace,
                                         float* coef, float* norms, float* sq, float* const* zero_ptrs_host,
                                         const size_t* zero_counts_host, int n_zero, uint64_t* count_state,
                                         vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(n_zero >= 0 && n_zero <= GC_MAX_ZERO, VQAF_E_SHAPE);
  VQAF_REQUIRE(n_zero == 0 || (zero_ptrs_host && zero_counts_host), VQAF_E_NULL);
  GhostZero Z = {};
  for (int i = 0; i < n_zero; ++i) {
    VQAF_REQUIRE(zero_ptrs_host[i] != nullptr, VQAF_E_NULL);
    VQAF_REQUIRE(zero_counts_host[i] < (1ull << 32), VQAF_E_SHAPE);
    Z.p[i] = zero_ptrs_host[i];
    Z.n[i] = (unsigned int)zero_counts_host[i];
  }
  Z.count = n_zero;
  Z.bump = reinterpret_cast<unsigned long long*>(count_state);
  return ghost_clip_launch(n_layers, a_host, g_host, in_dim_host, out_dim_host, kind_host, B, dtype, max_norm, scale,
                           scale_g_inplace, coef, norms, sq, Z, stream);
}

extern "C" int vqaf_clipped_sum_linear(const void* a, const void* g, const float* coef, int B,
                                       int In, int Out, int dtype, int accumulate, float* SW,
                                       float* Sb, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(a && g && SW, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_dtype_ok(dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(B > 0 && In > 0 && Out > 0, VQAF_E_SHAPE);
  if (dtype == VQAF_BF16) {
    dim3 grid((In + CB_TN - 1) / CB_TN, (Out + CB_TM - 1) / CB_TM);
    VQAF_REQUIRE(grid.y <= 65535, VQAF_E_SHAPE);
    // few output tiles: split the batch over workgroups (2..8 ways, >= 2 stages each)
    int splits = 1;
    const int tiles = (int)(grid.x * grid.y);
    while (tiles * splits < 128 && splits < 8 && B / (splits * 2) >= 2 * CB_KB) splits *= 2;
    const int rows_per_split = ((B + splits - 1) / splits + CB_KB - 1) / CB_KB * CB_KB;
    grid.z = splits;
    const bf16_t* ab = reinterpret_cast<const bf16_t*>(a);
    const bf16_t* gb = reinterpret_cast<const bf16_t*>(g);
    auto vec_of = [](const void* p, int pitch) {
      if (pitch % 8 == 0 && vqaf_aligned16(p)) return 8;
      if (pitch % 2 == 0 && (reinterpret_cast<uintptr_t>(p) & 3u) == 0) return 2;
      return 1;
    };
    const int vg = vec_of(g, Out), va = vec_of(a, In);
    const size_t cb_smem = (size_t)2 * CB_KB * (CB_PG + CB_PA) * sizeof(bf16_t);  // 80 KiB
    hipStream_t st = (hipStream_t)stream;
    if (splits > 1 && !accumulate) {  // atomics add into the output: start it from zero
      if (hipMemsetAsync(SW, 0, (size_t)Out * In * 4, st) != hipSuccess) return VQAF_E_LAUNCH;
      if (Sb != nullptr && hipMemsetAsync(Sb, 0, (size_t)Out * 4, st) != hipSuccess) return VQAF_E_LAUNCH;
    }
#define VQAF_CSB(VG, VA)                                                                              \
  do {                                                                                                \
    static std::once_flag attr_once;                                                                  \
    std::call_once(attr_once, [&] {                                                                   \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&clipped_sum_linear_bf16_kernel<VG, VA>), \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)cb_smem);            \
    });                                                                                               \
    hipLaunchKernelGGL((clipped_sum_linear_bf16_kernel<VG, VA>), grid, dim3(DP_THREADS), cb_smem, st, \
                       ab, gb, coef, B, In, Out, accumulate, SW, Sb, rows_per_split);                 \
  } while (0)
    if (vg == 8 && va == 8) VQAF_CSB(8, 8);
    else if (vg == 2 && va == 8) VQAF_CSB(2, 8);
    else if (vg == 8 && va == 2) VQAF_CSB(8, 2);
    else if (vg == 2 && va == 2) VQAF_CSB(2, 2);
    else VQAF_CSB(1, 1);
#undef VQAF_CSB
    VQAF_LAUNCH_CHECK();
  }
  if (coef == nullptr && B >= 64) {
    // rows already scaled: a plain G^T A, which the pipelined f32 MFMA GEMM (vector loads along
    // the contiguous index, two stages in flight) runs about three times faster than the kernel
    // below; the bias column sums follow in a launch of their own
    const int rc = vqaf_gemm_f32(g, VQAF_F32, 0, 1, Out, a, VQAF_F32, 0, In, 1, nullptr, VQAF_F32, 0, SW, 0, In,
                                 1, Out, In, B, 1.0f, accumulate, stream);
    if (rc != VQAF_OK || Sb == nullptr) return rc;
    if (!accumulate && hipMemsetAsync(Sb, 0, (size_t)Out * 4, (hipStream_t)stream) != hipSuccess) return VQAF_E_LAUNCH;
    hipLaunchKernelGGL(colsum_f32_kernel, dim3((Out + 63) / 64, (B + COLSUM_ROWS - 1) / COLSUM_ROWS),
                       dim3(DP_THREADS), 0, (hipStream_t)stream, reinterpret_cast<const float*>(g), B, Out, Sb);
    VQAF_LAUNCH_CHECK();
  }
  const dim3 grid((In + CS_T - 1) / CS_T, (Out + CS_T - 1) / CS_T);
  VQAF_REQUIRE(grid.y <= 65535, VQAF_E_SHAPE);
  hipLaunchKernelGGL(clipped_sum_linear_kernel<false>, grid, dim3(DP_THREADS), 0,
                     (hipStream_t)stream, a, g, coef, B, In, Out, accumulate, SW, Sb);
  VQAF_LAUNCH_CHECK();
}

extern "C" int vqaf_clipped_sum_affine(const void* xhat, const void* g, const float* coef, int B,
                                       int C, int dtype, int accumulate, float* Sgamma,
                                       float* Sbeta, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(xhat && g, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_dtype_ok(dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(B > 0 && C > 0, VQAF_E_SHAPE);
  VQAF_REQUIRE(C % 8 == 0 && vqaf_aligned16(xhat) && vqaf_aligned16(g), VQAF_E_ALIGN);
  hipStream_t st = (hipStream_t)stream;
  if (!accumulate) {
    if (Sgamma != nullptr && Sbeta == Sgamma + C) {  // adjacent slices of one flat bucket: one memset
      if (hipMemsetAsync(Sgamma, 0, (size_t)2 * C * 4, st) != hipSuccess) return VQAF_E_LAUNCH;
    } else {
      if (Sgamma != nullptr && hipMemsetAsync(Sgamma, 0, (size_t)C * 4, st) != hipSuccess) return VQAF_E_LAUNCH;
      if (Sbeta != nullptr && hipMemsetAsync(Sbeta, 0, (size_t)C * 4, st) != hipSuccess) return VQAF_E_LAUNCH;
    }
  }
  const dim3 grid((C + 127) / 128, (B + AF_ROWS - 1) / AF_ROWS);
  VQAF_REQUIRE(grid.y <= 65535, VQAF_E_SHAPE);
  VQAF_BY_DTYPE(clipped_sum_affine_kernel, grid, dim3(DP_THREADS), st, xhat, g, coef, B, C, Sgamma,
                Sbeta);
  VQAF_LAUNCH_CHECK();
}

extern "C" int vqaf_clipped_sum_affine_multi(int n, const void* const* xhat, const void* const* g,
                                             const float* coef, int B, const int* C, int dtype,
                                             float* const* Sgamma, float* const* Sbeta, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(xhat && g && C && Sgamma && Sbeta, VQAF_E_NULL);
  VQAF_REQUIRE(n >= 1 && n <= AF_MAX && B > 0, VQAF_E_SHAPE);
  VQAF_REQUIRE(vqaf_dtype_ok(dtype), VQAF_E_DTYPE);
  AffineLayers L = {};
  int widest = 0;
  for (int i = 0; i < n; ++i) {
    VQAF_REQUIRE(xhat[i] && g[i], VQAF_E_NULL);
    VQAF_REQUIRE(C[i] > 0, VQAF_E_SHAPE);
    VQAF_REQUIRE(C[i] % 8 == 0 && vqaf_aligned16(xhat[i]) && vqaf_aligned16(g[i]), VQAF_E_ALIGN);
    L.xhat[i] = xhat[i]; L.g[i] = g[i]; L.Sgamma[i] = Sgamma[i]; L.Sbeta[i] = Sbeta[i]; L.C[i] = C[i];
    widest = C[i] > widest ? C[i] : widest;
  }
  const dim3 grid((widest + 127) / 128, (B + AF_ROWS - 1) / AF_ROWS, n);
  VQAF_REQUIRE(grid.y <= 65535, VQAF_E_SHAPE);
  VQAF_BY_DTYPE(clipped_sum_affine_multi_kernel, grid, dim3(DP_THREADS), (hipStream_t)stream, L, coef, B);
  VQAF_LAUNCH_CHECK();
}

namespace {
constexpr int ZM_MAX = 8;
struct ZeroRanges {
  float* p[ZM_MAX];
  unsigned long long n[ZM_MAX];  // elements
  int count;
};
// blockIdx.y picks the range; 16-byte stores where the range allows.
__global__ void __launch_bounds__(DP_THREADS) zero_multi_kernel(ZeroRanges R) {
  float* p = nullptr;
  unsigned long long n = 0;
#pragma unroll
  for (int i = 0; i < ZM_MAX; ++i)
    if (i == (int)blockIdx.y) { p = R.p[i]; n = R.n[i]; }
  const bool vec = (reinterpret_cast<uintptr_t>(p) & 15u) == 0;
  const unsigned long long n4 = vec ? n / 4 : 0;
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
    reinterpret_cast<f32x4*>(p)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (unsigned long long i = 4 * n4 + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    p[i] = 0.f;
}
}  // namespace

extern "C" int vqaf_zero_multi(float* const* ptrs_host, const size_t* counts_host, int n_ranges,
                               vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(ptrs_host && counts_host, VQAF_E_NULL);
  VQAF_REQUIRE(n_ranges > 0 && n_ranges <= ZM_MAX, VQAF_E_SHAPE);
  ZeroRanges R = {};
  size_t longest = 0;
  for (int i = 0; i < n_ranges; ++i) {
    VQAF_REQUIRE(ptrs_host[i] != nullptr, VQAF_E_NULL);
    R.p[i] = ptrs_host[i];
    R.n[i] = counts_host[i];
    longest = counts_host[i] > longest ? counts_host[i] : longest;
  }
  R.count = n_ranges;
  size_t blocks = (longest / 4 + DP_THREADS - 1) / DP_THREADS;
  blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
  hipLaunchKernelGGL(zero_multi_kernel, dim3((unsigned)blocks, n_ranges), dim3(DP_THREADS), 0, (hipStream_t)stream, R);
  VQAF_LAUNCH_CHECK();
}

extern "C" int vqaf_noise_scale(const float* S, const float* noise, const uint64_t* rng_state,
                                float noise_std, float scale, size_t n, float* grad,
                                vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(S && grad, VQAF_E_NULL);
  VQAF_REQUIRE(n > 0, VQAF_E_SHAPE);
  VQAF_REQUIRE(noise != nullptr || noise_std == 0.f || rng_state != nullptr, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_aligned16(S) && vqaf_aligned16(grad) && (noise == nullptr || vqaf_aligned16(noise)),
               VQAF_E_ALIGN);
  const size_t ncall = (n + 3) / 4;
  size_t blocks = (ncall + DP_THREADS - 1) / DP_THREADS;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(noise_scale_kernel, dim3((unsigned)blocks), dim3(DP_THREADS), 0,
                     (hipStream_t)stream, S, noise, rng_state, noise_std, scale, n, grad);
  VQAF_LAUNCH_CHECK();
}

extern "C" int vqaf_dp_adamw(const float* S, const float* noise, const uint64_t* rng_state,
                             float noise_std, float scale, float* param, float* exp_avg,
                             float* exp_avg_sq, void* shadow_bf16, float* grad_out, size_t n, float lr,
                             float beta1, float beta2, float eps, float weight_decay,
                             uint64_t* step, int step_advance, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(S && param && exp_avg && exp_avg_sq && step, VQAF_E_NULL);
  VQAF_REQUIRE(n > 0 && step_advance >= -1, VQAF_E_SHAPE);
  VQAF_REQUIRE(noise != nullptr || noise_std == 0.f || rng_state != nullptr, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_aligned16(S) && vqaf_aligned16(param) && vqaf_aligned16(exp_avg) &&
                   vqaf_aligned16(exp_avg_sq) && (noise == nullptr || vqaf_aligned16(noise)) &&
                   (grad_out == nullptr || vqaf_aligned16(grad_out)) &&
                   (shadow_bf16 == nullptr || (reinterpret_cast<uintptr_t>(shadow_bf16) & 7u) == 0),
               VQAF_E_ALIGN);
  const size_t ncall = (n + 3) / 4;
  size_t blocks = (ncall + DP_THREADS - 1) / DP_THREADS;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(dp_adamw_kernel, dim3((unsigned)blocks), dim3(DP_THREADS), 0, (hipStream_t)stream,
                     S, noise, rng_state, noise_std, scale, param, exp_avg, exp_avg_sq,
                     reinterpret_cast<bf16_t*>(shadow_bf16), grad_out, n, lr, beta1, beta2, eps,
                     weight_decay, step, step_advance);
  VQAF_LAUNCH_CHECK();
}

extern "C" int vqaf_dp_adamw_ranges(int n_ranges, const long* offsets_host, const size_t* counts_host, const float* S,
                                    const uint64_t* rng_state, long noise_lo, float noise_std, float scale,
                                    float* param, float* exp_avg, float* exp_avg_sq, void* shadow_bf16,
                                    float* grad_out, float lr, float beta1, float beta2, float eps,
                                    float weight_decay, const uint64_t* step, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(offsets_host && counts_host && S && param && exp_avg && exp_avg_sq && step, VQAF_E_NULL);
  VQAF_REQUIRE(n_ranges > 0 && n_ranges <= DA_MAX_RANGES, VQAF_E_SHAPE);
  VQAF_REQUIRE(noise_std == 0.f || rng_state != nullptr, VQAF_E_NULL);
  AdamRanges R = {};
  size_t longest = 0;
  for (int i = 0; i < n_ranges; ++i) {
    VQAF_REQUIRE(offsets_host[i] >= noise_lo && (offsets_host[i] - noise_lo) % 4 == 0, VQAF_E_ALIGN);
    VQAF_REQUIRE(counts_host[i] > 0 && counts_host[i] < (1ull << 31), VQAF_E_SHAPE);
    R.off[i] = offsets_host[i];
    R.n[i] = (unsigned int)counts_host[i];
    longest = counts_host[i] > longest ? counts_host[i] : longest;
  }
  R.count = n_ranges;
  size_t blocks = ((longest + 3) / 4 + DP_THREADS - 1) / DP_THREADS;
  blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
  hipLaunchKernelGGL(dp_adamw_ranges_kernel, dim3((unsigned)blocks, n_ranges), dim3(DP_THREADS), 0, (hipStream_t)stream,
                     R, S, rng_state, noise_lo, noise_std, scale, param, exp_avg, exp_avg_sq,
                     reinterpret_cast<bf16_t*>(shadow_bf16), grad_out, lr, beta1, beta2, eps, weight_decay, step);
  VQAF_LAUNCH_CHECK();
}

extern "C" int vqaf_groupnorm_fwd(const void* x, const void* gamma, const void* beta, int B, int C,
                                  int groups, float eps, int relu, int dtype, void* y, void* xhat,
                                  float* rstd, float dropout_p, uint64_t dropout_seed,
                                  const uint64_t* dropout_counter, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(x && gamma && beta && y && xhat && rstd, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_dtype_ok(dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(B > 0 && C > 0 && gn_shape_ok(C, groups), VQAF_E_SHAPE);
  VQAF_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, VQAF_E_SHAPE);
  VQAF_REQUIRE(dropout_p == 0.f || (dropout_counter != nullptr && relu), VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_aligned16(x) && vqaf_aligned16(y) && vqaf_aligned16(xhat) &&
                   vqaf_aligned16(gamma) && vqaf_aligned16(beta), VQAF_E_ALIGN);
  VQAF_BY_DTYPE(groupnorm_fwd_kernel, dim3(B), dim3(DP_THREADS), (hipStream_t)stream, x, gamma,
                beta, C, groups, eps, relu, y, xhat, rstd, dropout_p, (unsigned long long)dropout_seed,
                dropout_p > 0.f ? dropout_counter : nullptr);
  VQAF_LAUNCH_CHECK();
}

extern "C" int vqaf_groupnorm_bwd(const void* grad_y, const void* y, const void* xhat,
                                  const float* rstd, const void* gamma, int B, int C, int groups,
                                  int relu, float out_scale, int dtype, void* g_pre, void* grad_x,
                                  vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(grad_y && xhat && rstd && gamma && g_pre && grad_x, VQAF_E_NULL);
  VQAF_REQUIRE(!relu || y != nullptr, VQAF_E_NULL);
  VQAF_REQUIRE(relu || out_scale == 1.0f, VQAF_E_SHAPE);
  VQAF_REQUIRE(vqaf_dtype_ok(dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(B > 0 && C > 0 && gn_shape_ok(C, groups), VQAF_E_SHAPE);
  VQAF_REQUIRE(vqaf_aligned16(grad_y) && vqaf_aligned16(xhat) && vqaf_aligned16(g_pre) &&
                   vqaf_aligned16(grad_x) && vqaf_aligned16(gamma), VQAF_E_ALIGN);
  VQAF_BY_DTYPE(groupnorm_bwd_kernel, dim3(B), dim3(DP_THREADS), (hipStream_t)stream, grad_y, y,
                xhat, rstd, gamma, C, groups, relu, out_scale, g_pre, grad_x);
  VQAF_LAUNCH_CHECK();
}

static int ce_launch(const void* logits, const int64_t* targets, int B, int C, int dtype, float* loss, float* conf,
                     int64_t* pred, void* grad_logits, long ldg, int grad_dtype, float grad_scale, float* loss_mean,
                     unsigned int* ticket, hipStream_t st) {
  VQAF_REQUIRE(logits != nullptr, VQAF_E_NULL);
  VQAF_REQUIRE((loss == nullptr && grad_logits == nullptr) || targets != nullptr, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_dtype_ok(dtype) && (grad_logits == nullptr || vqaf_dtype_ok(grad_dtype)), VQAF_E_DTYPE);
  VQAF_REQUIRE(B > 0 && C > 0 && (grad_logits == nullptr || ldg >= C), VQAF_E_SHAPE);
  const int rows_per_block = DP_THREADS / 64;
  const dim3 grid((B + rows_per_block - 1) / rows_per_block), block(DP_THREADS);
#define VQAF_CE(LB, GB) hipLaunchKernelGGL((ce_conf_kernel<LB, GB>), grid, block, 0, st, logits, targets, B, C, loss, conf, pred, grad_logits, ldg, grad_scale, loss_mean, ticket)
  const bool gb = grad_logits != nullptr && grad_dtype == VQAF_BF16;
  if (dtype == VQAF_BF16) { if (gb) VQAF_CE(true, true); else VQAF_CE(true, false); }
  else { if (gb) VQAF_CE(false, true); else VQAF_CE(false, false); }
#undef VQAF_CE
  VQAF_LAUNCH_CHECK();
}

extern "C" int vqaf_ce_conf(const void* logits, const int64_t* targets, int B, int C, int dtype,
                            float* loss, float* conf, int64_t* pred, void* grad_logits, int grad_dtype,
                            float grad_scale, vqaf_stream_t stream) {
  VQAF_ENTRY();
  return ce_launch(logits, targets, B, C, dtype, loss, conf, pred, grad_logits, C, grad_dtype, grad_scale, nullptr,
                   nullptr, (hipStream_t)stream);
}

extern "C" int vqaf_ce_mean(const void* logits, const int64_t* targets, int B, int C, int dtype, float* loss,
                            void* grad_logits, long ld_grad, int grad_dtype, float grad_scale, float* loss_mean,
                            unsigned int* ticket, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(loss != nullptr && loss_mean != nullptr && ticket != nullptr && targets != nullptr, VQAF_E_NULL);
  return ce_launch(logits, targets, B, C, dtype, loss, nullptr, nullptr, grad_logits, ld_grad, grad_dtype, grad_scale,
                   loss_mean, ticket, (hipStream_t)stream);
}

extern "C" int vqaf_scale_cast(const float* in, const float* scale, size_t n, int out_dtype, void* out,
                               vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(in && scale && out, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_dtype_ok(out_dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(n > 0, VQAF_E_SHAPE);
  VQAF_REQUIRE(vqaf_aligned16(in) && vqaf_aligned16(out), VQAF_E_ALIGN);
  size_t blocks = (n / 8 + DP_THREADS - 1) / DP_THREADS;
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  const int dtype = out_dtype;
  VQAF_BY_DTYPE(scale_cast_kernel, dim3((unsigned)blocks), dim3(DP_THREADS), (hipStream_t)stream, in, scale, n, out);
  VQAF_LAUNCH_CHECK();
}

extern "C" int vqaf_version(void) {
  VQAF_ENTRY(); return VQAF_VERSION; }

extern "C" const char* vqaf_status_string(int status) {
  switch (status) {
    case VQAF_OK: return "ok";
    case VQAF_E_SHAPE: return "unsupported or inconsistent shape";
    case VQAF_E_DTYPE: return "unsupported dtype";
    case VQAF_E_ALIGN: return "pointer or row pitch not 16-byte aligned";
    case VQAF_E_LAUNCH: return "HIP launch failed";
    case VQAF_E_NULL: return "required pointer is NULL";
    case VQAF_E_SCRATCH: return "sizes exceed the fixed device scratch of this entry point (use the variant that takes a workspace)";
    default: return "unknown status";
  }
}
