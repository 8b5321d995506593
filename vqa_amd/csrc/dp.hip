// DP-SGD step kernels for the classifier: per-sample ("ghost") gradient norms, clip
// coefficients, clipped gradient sums, Gaussian noise + scaling, GroupNorm forward/backward.
//
// Per-sample gradients are never materialised.  For y = x W^T + b on 2-D input the
// per-sample weight gradient is the rank-1 matrix g_i a_i^T (g_i = dL_i/dy_i, a_i = x_i), so
//   ||grad_i W||_F^2 = ||a_i||^2 ||g_i||^2,       sum_i c_i grad_i W = G^T diag(c) A,
// i.e. two row-norm passes plus one (Out x B) x (B x In) GEMM whose contraction index is the
// batch.  That GEMM runs on the matrix cores: operands are converted to f32 while they are
// staged, so the f32 path is exact f32 (v_mfma_f32_32x32x2_f32 is a k-ordered fmaf chain).
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
        if (m0 + col < Out) gv = coef[bb] * load1<BF16>(g, (size_t)bb * Out + m0 + col);
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

// Sgamma[c] (+)= sum_b coef[b] xhat[b,c] g[b,c];  Sbeta[c] (+)= sum_b coef[b] g[b,c].
template <bool BF16>
__global__ void __launch_bounds__(DP_THREADS)
clipped_sum_affine_kernel(const void* __restrict__ xhat, const void* __restrict__ g,
                          const float* __restrict__ coef, int B, int C, int accumulate,
                          float* __restrict__ Sgamma, float* __restrict__ Sbeta) {
  __shared__ float r0[4][64], r1[4][64];
  const int col = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + col;
  float sg = 0.f, sb = 0.f;
  if (c < C) {
    for (int b = rg; b < B; b += 4) {
      const float gv = coef[b] * load1<BF16>(g, (size_t)b * C + c);
      sg = fmaf(load1<BF16>(xhat, (size_t)b * C + c), gv, sg);
      sb += gv;
    }
  }
  r0[rg][col] = sg;
  r1[rg][col] = sb;
  __syncthreads();
  if (threadIdx.x < 64 && c < C) {
    const float vg = (r0[0][col] + r0[1][col]) + (r0[2][col] + r0[3][col]);
    const float vb = (r1[0][col] + r1[1][col]) + (r1[2][col] + r1[3][col]);
    if (Sgamma != nullptr) Sgamma[c] = accumulate ? Sgamma[c] + vg : vg;
    if (Sbeta != nullptr) Sbeta[c] = accumulate ? Sbeta[c] + vb : vb;
  }
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

// GroupNorm on (B,C): one workgroup per sample, a lane owns 8 consecutive channels, a group of
// gs = C/groups channels spans gs/8 adjacent lanes (a power of two <= 64).
template <bool BF16>
__global__ void __launch_bounds__(DP_THREADS)
groupnorm_fwd_kernel(const void* __restrict__ x, const void* __restrict__ gamma,
                     const void* __restrict__ beta, int C, int groups, float eps, int relu,
                     void* __restrict__ y, void* __restrict__ xhat, float* __restrict__ rstd) {
  const int b = blockIdx.x;
  const int gs = C / groups, lpg = gs >= 8 ? gs / 8 : 1;
  for (int c0 = threadIdx.x * 8; c0 < C; c0 += DP_THREADS * 8) {
    float v[8], rsv[8];
    load8<BF16>(x, (size_t)b * C + c0, v);
    if (gs >= 8) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) s += v[i];
      for (int o = lpg >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      const float mean = s / (float)gs;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        v[i] -= mean;
        q = fmaf(v[i], v[i], q);
      }
      for (int o = lpg >> 1; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
      const float rs = rsqrtf(q / (float)gs + eps);
#pragma unroll
      for (int i = 0; i < 8; ++i) rsv[i] = rs;
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
    store8<BF16>(y, (size_t)b * C + c0, yo);
    store8<BF16>(xhat, (size_t)b * C + c0, v);
  }
}

template <bool BF16>
__global__ void __launch_bounds__(DP_THREADS)
groupnorm_bwd_kernel(const void* __restrict__ grad_y, const void* __restrict__ y,
                     const void* __restrict__ xhat, const float* __restrict__ rstd,
                     const void* __restrict__ gamma, int C, int groups, int relu,
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
      for (int i = 0; i < 8; ++i) gy[i] = yo[i] > 0.f ? gy[i] : 0.f;
    }
    store8<BF16>(g_pre, (size_t)b * C + c0, gy);
    float dx[8];
    if (gs >= 8) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        dx[i] = gy[i] * ga[i];
        s1 += dx[i];
        s2 = fmaf(dx[i], xh[i], s2);
      }
      for (int o = lpg >> 1; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64);
      }
      const float rs = rstd[(size_t)b * groups + c0 / gs];
      const float m1 = s1 / (float)gs, m2 = s2 / (float)gs;
#pragma unroll
      for (int i = 0; i < 8; ++i) dx[i] = rs * (dx[i] - m1 - xh[i] * m2);
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

// One wave per row of logits.
template <bool BF16>
__global__ void __launch_bounds__(DP_THREADS)
ce_conf_kernel(const void* __restrict__ logits, const int64_t* __restrict__ targets, int B, int C,
               float* __restrict__ loss, float* __restrict__ conf, int64_t* __restrict__ pred,
               float* __restrict__ grad_logits, float grad_scale) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * (DP_THREADS / 64) + (threadIdx.x >> 6);
  if (b >= B) return;
  float m = -INFINITY;
  int am = 0x7fffffff;  // lowest index wins among equal maxima
  for (int c = lane; c < C; c += 64) {
    const float v = load1<BF16>(logits, (size_t)b * C + c);
    if (v > m) { m = v; am = c; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(m, o, 64);
    const int oa = __shfl_xor(am, o, 64);
    if (om > m || (om == m && oa < am)) { m = om; am = oa; }
  }
  float z = 0.f;
  for (int c = lane; c < C; c += 64) z += expf(load1<BF16>(logits, (size_t)b * C + c) - m);
  z = wave_sum(z);
  const int64_t t = targets != nullptr ? targets[b] : -1;
  if (lane == 0) {
    if (loss != nullptr && t >= 0 && t < C)
      loss[b] = (m + logf(z)) - load1<BF16>(logits, (size_t)b * C + t);
    if (conf != nullptr) conf[b] = 1.0f / z;
    if (pred != nullptr) pred[b] = (int64_t)(am == 0x7fffffff ? 0 : am);
  }
  if (grad_logits != nullptr) {
    for (int c = lane; c < C; c += 64) {
      const float p = expf(load1<BF16>(logits, (size_t)b * C + c) - m) / z;
      grad_logits[(size_t)b * C + c] = grad_scale * (p - ((int64_t)c == t ? 1.f : 0.f));
    }
  }
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
  VQAF_REQUIRE(a && g && sq, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_dtype_ok(dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(B > 0 && In > 0 && Out > 0, VQAF_E_SHAPE);
  VQAF_BY_DTYPE(ghost_sqnorm_linear_kernel, dim3(B), dim3(DP_THREADS), (hipStream_t)stream, a, g,
                In, Out, has_bias, accumulate, sq);
  VQAF_LAUNCH_CHECK();
}

extern "C" int vqaf_ghost_sqnorm_affine(const void* xhat, const void* g, int B, int C, int dtype,
                                        int accumulate, float* sq, vqaf_stream_t stream) {
  VQAF_REQUIRE(xhat && g && sq, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_dtype_ok(dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(B > 0 && C > 0, VQAF_E_SHAPE);
  VQAF_BY_DTYPE(ghost_sqnorm_affine_kernel, dim3(B), dim3(DP_THREADS), (hipStream_t)stream, xhat,
                g, C, accumulate, sq);
  VQAF_LAUNCH_CHECK();
}

extern "C" int vqaf_clip_coef(const float* sq, int B, float max_norm, float scale, float* coef,
                              float* norms, vqaf_stream_t stream) {
  VQAF_REQUIRE(sq && coef, VQAF_E_NULL);
  VQAF_REQUIRE(B > 0, VQAF_E_SHAPE);
  hipLaunchKernelGGL(clip_coef_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     sq, B, max_norm, scale, coef, norms);
  VQAF_LAUNCH_CHECK();
}

extern "C" int vqaf_clipped_sum_linear(const void* a, const void* g, const float* coef, int B,
                                       int In, int Out, int dtype, int accumulate, float* SW,
                                       float* Sb, vqaf_stream_t stream) {
  VQAF_REQUIRE(a && g && coef && SW, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_dtype_ok(dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(B > 0 && In > 0 && Out > 0, VQAF_E_SHAPE);
  const dim3 grid((In + CS_T - 1) / CS_T, (Out + CS_T - 1) / CS_T);
  VQAF_REQUIRE(grid.y <= 65535, VQAF_E_SHAPE);
  VQAF_BY_DTYPE(clipped_sum_linear_kernel, grid, dim3(DP_THREADS), (hipStream_t)stream, a, g, coef,
                B, In, Out, accumulate, SW, Sb);
  VQAF_LAUNCH_CHECK();
}

extern "C" int vqaf_clipped_sum_affine(const void* xhat, const void* g, const float* coef, int B,
                                       int C, int dtype, int accumulate, float* Sgamma,
                                       float* Sbeta, vqaf_stream_t stream) {
  VQAF_REQUIRE(xhat && g && coef, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_dtype_ok(dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(B > 0 && C > 0, VQAF_E_SHAPE);
  VQAF_BY_DTYPE(clipped_sum_affine_kernel, dim3((C + 63) / 64), dim3(DP_THREADS),
                (hipStream_t)stream, xhat, g, coef, B, C, accumulate, Sgamma, Sbeta);
  VQAF_LAUNCH_CHECK();
}

extern "C" int vqaf_noise_scale(const float* S, const float* noise, const uint64_t* rng_state,
                                float noise_std, float scale, size_t n, float* grad,
                                vqaf_stream_t stream) {
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

extern "C" int vqaf_groupnorm_fwd(const void* x, const void* gamma, const void* beta, int B, int C,
                                  int groups, float eps, int relu, int dtype, void* y, void* xhat,
                                  float* rstd, vqaf_stream_t stream) {
  VQAF_REQUIRE(x && gamma && beta && y && xhat && rstd, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_dtype_ok(dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(B > 0 && C > 0 && gn_shape_ok(C, groups), VQAF_E_SHAPE);
  VQAF_REQUIRE(vqaf_aligned16(x) && vqaf_aligned16(y) && vqaf_aligned16(xhat) &&
                   vqaf_aligned16(gamma) && vqaf_aligned16(beta), VQAF_E_ALIGN);
  VQAF_BY_DTYPE(groupnorm_fwd_kernel, dim3(B), dim3(DP_THREADS), (hipStream_t)stream, x, gamma,
                beta, C, groups, eps, relu, y, xhat, rstd);
  VQAF_LAUNCH_CHECK();
}

extern "C" int vqaf_groupnorm_bwd(const void* grad_y, const void* y, const void* xhat,
                                  const float* rstd, const void* gamma, int B, int C, int groups,
                                  int relu, int dtype, void* g_pre, void* grad_x,
                                  vqaf_stream_t stream) {
  VQAF_REQUIRE(grad_y && xhat && rstd && gamma && g_pre && grad_x, VQAF_E_NULL);
  VQAF_REQUIRE(!relu || y != nullptr, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_dtype_ok(dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(B > 0 && C > 0 && gn_shape_ok(C, groups), VQAF_E_SHAPE);
  VQAF_REQUIRE(vqaf_aligned16(grad_y) && vqaf_aligned16(xhat) && vqaf_aligned16(g_pre) &&
                   vqaf_aligned16(grad_x) && vqaf_aligned16(gamma), VQAF_E_ALIGN);
  VQAF_BY_DTYPE(groupnorm_bwd_kernel, dim3(B), dim3(DP_THREADS), (hipStream_t)stream, grad_y, y,
                xhat, rstd, gamma, C, groups, relu, g_pre, grad_x);
  VQAF_LAUNCH_CHECK();
}

extern "C" int vqaf_ce_conf(const void* logits, const int64_t* targets, int B, int C, int dtype,
                            float* loss, float* conf, int64_t* pred, float* grad_logits,
                            float grad_scale, vqaf_stream_t stream) {
  VQAF_REQUIRE(logits != nullptr, VQAF_E_NULL);
  VQAF_REQUIRE((loss == nullptr && grad_logits == nullptr) || targets != nullptr, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_dtype_ok(dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(B > 0 && C > 0, VQAF_E_SHAPE);
  const int rows_per_block = DP_THREADS / 64;
  VQAF_BY_DTYPE(ce_conf_kernel, dim3((B + rows_per_block - 1) / rows_per_block), dim3(DP_THREADS),
                (hipStream_t)stream, logits, targets, B, C, loss, conf, pred, grad_logits,
                grad_scale);
  VQAF_LAUNCH_CHECK();
}

extern "C" int vqaf_version(void) { return VQAF_VERSION; }

extern "C" const char* vqaf_status_string(int status) {
  switch (status) {
    case VQAF_OK: return "ok";
    case VQAF_E_SHAPE: return "unsupported or inconsistent shape";
    case VQAF_E_DTYPE: return "unsupported dtype";
    case VQAF_E_ALIGN: return "pointer or row pitch not 16-byte aligned";
    case VQAF_E_LAUNCH: return "HIP launch failed";
    case VQAF_E_NULL: return "required pointer is NULL";
    default: return "unknown status";
  }
}
