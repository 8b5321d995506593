// bf16 MFMA GEMM for encoder-size Linear layers: Y (M,N) = X (M,K) W (N,K)^T with both operands
// k-contiguous, f32 accumulation, bf16 output, and the layer's elementwise neighbour as the epilogue:
//   forward            y  = x W^T + b                      (model/vision_encoder.py:83-127, model/text_encoder.py:14-48:
//   forward + GELU     h  = x W^T + b,  y = gelu(h)         the Linear layers of the ViT / BERT blocks)
//   input gradient     gx = gy W                            (W (out,in) as it is stored: its rows are the contraction
//   ... + GELU back    gh = (gy W) * gelu'(h)                index, so its image is k-major and read transposing)
// The tokens of a batch make M large (64 x 197 = 12608 rows), so unlike the classifier's GEMMs these are
// MFMA-bound: tiles of 128 x 128 outputs, four waves of 64 x 64 (four 32x32x16 MFMAs per pair of fragment
// reads from each operand), operands HBM/L2 -> LDS by LDS-DMA in 128-byte rows with an XOR swizzle on the
// source address (LDS-DMA writes wave-linear bytes) and the same swizzle on the fragment reads.  One
// k-step of 64 is resident per workgroup; three workgroups per CU overlap each other's fetches.
// Output tiles go through LDS so that every store instruction writes whole 128-byte row pieces, and the
// epilogue works on 8 consecutive channels per lane.

#include "common.h"

namespace {

constexpr int LN_BM = 128, LN_BN = 128, LN_BK = 64, LN_THREADS = 256;
constexpr int LN_OPER_BYTES = LN_BM * LN_BK * 2;   // one operand's k-step image: 128 rows x 128 bytes
constexpr int LN_CP = 136;                         // row pitch (bytes) of a wave's 64 x 64 bf16 output image
constexpr int LN_LDS = 4 * 64 * LN_CP;             // 34816 >= the two operand images (32768)

typedef __bf16 ln_bf16x8 __attribute__((ext_vector_type(8)));

enum { LN_EPI_BIAS = 0, LN_EPI_GELU = 1, LN_EPI_GELU_BWD = 2 };

struct LinArgs {
  const bf16_t* X;
  const bf16_t* W;
  const void* bias;   // (N) or NULL
  bf16_t* Y;          // (M,N): the Linear output (pre-activation), or the gradient
  bf16_t* Y2;         // LN_EPI_GELU: gelu(Y)
  const bf16_t* H;    // LN_EPI_GELU_BWD: pre-activation of the forward
  const unsigned int* zeros;  // 16 bytes of zeros: the source of chunks outside a k-major operand
  long ldx, ldw, ldy, ld2;
  int M, N, K, bias_bf16, tiles_m, tiles_n;
};

__device__ __forceinline__ u32x2 ln_read_tr(const unsigned char* p) {
  typedef short ln_s16x4 __attribute__((ext_vector_type(4)));
  const ln_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) ln_s16x4*)(const_cast<unsigned char*>(p)));
  union { ln_s16x4 s; u32x2 u; } c;
  c.s = v;
  return c.u;
}

__device__ __forceinline__ void ln_glds16(const void* src, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
// chunk c (8 k values) of row r of a [rows][64 k] image is stored at chunk position c ^ ln_swz(r): the 16
// rows that one ds_read_b128 lane group touches then fall on 16 distinct 16-byte slots of the bank row
__device__ __forceinline__ int ln_swz(int r) { return (r >> 1) & 7; }

__device__ __forceinline__ float ln_gelu(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }
__device__ __forceinline__ float ln_gelu_grad(float v) {
  const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * __expf(-0.5f * v * v);
  return cdf + v * pdf;
}

template <int EPI, bool W_KMAJOR>
__global__ __launch_bounds__(LN_THREADS, 3) void linear_nt_kernel(const LinArgs g) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[LN_LDS];
  // Workgroup ids go round-robin over the 8 XCDs; each XCD gets one contiguous run of tiles (columns
  // fastest), so the workgroups sharing an L2 work on a few row blocks of X and all of W.
  const int nt = g.tiles_m * g.tiles_n;
  const int xcd = blockIdx.x & 7, q = nt >> 3, r = nt & 7;
  const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
  const int tile_m = lin / g.tiles_n, tile_n = lin - tile_m * g.tiles_n;
  const int m0 = tile_m * LN_BM, n0 = tile_n * LN_BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // ---- LDS-DMA sources: chunk q = j * 256 + tid -> (row q / 8, chunk position q % 8)
  //      (k-major W: chunk -> (k row q / 16, position q % 16) holding source chunk position ^ ((k & 3) << 2))
  const bf16_t* xs[4];
  const bf16_t* ws[4];
  bool w_ok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int qq = j * LN_THREADS + tid, row = qq >> 3, c = (qq & 7) ^ ln_swz(row);
    xs[j] = g.X + (long)min(m0 + row, g.M - 1) * g.ldx + c * 8;   // rows past the edge repeat the last row;
    if constexpr (W_KMAJOR) {
      const int k = qq >> 4, cw = (qq & 15) ^ ((k & 3) << 2);
      w_ok[j] = n0 + cw * 8 < g.N;   // columns past the edge come from the block of zeros
      ws[j] = g.W + (long)k * g.ldw + (w_ok[j] ? n0 + cw * 8 : 0);
    } else {
      w_ok[j] = true;
      ws[j] = g.W + (long)min(n0 + row, g.N - 1) * g.ldw + c * 8;   // their outputs are never stored
    }
  }
  const long w_step = W_KMAJOR ? (long)LN_BK * g.ldw : (long)LN_BK;
  unsigned char* const dma_x = lds + wave * 1024;
  unsigned char* const dma_w = lds + LN_OPER_BYTES + wave * 1024;

  // ---- fragment addresses: lane reads row (lane & 31) of a 32-row block, k chunk 2 ks + (lane >> 5)
  int xr[2], wr[2];
  unsigned wa[2];   // k-major W: byte address of the first transposing read of k16 step 0
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    xr[i] = wm * 64 + i * 32 + (lane & 31);
    wr[i] = wn * 64 + i * 32 + (lane & 31);
    const int li = lane & 15, g16 = lane >> 4, krow = 8 * (g16 >> 1) + (li >> 2);
    const int cb = (wn * 64 + i * 32 + 16 * (g16 & 1) + 4 * (li & 3)) * 2;
    wa[i] = (unsigned)(LN_OPER_BYTES + krow * 256 + (cb ^ ((krow & 3) << 6)));
  }
  const int khalf = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int T = g.K / LN_BK;
  for (int t = 0; t < T; ++t) {
#pragma unroll
    for (int j = 0; j < 4; ++j) ln_glds16(xs[j] + t * LN_BK, dma_x + j * 4096);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const void* src = ws[j] + (long)t * w_step;
      if constexpr (W_KMAJOR) src = w_ok[j] ? src : reinterpret_cast<const void*>(g.zeros);
      ln_glds16(src, dma_w + j * 4096);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      ln_bf16x8 xf[2], wf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        xf[i] = *reinterpret_cast<const ln_bf16x8*>(lds + xr[i] * 128 + (((2 * ks + khalf) ^ ln_swz(xr[i])) << 4));
        if constexpr (W_KMAJOR) {
          const unsigned a = wa[i] + ks * 16 * 256;
          const u32x2 lo = ln_read_tr(lds + a), hi = ln_read_tr(lds + a + 4 * 256);
          union { u32x4 u; ln_bf16x8 b; } c;
          c.u = u32x4{lo.x, lo.y, hi.x, hi.y};
          wf[i] = c.b;
        } else {
          wf[i] = *reinterpret_cast<const ln_bf16x8*>(lds + LN_OPER_BYTES + wr[i] * 128 +
                                                      (((2 * ks + khalf) ^ ln_swz(wr[i])) << 4));
        }
      }
      // W fragments in the A slot: the accumulator then holds, per lane, one token (column lane & 31) and
      // runs of 4 consecutive output channels, which pack into 8-byte LDS writes in the epilogue
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- epilogue.  acc[i][j][e]: token m = wm*64 + i*32 + (lane & 31),
  //      channel n = wn*64 + j*32 + (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
  unsigned char* const img = lds + wave * (64 * LN_CP);   // this wave's 64 tokens x 64 channels, bf16
  // Everything the epilogue reads from memory is requested here in one batch (one round trip instead of
  // one per use): the lane's 8 runs of 4 bias values, and for the GELU backward its 8 pieces of the
  // pre-activation.
  float bv[8][4];
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[q][e] = 0.f;
  if (EPI != LN_EPI_GELU_BWD && g.bias != nullptr) {
    if (g.bias_bf16) {
      u32x2 raw[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int n = min(n0 + wn * 64 + (q >> 2) * 32 + 8 * (q & 3) + 4 * khalf, g.N - 4);
        raw[q] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(g.bias) + n);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        bv[q][0] = bf16_lo(raw[q].x); bv[q][1] = bf16_hi(raw[q].x);
        bv[q][2] = bf16_lo(raw[q].y); bv[q][3] = bf16_hi(raw[q].y);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int n = min(n0 + wn * 64 + (q >> 2) * 32 + 8 * (q & 3) + 4 * khalf, g.N - 4);
        const f32x4 bb = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(g.bias) + n);
        bv[q][0] = bb.x; bv[q][1] = bb.y; bv[q][2] = bb.z; bv[q][3] = bb.w;
      }
    }
  }
  u32x4 hv[8];
  if constexpr (EPI == LN_EPI_GELU_BWD) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int m = min(m0 + wm * 64 + it * 8 + (lane >> 3), g.M - 1);
      const int n = min(n0 + wn * 64 + 8 * (lane & 7), g.N - 8);
      hv[it] = *reinterpret_cast<const u32x4*>(g.H + (long)m * g.ld2 + n);
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const int nl = j * 32 + 8 * rq + 4 * khalf;   // channel within the wave's 64
      const float* b = bv[j * 4 + rq];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ml = i * 32 + (lane & 31);
        u32x2 pk;
        pk.x = pack_bf16x2(acc[i][j][4 * rq + 0] + b[0], acc[i][j][4 * rq + 1] + b[1]);
        pk.y = pack_bf16x2(acc[i][j][4 * rq + 2] + b[2], acc[i][j][4 * rq + 3] + b[3]);
        *reinterpret_cast<u32x2*>(img + ml * LN_CP + nl * 2) = pk;
      }
    }
  // the image is private to the wave; LDS operations of one wave complete in order
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int ml = it * 8 + (lane >> 3), c = lane & 7;
    const int m = m0 + wm * 64 + ml, n = n0 + wn * 64 + 8 * c;
    const u32x4 v = *reinterpret_cast<const u32x4*>(img + ml * LN_CP + c * 16);
    if (m < g.M && n < g.N) {
      if constexpr (EPI == LN_EPI_BIAS) {
        *reinterpret_cast<u32x4*>(g.Y + (long)m * g.ldy + n) = v;
      } else if constexpr (EPI == LN_EPI_GELU) {
        *reinterpret_cast<u32x4*>(g.Y + (long)m * g.ldy + n) = v;
        u32x4 o;
        o.x = pack_bf16x2(ln_gelu(bf16_lo(v.x)), ln_gelu(bf16_hi(v.x)));
        o.y = pack_bf16x2(ln_gelu(bf16_lo(v.y)), ln_gelu(bf16_hi(v.y)));
        o.z = pack_bf16x2(ln_gelu(bf16_lo(v.z)), ln_gelu(bf16_hi(v.z)));
        o.w = pack_bf16x2(ln_gelu(bf16_lo(v.w)), ln_gelu(bf16_hi(v.w)));
        *reinterpret_cast<u32x4*>(g.Y2 + (long)m * g.ld2 + n) = o;
      } else {
        const u32x4 h = hv[it];
        u32x4 o;
        o.x = pack_bf16x2(bf16_lo(v.x) * ln_gelu_grad(bf16_lo(h.x)), bf16_hi(v.x) * ln_gelu_grad(bf16_hi(h.x)));
        o.y = pack_bf16x2(bf16_lo(v.y) * ln_gelu_grad(bf16_lo(h.y)), bf16_hi(v.y) * ln_gelu_grad(bf16_hi(h.y)));
        o.z = pack_bf16x2(bf16_lo(v.z) * ln_gelu_grad(bf16_lo(h.z)), bf16_hi(v.z) * ln_gelu_grad(bf16_hi(h.z)));
        o.w = pack_bf16x2(bf16_lo(v.w) * ln_gelu_grad(bf16_lo(h.w)), bf16_hi(v.w) * ln_gelu_grad(bf16_hi(h.w)));
        *reinterpret_cast<u32x4*>(g.Y + (long)m * g.ldy + n) = o;
      }
    }
  }
}

// ----------------------------------------------------------------------------------------------
// Weight gradient of the same layers: C (M,N) = G^T X with G (K,M) and X (K,N) both k-major (the contraction
// index is the token, the slow index of both), plus the bias gradient (column sums of G).  Same tile and
// wave shape as above; the k-major images [64 k][128 columns] are read with the transposing LDS read.
// The output has few tiles (768 x 3072 weights = 144) and a long contraction (12608 tokens), so the token
// range is split over workgroups.  Each writes its f32 partial tile into its own slab of a workspace with
// plain stores (f32 atomics run at a fifth of the store rate on this chip), and a second small kernel adds
// the slabs in a fixed order and rounds to the output type.
struct WgradArgs {
  const bf16_t* G;
  const bf16_t* X;
  float* slabs;              // (splits, M, N) f32 partial products
  float* cs_slabs;           // (splits, M) f32 partial column sums of G, or NULL
  const unsigned int* zeros; // 16 bytes of zeros: the source of chunks outside the matrices
  long ldg, ldx;
  int M, N, K, k_per_split, splits, tiles_m, tiles_n;
};

__global__ __launch_bounds__(LN_THREADS, 3) void wgrad_tn_kernel(const WgradArgs g) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * LN_OPER_BYTES];
  const int nt = g.tiles_m * g.tiles_n, total = nt * g.splits;
  const int xcd = blockIdx.x & 7, q8 = total >> 3, r8 = total & 7;
  const int lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
  const int split = lin / nt, tile = lin - split * nt;
  const int tile_m = tile / g.tiles_n, tile_n = tile - tile_m * g.tiles_n;
  const int m0 = tile_m * LN_BM, n0 = tile_n * LN_BN;
  const int k_lo = split * g.k_per_split, kspan = min(g.K, k_lo + g.k_per_split) - k_lo;
  const int T = (kspan + LN_BK - 1) / LN_BK;   // >= 1: the host sizes the slices so that none is empty
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // ---- LDS-DMA sources: chunk q = j * 256 + tid -> (k row q / 16, chunk position q % 16); chunk c of
  //      k row k is stored at position c ^ ((k & 3) << 2), which spreads the four k rows a transposing
  //      read touches over disjoint banks
  const bf16_t* gs[4];
  const bf16_t* xs[4];
  int kk[4];
  bool g_ok[4], x_ok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int qq = j * LN_THREADS + tid, k = qq >> 4, c = (qq & 15) ^ ((k & 3) << 2);
    kk[j] = k;
    g_ok[j] = m0 + c * 8 < g.M;
    x_ok[j] = n0 + c * 8 < g.N;
    gs[j] = g.G + (long)(k_lo + k) * g.ldg + m0 + c * 8;
    xs[j] = g.X + (long)(k_lo + k) * g.ldx + n0 + c * 8;
  }
  const unsigned long long zeros = reinterpret_cast<unsigned long long>(g.zeros);
  auto pick = [&](bool ok, const bf16_t* p) -> const void* {
    const unsigned long long m = ok ? ~0ull : 0ull;
    return reinterpret_cast<const void*>((reinterpret_cast<unsigned long long>(p) & m) | (zeros & ~m));
  };
  unsigned char* const dma_g = lds + wave * 1024;
  unsigned char* const dma_x = lds + LN_OPER_BYTES + wave * 1024;

  // ---- fragment addresses (k16 step 0; the second read of a fragment is 4 k rows further, step ks adds
  //      16 rows; the swizzle term depends on k & 3 only, so it is the same for all of them)
  const int li = lane & 15, g16 = lane >> 4;
  const int krow = 8 * (g16 >> 1) + (li >> 2);
  unsigned ga[2], xa[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int cg = (wm * 64 + i * 32 + 16 * (g16 & 1) + 4 * (li & 3)) * 2;
    const int cx = (wn * 64 + i * 32 + 16 * (g16 & 1) + 4 * (li & 3)) * 2;
    ga[i] = (unsigned)(krow * 256 + (cg ^ ((krow & 3) << 6)));
    xa[i] = (unsigned)(LN_OPER_BYTES + krow * 256 + (cx ^ ((krow & 3) << 6)));
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // column sums of G over the tokens (bias gradient) on the vector ALU, from the fragments the MFMAs use
  // anyway: a lane holds G(8 tokens, channel lane & 31), one two-element dot product with ones per pair
  float csum[2] = {0.f, 0.f};
  const bool sums = g.cs_slabs != nullptr && tile_n == 0 && wn == 0;   // wave-uniform
  typedef __bf16 ln_bf16x2 __attribute__((ext_vector_type(2)));
  ln_bf16x2 one2;
  one2[0] = (__bf16)1.0f;
  one2[1] = (__bf16)1.0f;
  auto frag = [&](unsigned a) -> u32x4 {
    const u32x2 lo = ln_read_tr(lds + a), hi = ln_read_tr(lds + a + 4 * 256);
    return u32x4{lo.x, lo.y, hi.x, hi.y};
  };
  auto as_frag = [](const u32x4& v) -> ln_bf16x8 {
    union { u32x4 u; ln_bf16x8 b; } c;
    c.u = v;
    return c.b;
  };

  for (int t = 0; t < T; ++t) {
    const int kbase = t * LN_BK;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      ln_glds16(pick(g_ok[j] && kbase + kk[j] < kspan, gs[j] + (long)kbase * g.ldg), dma_g + j * 4096);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      ln_glds16(pick(x_ok[j] && kbase + kk[j] < kspan, xs[j] + (long)kbase * g.ldx), dma_x + j * 4096);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      u32x4 gf[2], xf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        gf[i] = frag(ga[i] + ks * 16 * 256);
        xf[i] = frag(xa[i] + ks * 16 * 256);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(gf[i]), as_frag(xf[j]), acc[i][j], 0, 0, 0);
      if (sums) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const unsigned w[4] = {gf[i].x, gf[i].y, gf[i].z, gf[i].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            union { unsigned u; ln_bf16x2 b; } c;
            c.u = w[e];
            csum[i] = __builtin_amdgcn_fdot2_f32_bf16(c.b, one2, csum[i], false);
          }
        }
      }
    }
    __syncthreads();
  }

  // ---- epilogue: acc[i][j][e] is C(m, n) with m = m0 + wm*64 + i*32 + (e & 3) + 8 (e >> 2) + 4 (lane >> 5),
  //      n = n0 + wn*64 + j*32 + (lane & 31); 32 lanes write 128 contiguous bytes of a slab row
  float* const slab = g.slabs + (size_t)split * g.M * g.N;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
      if (m < g.M) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int n = n0 + wn * 64 + j * 32 + (lane & 31);
          if (n < g.N) slab[(size_t)m * g.N + n] = acc[i][j][e];
        }
      }
    }
  if (sums) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      // lanes l and l + 32 hold the two token halves of channel (l & 31)
      const float tot = csum[i] + __shfl_xor(csum[i], 32);
      const int m = m0 + wm * 64 + i * 32 + (lane & 31);
      if ((lane >> 5) == 0 && m < g.M) g.cs_slabs[(size_t)split * g.M + m] = tot;
    }
  }
}

// out(m, n) = sum over slabs, in slab order; 8 consecutive elements per thread.  The threads past the matrix
// do the same for the bias gradient, one channel each.
template <bool OUT_BF16, bool B_BF16>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* slabs, const float* cs_slabs, int splits, int M,
                                                           int N, void* out, long ld_out, void* bias_out) {
  const size_t n8 = (size_t)M * N / 8, i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n8) {
    const size_t e = i * 8, stride = (size_t)M * N;
    f32x4 a = *reinterpret_cast<const f32x4*>(slabs + e), b = *reinterpret_cast<const f32x4*>(slabs + e + 4);
    for (int s = 1; s < splits; ++s) {
      a += *reinterpret_cast<const f32x4*>(slabs + s * stride + e);
      b += *reinterpret_cast<const f32x4*>(slabs + s * stride + e + 4);
    }
    const size_t m = e / N, n = e - m * N;
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    store8<OUT_BF16>(out, m * ld_out + n, v);
  } else if (cs_slabs != nullptr && i - n8 < (size_t)M) {
    const size_t m = i - n8;
    float t = cs_slabs[m];
    for (int s = 1; s < splits; ++s) t += cs_slabs[(size_t)s * M + m];
    if constexpr (B_BF16) reinterpret_cast<bf16_t*>(bias_out)[m] = f32_to_bf16(t);
    else reinterpret_cast<float*>(bias_out)[m] = t;
  }
}

// how the token range is cut: about 2.5 workgroups per CU (three fit), each with at least 8 k-steps
void wgrad_plan(int M, int N, int K, int* k_per_split, int* splits) {
  const long nt = (long)((M + LN_BM - 1) / LN_BM) * ((N + LN_BN - 1) / LN_BN);
  const int ksteps = (K + LN_BK - 1) / LN_BK;
  long want = nt >= 640 ? 1 : (640 + nt - 1) / nt;
  if (want > ksteps / 8) want = ksteps / 8;
  if (want < 1) want = 1;
  *k_per_split = (int)(((ksteps + want - 1) / want) * LN_BK);
  *splits = (K + *k_per_split - 1) / *k_per_split;
}

bool linear_nt_shape_ok(long ldx, long ldw, long ldy, int M, int N, int K) {
  return M > 0 && N >= 8 && K >= LN_BK && K % LN_BK == 0 && N % 8 == 0 && ldx >= K && ldw >= K && ldy >= N &&
         ldx % 8 == 0 && ldw % 8 == 0 && ldy % 8 == 0 &&
         (long)((M + LN_BM - 1) / LN_BM) * ((N + LN_BN - 1) / LN_BN) < 0x7fffffffL;
}

}  // namespace

extern "C" int vqaf_linear_nt_supported(long ldx, long ldw, int w_kmajor, long ldy, int M, int N, int K) {
  if (!linear_nt_shape_ok(ldx, w_kmajor ? (long)K : ldw, ldy, M, N, K)) return 0;
  return (!w_kmajor || (ldw >= N && ldw % 8 == 0)) ? 1 : 0;
}

template <int EPI>
static void linear_nt_launch(const LinArgs& g, int w_kmajor, dim3 grid, hipStream_t st) {
  if (w_kmajor) hipLaunchKernelGGL((linear_nt_kernel<EPI, true>), grid, dim3(LN_THREADS), 0, st, g);
  else hipLaunchKernelGGL((linear_nt_kernel<EPI, false>), grid, dim3(LN_THREADS), 0, st, g);
}

extern "C" int vqaf_linear_nt(const void* x, long ldx, const void* w, long ldw, int w_kmajor, const void* bias,
                              int bias_dtype, int M, int N, int K, void* y, long ldy, int epilogue, void* aux,
                              long ld_aux, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(x && w && y, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_linear_nt_supported(ldx, ldw, w_kmajor, ldy, M, N, K) == 1, VQAF_E_SHAPE);
  VQAF_REQUIRE(epilogue >= LN_EPI_BIAS && epilogue <= LN_EPI_GELU_BWD, VQAF_E_SHAPE);
  VQAF_REQUIRE(bias == nullptr || vqaf_dtype_ok(bias_dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(vqaf_aligned16(x) && vqaf_aligned16(w) && vqaf_aligned16(y), VQAF_E_ALIGN);
  VQAF_REQUIRE(bias == nullptr || vqaf_aligned16(bias), VQAF_E_ALIGN);
  if (epilogue != LN_EPI_BIAS) {
    VQAF_REQUIRE(aux != nullptr, VQAF_E_NULL);
    VQAF_REQUIRE(ld_aux >= N && ld_aux % 8 == 0 && vqaf_aligned16(aux), VQAF_E_ALIGN);
  }
  VQAF_REQUIRE(epilogue != LN_EPI_GELU_BWD || bias == nullptr, VQAF_E_SHAPE);
  LinArgs g;
  VQAF_REQUIRE(vqaf_internal_zero_block(&g.zeros) == VQAF_OK, VQAF_E_LAUNCH);
  g.X = reinterpret_cast<const bf16_t*>(x);
  g.W = reinterpret_cast<const bf16_t*>(w);
  g.bias = bias;
  g.bias_bf16 = bias_dtype == VQAF_BF16;
  g.Y = reinterpret_cast<bf16_t*>(y);
  g.Y2 = epilogue == LN_EPI_GELU ? reinterpret_cast<bf16_t*>(aux) : nullptr;
  g.H = epilogue == LN_EPI_GELU_BWD ? reinterpret_cast<const bf16_t*>(aux) : nullptr;
  g.ldx = ldx; g.ldw = ldw; g.ldy = ldy; g.ld2 = ld_aux;
  g.M = M; g.N = N; g.K = K;
  g.tiles_m = (M + LN_BM - 1) / LN_BM;
  g.tiles_n = (N + LN_BN - 1) / LN_BN;
  const dim3 grid((unsigned)(g.tiles_m * g.tiles_n));
  hipStream_t st = (hipStream_t)stream;
  if (epilogue == LN_EPI_BIAS) linear_nt_launch<LN_EPI_BIAS>(g, w_kmajor, grid, st);
  else if (epilogue == LN_EPI_GELU) linear_nt_launch<LN_EPI_GELU>(g, w_kmajor, grid, st);
  else linear_nt_launch<LN_EPI_GELU_BWD>(g, w_kmajor, grid, st);
  VQAF_LAUNCH_CHECK();
}

extern "C" int vqaf_wgrad_tn_supported(long ldg, long ldx, long ld_out, int M, int N, int K) {
  return (M >= 8 && N >= 8 && K > 0 && M % 8 == 0 && N % 8 == 0 && ldg >= M && ldx >= N && ld_out >= N &&
          ldg % 8 == 0 && ldx % 8 == 0 && ld_out % 8 == 0 && (size_t)M * N / 8 / 256 < 0x7fffff00UL) ? 1 : 0;
}

// Large outputs with a long token range run on the 256 x 256-tile kernel (wgrad256.hip).
// (measured at 12608 tokens: 3072 x 768 80 against 87 us with the reduction, 2304 x 768 68 against 74;
// 768 x 768, nine tiles cut into 28 slices, is faster on the 128 x 128 tiles)
static bool wgrad_use256(int M, int N, int K) {
  return M >= 256 && N >= 256 && (long)M * N >= (1L << 20) && K >= 4096;
}

extern "C" size_t vqaf_wgrad_tn_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  int kps, splits;
  if (wgrad_use256(M, N, K)) vqaf_internal_wgrad256_plan(M, N, K, &kps, &splits);
  else wgrad_plan(M, N, K, &kps, &splits);
  return (size_t)splits * ((size_t)M * N + M) * sizeof(float);
}

extern "C" int vqaf_wgrad_tn(const void* gy, long ldg, const void* x, long ldx, int M, int N, int K, void* gw,
                             long ld_out, int gw_dtype, void* gb, int gb_dtype, void* workspace,
                             size_t workspace_bytes, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(gy && x && gw && workspace, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_wgrad_tn_supported(ldg, ldx, ld_out, M, N, K) == 1, VQAF_E_SHAPE);
  VQAF_REQUIRE(vqaf_dtype_ok(gw_dtype) && (gb == nullptr || vqaf_dtype_ok(gb_dtype)), VQAF_E_DTYPE);
  VQAF_REQUIRE(vqaf_aligned16(gy) && vqaf_aligned16(x) && vqaf_aligned16(gw) && vqaf_aligned16(workspace),
               VQAF_E_ALIGN);
  VQAF_REQUIRE(workspace_bytes >= vqaf_wgrad_tn_workspace_bytes(M, N, K), VQAF_E_SHAPE);
  WgradArgs g;
  hipStream_t st = (hipStream_t)stream;
  g.slabs = reinterpret_cast<float*>(workspace);
  if (wgrad_use256(M, N, K)) {
    vqaf_internal_wgrad256_plan(M, N, K, &g.k_per_split, &g.splits);
    g.cs_slabs = gb != nullptr ? g.slabs + (size_t)g.splits * M * N : nullptr;
    const int rc = vqaf_internal_wgrad256_launch(gy, ldg, x, ldx, M, N, K, g.slabs, g.cs_slabs, st);
    if (rc != VQAF_OK) return rc;
  } else {
    VQAF_REQUIRE(vqaf_internal_zero_block(&g.zeros) == VQAF_OK, VQAF_E_LAUNCH);
    wgrad_plan(M, N, K, &g.k_per_split, &g.splits);
    g.G = reinterpret_cast<const bf16_t*>(gy);
    g.X = reinterpret_cast<const bf16_t*>(x);
    g.cs_slabs = gb != nullptr ? g.slabs + (size_t)g.splits * M * N : nullptr;
    g.ldg = ldg; g.ldx = ldx;
    g.M = M; g.N = N; g.K = K;
    g.tiles_m = (M + LN_BM - 1) / LN_BM;
    g.tiles_n = (N + LN_BN - 1) / LN_BN;
    const long nt = (long)g.tiles_m * g.tiles_n;
    VQAF_REQUIRE(nt * g.splits < 0x7fffffffL, VQAF_E_SHAPE);
    hipLaunchKernelGGL(wgrad_tn_kernel, dim3((unsigned)(nt * g.splits)), dim3(LN_THREADS), 0, st, g);
  }
  const size_t work = (size_t)M * N / 8 + (gb != nullptr ? (size_t)M : 0);
  const dim3 rgrid((unsigned)((work + 255) / 256));
  const bool ob = gw_dtype == VQAF_BF16, bb = gb_dtype == VQAF_BF16;
  if (ob && bb) hipLaunchKernelGGL((wgrad_reduce_kernel<true, true>), rgrid, dim3(256), 0, st, g.slabs, g.cs_slabs, g.splits, M, N, gw, ld_out, gb);
  else if (ob) hipLaunchKernelGGL((wgrad_reduce_kernel<true, false>), rgrid, dim3(256), 0, st, g.slabs, g.cs_slabs, g.splits, M, N, gw, ld_out, gb);
  else if (bb) hipLaunchKernelGGL((wgrad_reduce_kernel<false, true>), rgrid, dim3(256), 0, st, g.slabs, g.cs_slabs, g.splits, M, N, gw, ld_out, gb);
  else hipLaunchKernelGGL((wgrad_reduce_kernel<false, false>), rgrid, dim3(256), 0, st, g.slabs, g.cs_slabs, g.splits, M, N, gw, ld_out, gb);
  VQAF_LAUNCH_CHECK();
}
