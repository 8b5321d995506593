// EXPERIMENT, not built into the library: a 256 x 256-tile, one-workgroup-per-CU form of the encoder GEMMs.
// Correct on every shape tried, but slower than the 128 x 128 kernels of vqa_amd/csrc/linear_nt.hip at the
// encoder shapes (603-620 against 673-693 TFLOP/s forward): with K = 768 a tile has 12 k-steps, and with one
// workgroup per CU nothing overlaps its first fetch and its output stage; a persistent tile loop would be
// needed.  Kept for that follow-up.
// 256 x 256-tile bf16 MFMA GEMM for the encoder Linears at training batch sizes (model/vision_encoder.py:83-127,
// model/text_encoder.py:14-48: 64 images x 197 tokens = 12608 rows through 768/2304/3072-wide layers).
// One CU can pull operands through its LDS-DMA path at a limited rate, so what sets the MFMA rate of such a
// GEMM is flops per fetched byte, i.e. the tile a CU works on: 256 x 256 outputs per k-step of 64 fetch
// 64 KB for 8.4 MFLOP, half the bytes per flop of a 128 x 128 tile.  One workgroup of 16 waves (4 x 4 wave
// tiles of 64 x 64, four 32x32x16 MFMAs per k16 step and wave) owns the CU; operands go HBM/L2 -> LDS by
// LDS-DMA into two 64 KB stages, the fetch of k-step t+1 in flight while k-step t is multiplied, one
// barrier per k-step.  D(p,q) = sum_k P(p,k) Q(q,k); either operand is k-contiguous ([rows][64 k] image,
// ds_read_b128 fragments) or k-major ([64 k][256 columns] image, transposing LDS reads), so the same
// kernel serves
//   forward   y(q=token, p=channel)  = x W^T + b [+ GELU]     P = W (k-contiguous), Q = x (k-contiguous)
//   dgrad     gx(q=token, p=in)      = gy W [* gelu'(h)]      P = W (k-major: rows = contraction), Q = gy
//   wgrad     C(p=out, q=in) f32     = gy^T x, + column sums   P = gy (k-major), Q = x (k-major), the
//             token range split over workgroups, partial tiles added with f32 atomics.
#include <mutex>
#include <type_traits>

#include "common.h"

namespace {

constexpr int G2_T = 256, G2_BK = 64, G2_THREADS = 1024;
constexpr int G2_OPER = G2_T * G2_BK * 2;        // 32 KB: one operand, one k-step
constexpr int G2_STAGE = 2 * G2_OPER;            // 64 KB
constexpr int G2_CP = 136;                       // row pitch (bytes) of a wave's 64 x 64 bf16 output image
constexpr int G2_LDS = 16 * 64 * G2_CP;          // 139264 >= two stages (131072)

typedef __bf16 g2_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 g2_bf16x2 __attribute__((ext_vector_type(2)));

enum { G2_EPI_BIAS = 0, G2_EPI_GELU = 1, G2_EPI_GELU_BWD = 2, G2_EPI_F32 = 3 };

struct G2Args {
  const bf16_t* P;
  const bf16_t* Q;
  long ldp, ldq;           // row pitch (elements): of a p/q row when k-contiguous, of a k row when k-major
  int np, nq, K;           // extents of p, q and the contraction
  const void* bias;        // (np) or NULL                                  [EPI 0, 1]
  int bias_bf16;
  bf16_t* Y;               // (nq, np) bf16, pitch ldy                      [EPI 0, 1, 2]
  bf16_t* Y2;              // EPI 1: gelu(Y)
  const bf16_t* H;         // EPI 2: pre-activation (nq, np)
  long ldy, ld2;
  float* C;                // (np, nq) f32, pitch ldc                       [EPI 3]
  float* colsum;           // (np) column sums of P over k, or NULL         [EPI 3]
  long ldc;
  int atomic;              // EPI 3: add into C / colsum
  const unsigned int* zeros;
  int k_per_split, splits, tiles_p, tiles_q;
};

__device__ __forceinline__ void g2_glds16(const void* src, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
__device__ __forceinline__ float g2_gelu(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }
__device__ __forceinline__ float g2_gelu_grad(float v) {
  const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * __expf(-0.5f * v * v);
  return cdf + v * pdf;
}

template <bool P_KMAJOR, bool Q_KMAJOR, int EPI>
__global__ __launch_bounds__(G2_THREADS) void gemm256_kernel(const G2Args g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char g2_lds[];
  // ---- which tile, which slice of the contraction.  Workgroup ids go round-robin over the 8 XCDs; each
  //      XCD gets one contiguous run of (split, q tile, p tile) with p fastest, so the workgroups that
  //      share an L2 read the same few operand slabs.
  const int nt = g.tiles_p * g.tiles_q, total = nt * g.splits;
  const int xcd = blockIdx.x & 7, q8 = total >> 3, r8 = total & 7;
  const int lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
  const int split = lin / nt, tile = lin - split * nt;
  const int tq = tile / g.tiles_p, tp = tile - tq * g.tiles_p;
  const int p0 = tp * G2_T, q0 = tq * G2_T;
  const int k_lo = split * g.k_per_split, kspan = min(g.K, k_lo + g.k_per_split) - k_lo;
  const int T = (kspan + G2_BK - 1) / G2_BK;
  if (T <= 0) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave >> 2, wq = wave & 3;

  // ---- LDS-DMA sources, two 16-byte chunks per thread, operand and k-step: chunk id = j * 1024 + tid.
  //      k-contiguous image [256 rows][128 B]: chunk -> (row id / 8, position id % 8), holding source chunk
  //      position ^ ((row >> 1) & 7).   k-major image [64 k][512 B]: chunk -> (k row id / 32, position
  //      id % 32), holding source chunk position ^ ((k & 3) << 2).  Rows past the end of a k-contiguous
  //      operand repeat its last row (their outputs are never stored); columns past the end of a k-major
  //      operand and k rows past the contraction come from a block of zeros.
  const bf16_t* ps[2];
  const bf16_t* qs[2];
  int pk[2], qk[2];
  bool p_ok[2], q_ok[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int id = j * G2_THREADS + tid;
    if constexpr (P_KMAJOR) {
      const int k = id >> 5, c = (id & 31) ^ ((k & 3) << 2);
      pk[j] = k;
      p_ok[j] = p0 + c * 8 < g.np;
      ps[j] = g.P + (long)(k_lo + k) * g.ldp + p0 + c * 8;
    } else {
      const int row = id >> 3, c = (id & 7) ^ ((row >> 1) & 7);
      pk[j] = c * 8;
      p_ok[j] = true;
      ps[j] = g.P + (long)min(p0 + row, g.np - 1) * g.ldp + k_lo + c * 8;
    }
    if constexpr (Q_KMAJOR) {
      const int k = id >> 5, c = (id & 31) ^ ((k & 3) << 2);
      qk[j] = k;
      q_ok[j] = q0 + c * 8 < g.nq;
      qs[j] = g.Q + (long)(k_lo + k) * g.ldq + q0 + c * 8;
    } else {
      const int row = id >> 3, c = (id & 7) ^ ((row >> 1) & 7);
      qk[j] = c * 8;
      q_ok[j] = true;
      qs[j] = g.Q + (long)min(q0 + row, g.nq - 1) * g.ldq + k_lo + c * 8;
    }
  }
  const long p_step = P_KMAJOR ? (long)G2_BK * g.ldp : (long)G2_BK;
  const long q_step = Q_KMAJOR ? (long)G2_BK * g.ldq : (long)G2_BK;
  const unsigned long long zeros = reinterpret_cast<unsigned long long>(g.zeros);
  auto pick = [&](bool ok, const bf16_t* p) -> const void* {
    const unsigned long long m = ok ? ~0ull : 0ull;
    return reinterpret_cast<const void*>((reinterpret_cast<unsigned long long>(p) & m) | (zeros & ~m));
  };
  auto issue = [&](int t) {
    unsigned char* st = g2_lds + (t & 1) * G2_STAGE + wave * 1024;
    const int kbase = t * G2_BK;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      g2_glds16(pick(p_ok[j] && kbase + pk[j] < kspan, ps[j] + (long)t * p_step), st + j * 16384);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      g2_glds16(pick(q_ok[j] && kbase + qk[j] < kspan, qs[j] + (long)t * q_step), st + G2_OPER + j * 16384);
  };

  // ---- fragment addresses within a stage.  Reads are inline assembly on purpose: for an LDS read it can
  //      see, hipcc waits until every outstanding LDS-DMA has landed, which would serialise the prefetch
  //      with the multiplication.  The vmcnt(0) + barrier at the top of each k-step orders these reads.
  const unsigned lds_base = (unsigned)reinterpret_cast<size_t>((__attribute__((address_space(3))) unsigned char*)g2_lds);
  // k-contiguous: lane reads row (lane & 31) of a 32-row block, chunk (2 ks + (lane >> 5)) ^ swizzle(row)
  //               = base ^ (ks << 5)
  // k-major:      transposing reads at k row 8 (lane >> 5) + ((lane & 15) >> 2) [+4], byte column
  //               2 (block + 16 ((lane >> 4) & 1) + 4 (lane & 3)) ^ ((k & 3) << 6); k16 step ks adds 16 rows
  unsigned pa[2], qa[2];
  {
    const int li = lane & 15, g16 = lane >> 4;
    const int krow = 8 * (g16 >> 1) + (li >> 2);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if constexpr (P_KMAJOR) {
        const int cb = (wp * 64 + i * 32 + 16 * (g16 & 1) + 4 * (li & 3)) * 2;
        pa[i] = (unsigned)(krow * 512 + (cb ^ ((krow & 3) << 6)));
      } else {
        const int row = wp * 64 + i * 32 + (lane & 31);
        pa[i] = (unsigned)(row * 128 + ((((lane >> 5) ^ ((row >> 1) & 7))) << 4));
      }
      if constexpr (Q_KMAJOR) {
        const int cb = (wq * 64 + i * 32 + 16 * (g16 & 1) + 4 * (li & 3)) * 2;
        qa[i] = (unsigned)(G2_OPER + krow * 512 + (cb ^ ((krow & 3) << 6)));
      } else {
        const int row = wq * 64 + i * 32 + (lane & 31);
        qa[i] = (unsigned)(G2_OPER + row * 128 + ((((lane >> 5) ^ ((row >> 1) & 7))) << 4));
      }
    }
  }
  auto read_tr = [&](unsigned addr) -> u32x2 {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
  };
  auto read_b128 = [&](unsigned addr) -> u32x4 {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
  };
  struct Frags {
    u32x4 p[2], q[2];
  };
  auto load_frags = [&](Frags& f, unsigned img, int ks) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if constexpr (P_KMAJOR) {
        const unsigned ad = img + pa[i] + (unsigned)(ks * 16 * 512);
        const u32x2 lo = read_tr(ad), hi = read_tr(ad + 4 * 512);
        f.p[i] = u32x4{lo.x, lo.y, hi.x, hi.y};
      } else {
        f.p[i] = read_b128(img + (pa[i] ^ (unsigned)(ks << 5)));
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if constexpr (Q_KMAJOR) {
        const unsigned ad = img + qa[i] + (unsigned)(ks * 16 * 512);
        const u32x2 lo = read_tr(ad), hi = read_tr(ad + 4 * 512);
        f.q[i] = u32x4{lo.x, lo.y, hi.x, hi.y};
      } else {
        f.q[i] = read_b128(img + (qa[i] ^ (unsigned)(ks << 5)));
      }
    }
  };
  constexpr int READS = (P_KMAJOR ? 4 : 2) + (Q_KMAJOR ? 4 : 2);
  auto as_frag = [](const u32x4& v) -> g2_bf16x8 {
    union { u32x4 u; g2_bf16x8 b; } c;
    c.u = v;
    return c.b;
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // column sums of P over the contraction (bias gradient), on the vector ALU from the fragments the MFMAs
  // use anyway: lane holds P(row lane & 31, 8 k values), one dot product with ones per pair.  The four waves
  // that do it (one per 64-row block of P) sit on four different SIMDs.
  float csum[2] = {0.f, 0.f};
  const bool sums = EPI == G2_EPI_F32 && g.colsum != nullptr && tq == 0 && wq == wp;   // wave-uniform
  g2_bf16x2 one2;
  one2[0] = (__bf16)1.0f;
  one2[1] = (__bf16)1.0f;
  auto mfma_step = [&](const Frags& f) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(f.p[i]), as_frag(f.q[j]), acc[i][j], 0, 0, 0);
    if constexpr (EPI == G2_EPI_F32) {
      if (sums) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const unsigned w[4] = {f.p[i].x, f.p[i].y, f.p[i].z, f.p[i].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            union { unsigned u; g2_bf16x2 b; } c;
            c.u = w[e];
            csum[i] = __builtin_amdgcn_fdot2_f32_bf16(c.b, one2, csum[i], false);
          }
        }
      }
    }
  };
  auto wait_lgkm = [&](auto tag) {
    constexpr int N = decltype(tag)::value;
    if constexpr (N == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
  };
  typedef std::integral_constant<int, READS> R1;
  typedef std::integral_constant<int, 0> R0;

  issue(0);
  for (int t = 0; t < T; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's chunks of k-step t have landed
    __builtin_amdgcn_s_barrier();                      // ... everyone's have, and everyone is done with t-1
    if (t + 1 < T) issue(t + 1);
    const unsigned img = lds_base + (unsigned)((t & 1) * G2_STAGE);
    Frags f0, f1;
    load_frags(f0, img, 0);
    load_frags(f1, img, 1);
    wait_lgkm(R1{});
    __builtin_amdgcn_sched_barrier(0);
    mfma_step(f0);
    __builtin_amdgcn_sched_barrier(0);
    load_frags(f0, img, 2);
    wait_lgkm(R1{});
    __builtin_amdgcn_sched_barrier(0);
    mfma_step(f1);
    __builtin_amdgcn_sched_barrier(0);
    load_frags(f1, img, 3);
    wait_lgkm(R1{});
    __builtin_amdgcn_sched_barrier(0);
    mfma_step(f0);
    wait_lgkm(R0{});
    __builtin_amdgcn_sched_barrier(0);
    mfma_step(f1);
  }

  // ---- epilogue.  acc[i][j][e] = D(p, q) with p = p0 + wp*64 + i*32 + (e & 3) + 8 (e >> 2) + 4 (lane >> 5),
  //      q = q0 + wq*64 + j*32 + (lane & 31)
  const int khalf = lane >> 5;
  if constexpr (EPI == G2_EPI_F32) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int p = p0 + wp * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * khalf;
        if (p < g.np) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int q = q0 + wq * 64 + j * 32 + (lane & 31);
            if (q < g.nq) {
              float* dst = g.C + (long)p * g.ldc + q;
              if (g.atomic) unsafeAtomicAdd(dst, acc[i][j][e]);
              else *dst = acc[i][j][e];
            }
          }
        }
      }
    if (sums) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        // lanes l and l + 32 hold the two k halves of row (l & 31)
        const float tot = csum[i] + __shfl_xor(csum[i], 32);
        const int p = p0 + wp * 64 + i * 32 + (lane & 31);
        if (khalf == 0 && p < g.np) {
          if (g.atomic) unsafeAtomicAdd(g.colsum + p, tot);
          else g.colsum[p] = tot;
        }
      }
    }
  } else {
    __syncthreads();   // every wave is done with the operand stages: they become the output images
    unsigned char* const img = g2_lds + wave * (64 * G2_CP);   // this wave's 64 tokens (q) x 64 channels (p), bf16
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int pl = i * 32 + 8 * rq + 4 * khalf;   // channel within the wave's 64
        float b[4] = {0.f, 0.f, 0.f, 0.f};
        if (EPI != G2_EPI_GELU_BWD && g.bias != nullptr) {
          const int p = min(p0 + wp * 64 + pl, g.np - 4);
          if (g.bias_bf16) {
            const u32x2 bb = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(g.bias) + p);
            b[0] = bf16_lo(bb.x); b[1] = bf16_hi(bb.x); b[2] = bf16_lo(bb.y); b[3] = bf16_hi(bb.y);
          } else {
            const f32x4 bb = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(g.bias) + p);
            b[0] = bb.x; b[1] = bb.y; b[2] = bb.z; b[3] = bb.w;
          }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int ql = j * 32 + (lane & 31);
          u32x2 pkd;
          pkd.x = pack_bf16x2(acc[i][j][4 * rq + 0] + b[0], acc[i][j][4 * rq + 1] + b[1]);
          pkd.y = pack_bf16x2(acc[i][j][4 * rq + 2] + b[2], acc[i][j][4 * rq + 3] + b[3]);
          *reinterpret_cast<u32x2*>(img + ql * G2_CP + pl * 2) = pkd;
        }
      }
    // the image is private to the wave; LDS operations of one wave complete in order
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int ql = it * 8 + (lane >> 3), c = lane & 7;
      const int q = q0 + wq * 64 + ql, p = p0 + wp * 64 + 8 * c;
      const u32x4 v = *reinterpret_cast<const u32x4*>(img + ql * G2_CP + c * 16);
      if (q < g.nq && p < g.np) {
        if constexpr (EPI == G2_EPI_BIAS) {
          *reinterpret_cast<u32x4*>(g.Y + (long)q * g.ldy + p) = v;
        } else if constexpr (EPI == G2_EPI_GELU) {
          *reinterpret_cast<u32x4*>(g.Y + (long)q * g.ldy + p) = v;
          u32x4 o;
          o.x = pack_bf16x2(g2_gelu(bf16_lo(v.x)), g2_gelu(bf16_hi(v.x)));
          o.y = pack_bf16x2(g2_gelu(bf16_lo(v.y)), g2_gelu(bf16_hi(v.y)));
          o.z = pack_bf16x2(g2_gelu(bf16_lo(v.z)), g2_gelu(bf16_hi(v.z)));
          o.w = pack_bf16x2(g2_gelu(bf16_lo(v.w)), g2_gelu(bf16_hi(v.w)));
          *reinterpret_cast<u32x4*>(g.Y2 + (long)q * g.ld2 + p) = o;
        } else {
          const u32x4 h = *reinterpret_cast<const u32x4*>(g.H + (long)q * g.ld2 + p);
          u32x4 o;
          o.x = pack_bf16x2(bf16_lo(v.x) * g2_gelu_grad(bf16_lo(h.x)), bf16_hi(v.x) * g2_gelu_grad(bf16_hi(h.x)));
          o.y = pack_bf16x2(bf16_lo(v.y) * g2_gelu_grad(bf16_lo(h.y)), bf16_hi(v.y) * g2_gelu_grad(bf16_hi(h.y)));
          o.z = pack_bf16x2(bf16_lo(v.z) * g2_gelu_grad(bf16_lo(h.z)), bf16_hi(v.z) * g2_gelu_grad(bf16_hi(h.z)));
          o.w = pack_bf16x2(bf16_lo(v.w) * g2_gelu_grad(bf16_lo(h.w)), bf16_hi(v.w) * g2_gelu_grad(bf16_hi(h.w)));
          *reinterpret_cast<u32x4*>(g.Y + (long)q * g.ldy + p) = o;
        }
      }
    }
  }
}

template <bool PK, bool QK, int EPI>
int g2_launch(const G2Args& g, hipStream_t st) {
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<PK, QK, EPI>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS);
  });
  const dim3 grid((unsigned)(g.tiles_p * g.tiles_q * g.splits));
  hipLaunchKernelGGL((gemm256_kernel<PK, QK, EPI>), grid, dim3(G2_THREADS), G2_LDS, st, g);
  return hipGetLastError() == hipSuccess ? VQAF_OK : VQAF_E_LAUNCH;
}

bool g2_pitch_ok(long ld, long need) { return ld >= need && ld % 8 == 0; }

}  // namespace

extern "C" int vqaf_linear256_supported(long ldx, long ldw, int w_kmajor, long ldy, int M, int N, int K) {
  // y (M,N) = x (M,K) op(w): w is (N,K) k-contiguous (w_kmajor 0) or (K,N) (w_kmajor 1)
  const bool ok = M > 0 && N >= 8 && N % 8 == 0 && K >= 8 && K % 8 == 0 && g2_pitch_ok(ldx, K) &&
                  g2_pitch_ok(ldw, w_kmajor ? N : K) && g2_pitch_ok(ldy, N) &&
                  (long)((M + G2_T - 1) / G2_T) * ((N + G2_T - 1) / G2_T) < 0x7fffffffL;
  return ok ? 1 : 0;
}

extern "C" int vqaf_linear256(const void* x, long ldx, const void* w, long ldw, int w_kmajor, const void* bias,
                              int bias_dtype, int M, int N, int K, void* y, long ldy, int epilogue, void* aux,
                              long ld_aux, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(x && w && y, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_linear256_supported(ldx, ldw, w_kmajor, ldy, M, N, K) == 1, VQAF_E_SHAPE);
  VQAF_REQUIRE(epilogue >= G2_EPI_BIAS && epilogue <= G2_EPI_GELU_BWD, VQAF_E_SHAPE);
  VQAF_REQUIRE(bias == nullptr || vqaf_dtype_ok(bias_dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(vqaf_aligned16(x) && vqaf_aligned16(w) && vqaf_aligned16(y), VQAF_E_ALIGN);
  VQAF_REQUIRE(bias == nullptr || vqaf_aligned16(bias), VQAF_E_ALIGN);
  if (epilogue != G2_EPI_BIAS) {
    VQAF_REQUIRE(aux != nullptr, VQAF_E_NULL);
    VQAF_REQUIRE(ld_aux >= N && ld_aux % 8 == 0 && vqaf_aligned16(aux), VQAF_E_ALIGN);
  }
  VQAF_REQUIRE(epilogue != G2_EPI_GELU_BWD || bias == nullptr, VQAF_E_SHAPE);
  G2Args g = {};
  VQAF_REQUIRE(vqaf_internal_zero_block(&g.zeros) == VQAF_OK, VQAF_E_LAUNCH);
  g.P = reinterpret_cast<const bf16_t*>(w);
  g.Q = reinterpret_cast<const bf16_t*>(x);
  g.ldp = ldw; g.ldq = ldx;
  g.np = N; g.nq = M; g.K = K;
  g.bias = bias;
  g.bias_bf16 = bias_dtype == VQAF_BF16;
  g.Y = reinterpret_cast<bf16_t*>(y);
  g.Y2 = epilogue == G2_EPI_GELU ? reinterpret_cast<bf16_t*>(aux) : nullptr;
  g.H = epilogue == G2_EPI_GELU_BWD ? reinterpret_cast<const bf16_t*>(aux) : nullptr;
  g.ldy = ldy; g.ld2 = ld_aux;
  g.k_per_split = ((K + G2_BK - 1) / G2_BK) * G2_BK;
  g.splits = 1;
  g.tiles_p = (N + G2_T - 1) / G2_T;
  g.tiles_q = (M + G2_T - 1) / G2_T;
  hipStream_t st = (hipStream_t)stream;
  if (w_kmajor) {
    if (epilogue == G2_EPI_BIAS) return g2_launch<true, false, G2_EPI_BIAS>(g, st);
    if (epilogue == G2_EPI_GELU) return g2_launch<true, false, G2_EPI_GELU>(g, st);
    return g2_launch<true, false, G2_EPI_GELU_BWD>(g, st);
  }
  if (epilogue == G2_EPI_BIAS) return g2_launch<false, false, G2_EPI_BIAS>(g, st);
  if (epilogue == G2_EPI_GELU) return g2_launch<false, false, G2_EPI_GELU>(g, st);
  return g2_launch<false, false, G2_EPI_GELU_BWD>(g, st);
}

extern "C" int vqaf_wgrad256_supported(long ldg, long ldx, long ldc, int M, int N, int K) {
  return (M >= 8 && N >= 8 && K > 0 && M % 8 == 0 && N % 8 == 0 && g2_pitch_ok(ldg, M) && g2_pitch_ok(ldx, N) &&
          ldc >= N) ? 1 : 0;
}

extern "C" int vqaf_wgrad256(const void* gy, long ldg, const void* x, long ldx, int M, int N, int K, float* c,
                             long ldc, float* colsum, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(gy && x && c, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_wgrad256_supported(ldg, ldx, ldc, M, N, K) == 1, VQAF_E_SHAPE);
  VQAF_REQUIRE(vqaf_aligned16(gy) && vqaf_aligned16(x), VQAF_E_ALIGN);
  G2Args g = {};
  VQAF_REQUIRE(vqaf_internal_zero_block(&g.zeros) == VQAF_OK, VQAF_E_LAUNCH);
  g.P = reinterpret_cast<const bf16_t*>(gy);
  g.Q = reinterpret_cast<const bf16_t*>(x);
  g.ldp = ldg; g.ldq = ldx;
  g.np = M; g.nq = N; g.K = K;
  g.C = c; g.colsum = colsum; g.ldc = ldc;
  g.tiles_p = (M + G2_T - 1) / G2_T;
  g.tiles_q = (N + G2_T - 1) / G2_T;
  // one workgroup per CU: as many slices of the token range as fill 256 CUs, each at least 4 k-steps long
  const long nt = (long)g.tiles_p * g.tiles_q;
  const int ksteps = (K + G2_BK - 1) / G2_BK;
  long want = nt >= 256 ? 1 : 256 / nt;
  if (want > ksteps / 4) want = ksteps / 4;
  if (want < 1) want = 1;
  g.k_per_split = (int)(((ksteps + want - 1) / want) * G2_BK);
  g.splits = (K + g.k_per_split - 1) / g.k_per_split;
  g.atomic = g.splits > 1;
  VQAF_REQUIRE(nt * g.splits < 0x7fffffffL, VQAF_E_SHAPE);
  hipStream_t st = (hipStream_t)stream;
  if (g.atomic) {
    if (ldc == N) {
      if (hipMemsetAsync(c, 0, (size_t)M * N * 4, st) != hipSuccess) return VQAF_E_LAUNCH;
    } else if (hipMemset2DAsync(c, (size_t)ldc * 4, 0, (size_t)N * 4, (size_t)M, st) != hipSuccess) {
      return VQAF_E_LAUNCH;
    }
    if (colsum != nullptr && hipMemsetAsync(colsum, 0, (size_t)M * 4, st) != hipSuccess) return VQAF_E_LAUNCH;
  }
  return g2_launch<true, true, G2_EPI_F32>(g, st);
}
