// Weight gradient of the encoder Linears on 256 x 256 tiles (model/vision_encoder.py:83-127,
// model/text_encoder.py:14-48: what autograd computes as grad_output.t() @ input for every nn.Linear of the
// stacks): C (M,N) = G^T X, G (K,M) and X (K,N) bf16 with the token index K slow in both.
// A CU fetches operands at a limited rate (about 40-50 GB/s through its LDS-DMA path when most of it comes
// from L2), so what sets the MFMA rate is flops per fetched byte and bytes in flight.  One workgroup of 16
// waves (4 x 4 wave tiles of 64 x 64) owns a CU and a 256 x 256 tile: half the bytes per flop of a 128 x 128
// tile.  Operands go HBM/L2 -> LDS by LDS-DMA in k-steps of 32 through a ring of four 32 KB stages, two to
// three k-steps in flight while one is multiplied, counted waits, one barrier per k-step, and the LDS reads
// of each k16 half-step issued under the MFMAs of the one before.  Both images are
// k-major ([32 k][256 columns], XOR swizzle on the source chunk) and are read with the transposing LDS
// read.  The token range is split so that about one workgroup per CU runs; each writes its f32 partial tile
// to its own slab and vqaf_wgrad_tn's reduction kernel adds the slabs in a fixed order.
#include <mutex>

#include "common.h"

namespace {

constexpr int W2_T = 256, W2_BK = 32, W2_THREADS = 1024, W2_STAGES = 4;
constexpr int W2_OPER = W2_BK * W2_T * 2;       // 16 KB: one operand, one k-step
constexpr int W2_STAGE = 2 * W2_OPER;           // 32 KB
constexpr int W2_LDS = W2_STAGES * W2_STAGE;    // 128 KB

typedef __bf16 w2_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 w2_bf16x2 __attribute__((ext_vector_type(2)));

struct W2Args {
  const bf16_t* G;
  const bf16_t* X;
  float* slabs;              // (splits, M, N) f32
  float* cs_slabs;           // (splits, M) f32 partial column sums of G, or NULL
  const unsigned int* zeros; // 16 bytes of zeros: the source of chunks outside the matrices
  long ldg, ldx;
  int M, N, K, k_per_split, splits, tiles_m, tiles_n;
};

__device__ __forceinline__ void w2_glds16(const void* src, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

__global__ __launch_bounds__(W2_THREADS) void wgrad256_kernel(const W2Args g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char w2_lds[];
  // Workgroup ids go round-robin over the 8 XCDs; each XCD gets one contiguous run of (slice, tile), so the
  // workgroups sharing an L2 read the same token slabs.
  const int nt = g.tiles_m * g.tiles_n, total = nt * g.splits;
  const int xcd = blockIdx.x & 7, q8 = total >> 3, r8 = total & 7;
  const int lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
  const int split = lin / nt, tile = lin - split * nt;
  const int tile_m = tile / g.tiles_n, tile_n = tile - tile_m * g.tiles_n;
  const int m0 = tile_m * W2_T, n0 = tile_n * W2_T;
  const int k_lo = split * g.k_per_split, kspan = min(g.K, k_lo + g.k_per_split) - k_lo;
  const int T = (kspan + W2_BK - 1) / W2_BK;   // >= 1: the host sizes the slices so that none is empty
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  // ---- LDS-DMA: one 16-byte chunk of each operand per thread and k-step.  Chunk tid -> (k row tid / 32,
  //      position tid % 32) holding source chunk position ^ ((k & 3) << 2): the four k rows one transposing
  //      read touches then lie on disjoint banks.
  const int kk = tid >> 5, cc = (tid & 31) ^ ((kk & 3) << 2);
  const bool g_ok = m0 + cc * 8 < g.M, x_ok = n0 + cc * 8 < g.N;
  const bf16_t* gs = g.G + (long)(k_lo + kk) * g.ldg + (g_ok ? m0 + cc * 8 : 0);
  const bf16_t* xs = g.X + (long)(k_lo + kk) * g.ldx + (x_ok ? n0 + cc * 8 : 0);
  const long g_step = (long)W2_BK * g.ldg, x_step = (long)W2_BK * g.ldx;
  const void* const zeros = g.zeros;
  auto issue = [&](int t) {
    unsigned char* st = w2_lds + (t & (W2_STAGES - 1)) * W2_STAGE + wave * 1024;
    const bool k_in = t * W2_BK + kk < kspan;
    w2_glds16(g_ok && k_in ? (const void*)(gs + (long)t * g_step) : zeros, st);
    w2_glds16(x_ok && k_in ? (const void*)(xs + (long)t * x_step) : zeros, st + W2_OPER);
  };

  // ---- fragment addresses within a stage.  The reads are inline assembly on purpose: for an LDS read it
  //      can see, hipcc waits until every outstanding LDS-DMA has landed, which would drain the ring at
  //      each k-step.  The counted vmcnt + barrier at the top of each k-step orders these reads.
  //      Transposing read of a 32-column block at byte column cb: k row 8 (lane >> 5) + ((lane & 15) >> 2)
  //      (+4 for the second read), bytes cb + 2 (16 ((lane >> 4) & 1) + 4 (lane & 3)), swizzled;
  //      k16 step ks adds 16 rows.
  const unsigned lds_base = (unsigned)reinterpret_cast<size_t>((__attribute__((address_space(3))) unsigned char*)w2_lds);
  const int li = lane & 15, g16 = lane >> 4;
  const int krow = 8 * (g16 >> 1) + (li >> 2);
  unsigned ga[2], xa[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int cg = (wm * 64 + i * 32 + 16 * (g16 & 1) + 4 * (li & 3)) * 2;
    const int cx = (wn * 64 + i * 32 + 16 * (g16 & 1) + 4 * (li & 3)) * 2;
    ga[i] = (unsigned)(krow * 512 + (cg ^ ((krow & 3) << 6)));
    xa[i] = (unsigned)(W2_OPER + krow * 512 + (cx ^ ((krow & 3) << 6)));
  }
  auto read_tr = [&](unsigned addr) -> u32x2 {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
  };
  struct Frags {
    u32x4 p[2], q[2];
  };
  auto load_frags = [&](Frags& f, unsigned img, int ks) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned ad = img + ga[i] + (unsigned)(ks * 16 * 512);
      const u32x2 lo = read_tr(ad), hi = read_tr(ad + 4 * 512);
      f.p[i] = u32x4{lo.x, lo.y, hi.x, hi.y};
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned ad = img + xa[i] + (unsigned)(ks * 16 * 512);
      const u32x2 lo = read_tr(ad), hi = read_tr(ad + 4 * 512);
      f.q[i] = u32x4{lo.x, lo.y, hi.x, hi.y};
    }
  };
  auto as_frag = [](const u32x4& v) -> w2_bf16x8 {
    union { u32x4 u; w2_bf16x8 b; } c;
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
  // column sums of G over the tokens (bias gradient), on the vector ALU from the fragments the MFMAs use
  // anyway: a lane holds G(8 tokens, channel lane & 31), one two-element dot product with ones per pair.
  // The four waves that do it (one per 64-channel block) sit on four different SIMDs.
  float csum[2] = {0.f, 0.f};
  const bool sums = g.cs_slabs != nullptr && tile_n == 0 && wn == wm;   // wave-uniform
  w2_bf16x2 one2;
  one2[0] = (__bf16)1.0f;
  one2[1] = (__bf16)1.0f;
  auto mfma_step = [&](const Frags& f) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(f.p[i]), as_frag(f.q[j]), acc[i][j], 0, 0, 0);
    if (sums) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const unsigned w[4] = {f.p[i].x, f.p[i].y, f.p[i].z, f.p[i].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          union { unsigned u; w2_bf16x2 b; } c;
          c.u = w[e];
          csum[i] = __builtin_amdgcn_fdot2_f32_bf16(c.b, one2, csum[i], false);
        }
      }
    }
  };

  // Software pipeline over k16 half-steps with two fragment sets: the LDS reads of the next half-step are
  // always in flight while the MFMAs of the current one run.  Per k-step t: read (t, second half) | multiply
  // (t, first half) | wait until this wave's chunks of k-step t+1 have landed, barrier (now everyone's have,
  // and everyone is done reading k-step t-1) | refill k-step t-1's ring slot with k-step t+3 | read (t+1,
  // first half) | multiply (t, second half).  Two LDS-DMA instructions per wave and k-step: at the wait,
  // k-steps t+1 and t+2 are outstanding, so "t+1 has landed" is vmcnt(2) (vmcnt(0) at the tail).
#pragma unroll
  for (int s = 0; s < W2_STAGES - 1; ++s)
    if (s < T) issue(s);
  if (T >= 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (T == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  Frags f0, f1;
  load_frags(f0, lds_base, 0);
  for (int t = 0; t < T; ++t) {
    const unsigned img = lds_base + (unsigned)((t & (W2_STAGES - 1)) * W2_STAGE);
    load_frags(f1, img, 1);
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");   // the 8 reads of f0 were issued before f1's
    __builtin_amdgcn_sched_barrier(0);
    mfma_step(f0);
    __builtin_amdgcn_sched_barrier(0);
    if (t + 1 < T) {
      if (t + 2 < T) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (t + W2_STAGES - 1 < T) issue(t + W2_STAGES - 1);
    if (t + 1 < T) {
      load_frags(f0, lds_base + (unsigned)(((t + 1) & (W2_STAGES - 1)) * W2_STAGE), 0);
      asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma_step(f1);
    __builtin_amdgcn_sched_barrier(0);
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

}  // namespace

// how the token range is cut for the 256 x 256 tiles: about one workgroup per CU, each at least 8 k-steps
void vqaf_internal_wgrad256_plan(int M, int N, int K, int* k_per_split, int* splits) {
  const long nt = (long)((M + W2_T - 1) / W2_T) * ((N + W2_T - 1) / W2_T);
  const int ksteps = (K + W2_BK - 1) / W2_BK;
  long want = nt >= 256 ? 1 : 256 / nt;
  if (want > ksteps / 8) want = ksteps / 8;
  if (want < 1) want = 1;
  *k_per_split = (int)(((ksteps + want - 1) / want) * W2_BK);
  *splits = (K + *k_per_split - 1) / *k_per_split;
}

// launches the partial products into `slabs` (splits x (M*N) f32) and `cs_slabs` (splits x M, or NULL)
int vqaf_internal_wgrad256_launch(const void* gy, long ldg, const void* x, long ldx, int M, int N, int K, float* slabs,
                                  float* cs_slabs, hipStream_t st) {
  W2Args g;
  if (vqaf_internal_zero_block(&g.zeros) != VQAF_OK) return VQAF_E_LAUNCH;
  vqaf_internal_wgrad256_plan(M, N, K, &g.k_per_split, &g.splits);
  g.G = reinterpret_cast<const bf16_t*>(gy);
  g.X = reinterpret_cast<const bf16_t*>(x);
  g.slabs = slabs;
  g.cs_slabs = cs_slabs;
  g.ldg = ldg; g.ldx = ldx;
  g.M = M; g.N = N; g.K = K;
  g.tiles_m = (M + W2_T - 1) / W2_T;
  g.tiles_n = (N + W2_T - 1) / W2_T;
  const long blocks = (long)g.tiles_m * g.tiles_n * g.splits;
  if (blocks >= 0x7fffffffL) return VQAF_E_SHAPE;
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad256_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              W2_LDS);
  });
  hipLaunchKernelGGL(wgrad256_kernel, dim3((unsigned)blocks), dim3(W2_THREADS), W2_LDS, st, g);
  return hipGetLastError() == hipSuccess ? VQAF_OK : VQAF_E_LAUNCH;
}
