// bf16 MFMA GEMM for the classifier under DP-SGD: C (M,N) = op(A) op(B) with f32 accumulation,
//   forward   y  = x W^T + b        A = x (M,K) k-contiguous,  B = W (N,K) k-contiguous
//   dgrad     gx = gy W             A = gy (M,K) k-contiguous, B = W (K,N) k-major
//   wgrad     S  = G^T A            A = G (K,M) k-major,       B = A (K,N) k-major   (clipped sums:
//             the contraction index is the batch; rows of G are already scaled by the clip factor)
// These are small GEMMs (about 1024 x 2048 x 1024): a few hundred output tiles and 16-32 k-steps,
// so what bounds them is the latency of each k-step's operand fetch, not MFMA issue.  Structure:
//   * tiles of 128 x 64 or 64 x 64 outputs so that every shape gives >= 256 workgroups;
//   * operands go HBM/L2 -> LDS by LDS-DMA (global_load_lds, 16 B per lane, no registers), four
//     k-steps of 64 deep in a ring: three k-steps are in flight while one is multiplied, waits are
//     counted (s_waitcnt vmcnt(N)) and there is ONE raw s_barrier per k-step;
//   * an LDS-DMA writes wave-linear bytes, so the LDS images are unpadded and bank conflicts are
//     removed by an XOR swizzle applied to the SOURCE address when the tile is fetched and to the
//     LDS address when fragments are read (k-major images: ds_read_b64_tr_b16, the transposing
//     read; k-contiguous images: ds_read_b128);
//   * rows / k-ranges outside the matrix are fetched from a 16-byte block of zeros, so the loop
//     has no edge cases; out-of-range outputs are masked at the store.
#include <atomic>
#include <mutex>
#include <type_traits>

#include "common.h"

// 16 bytes of zeros in device memory: the source of every out-of-range operand chunk.  As with any
// file-scope device array, host code must not pass the NAME to a kernel; it asks for the device
// address once per device and uses that.
static __device__ __align__(16) unsigned int gb_zero_block[4] = {0u, 0u, 0u, 0u};

namespace {

constexpr int GB_THREADS = 256;
constexpr int GB_BK = 64;
constexpr int GB_BN = 64;
constexpr int GB_STAGES = 4;

typedef short gb_s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 gb_bf16x8 __attribute__((ext_vector_type(8)));

enum { GB_OUT_F32 = 0, GB_OUT_F32_ACC = 1, GB_OUT_F32_ATOMIC = 2, GB_OUT_BF16 = 3 };

struct GemmArgs {
  const bf16_t* A;
  const bf16_t* B;
  const void* bias;       // (N) or NULL
  void* C;
  float* colsum;          // (M) sum over k of A(m,k), or NULL
  const unsigned int* zeros;
  long lda, ldb, ldc;
  int M, N, K, k_per_split, out_mode, bias_bf16;
  int tiles_m, tiles_n, reg_m, reg_n;  // output tiles; XCD regions (reg_m * reg_n == 8) or 0
};

// GroupNorm (+ReLU +inverted dropout) applied to the output tile of a Linear before it leaves the
// workgroup (model/model.py:55-58: Linear -> GroupNorm -> ReLU -> Dropout).  64-column tiles hold whole
// groups of 8..64 channels, so the statistics of a (sample, group) never leave the tile.
struct GnEpi {
  const void* gamma;             // (N) bf16 or f32
  const void* beta;
  void* y;                       // (M,N) bf16: the activation handed to the next layer
  void* xhat;                    // (M,N) bf16: normalised input, kept for the backward and the ghost norm
  float* rstd;                   // (M, groups)
  const uint64_t* drop_counter;  // {unused, counter} or NULL
  unsigned long long drop_seed;
  float eps, drop_p;
  int groups, relu, param_bf16;
  // backward form (the GEMM is the input gradient of the NEXT Linear; its output is dL/dy of this norm)
  const void* y_in;     // (M,N) bf16: the forward output (ReLU gate and dropout keep mask), or NULL without ReLU
  const void* xhat_in;  // (M,N) bf16
  const float* rstd_in; // (M, groups)
  void* g_pre;          // (M,N) bf16: gradient at the affine output (the norm layer's backprop record)
  void* grad_x;         // (M,N) bf16: gradient at the norm's input (= the output of the Linear in front of it)
  float out_scale;      // 1 / (1 - p) of the inverted dropout, 1 without it
};

// The DP step's parameter update applied to a clipped-sum tile before it leaves the workgroup: the tile IS
// the clipped gradient sum of 64 x 64 weights, so noise, scaling and AdamW run on it in registers and the
// sum itself is never written.  Flat buckets as DPOptimizer keeps them (one offset per parameter).
constexpr int GB_MAX_GROUP = 4;
struct AdamEpi {
  float* P; float* M; float* V; float* G;  // parameters, moments, final gradient (G may be NULL)
  bf16_t* shadow;                          // bf16 copy of the parameters, or NULL
  const uint64_t* rng;                     // {seed, counter} of the noise stream
  const uint64_t* step;                    // {unused, updates applied}
  float noise_std, scale, lr, beta1, beta2, eps, weight_decay;
  long noise_lo;                           // flat index of element 0 of the noise stream
  long w_off[GB_MAX_GROUP], b_off[GB_MAX_GROUP];  // flat offsets of each product's weight / bias (b_off < 0: none)
};

__device__ __forceinline__ void glds16(const void* src, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else static_assert(N == 0, "add the literal");
}

template <int N>
__device__ __forceinline__ void wait_lgkmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
  else if constexpr (N == 5) asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
  else static_assert(N == 0, "add the literal");
}

// Swizzle of a k-major image [64 k][W columns] (row pitch 2W bytes, a multiple of 128): the 16-byte
// chunk c of k-row k is stored at chunk position c ^ kswz(k).  A transposing read touches 4
// consecutive k-rows x 64 bytes per 32 lanes; the XOR moves those four segments onto disjoint banks.
template <int W>
__device__ __forceinline__ int kswz(int k) {
  if constexpr (W >= 128) return (k & 3) << 2;   // 64-byte steps inside a >= 256-byte row
  else return ((k >> 1) & 1) << 2;               // 128-byte rows: rows k and k+2 share banks
}
// Swizzle of a k-contiguous image [rows][64 k] (128-byte rows): chunk c (8 k values) of row r is
// stored at chunk position c ^ ((r >> 1) & 7): the 16 rows a ds_read_b128 lane group touches then
// fall on 16 distinct 16-byte slots of the 256-byte bank row.
__device__ __forceinline__ int rswz(int r) { return (r >> 1) & 7; }

template <int MT, bool A_KMAJOR, bool B_KMAJOR, bool COLSUM, int STAGES = GB_STAGES, int EPI = 0>
__device__ __forceinline__ void gemm_bf16_body(const GemmArgs& g, const int tile_m, const int tile_n,
                                               const GnEpi* epi = nullptr, const AdamEpi* adam = nullptr,
                                               const long adam_w_off = 0, const long adam_b_off = -1) {
  constexpr int BM = 64 * MT;
  constexpr int A_BYTES = BM * GB_BK * 2;
  constexpr int B_BYTES = GB_BN * GB_BK * 2;
  constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int NLA = A_BYTES / (GB_THREADS * 16);  // LDS-DMA instructions per thread and k-step, A
  constexpr int NLB = B_BYTES / (GB_THREADS * 16);
  constexpr int LPS = NLA + NLB;
  extern __shared__ __attribute__((aligned(16))) unsigned char gb_lds[];

  // Waves 0-3 multiply, waves 4-7 only fetch: an LDS-DMA instruction holds its wave's issue slot for
  // a long time, so fetch and MFMA streams live in different waves and overlap on the SIMDs.
  const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: LDS-DMA bases stay in SGPRs
  const bool is_loader = wave8 >= 4;
  const int wave = wave8 & 3;
  const int tid = threadIdx.x & (GB_THREADS - 1), lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = tile_m * BM, n0 = tile_n * GB_BN;
  const int k_lo = blockIdx.z * g.k_per_split;
  const int k_hi = min(g.K, k_lo + g.k_per_split);
  const int T = (k_hi - k_lo + GB_BK - 1) / GB_BK;
  if (T <= 0) return;
  const unsigned char* zeros = reinterpret_cast<const unsigned char*>(g.zeros);

  if (is_loader) {
  // ---- per-thread source descriptors of its chunks (fixed over the k loop) ----
  // k-major operand: chunk q -> (k row q / CPR, swizzled chunk position q % CPR)
  // k-contiguous   : chunk q -> (row q / 8, swizzled chunk position q % 8)
  long a_off[NLA], b_off[NLB];   // element offset of the chunk at k-step 0 (k_lo included)
  int a_k[NLA], b_k[NLB];        // k (within the k-step) of the chunk's first element
  bool a_ok[NLA], b_ok[NLB];     // row / column inside the matrix
#pragma unroll
  for (int j = 0; j < NLA; ++j) {
    const int q = j * GB_THREADS + tid;
    if constexpr (A_KMAJOR) {
      constexpr int CPR = BM / 8;
      const int k = q / CPR, c = (q % CPR) ^ kswz<BM>(k);
      const int m = m0 + c * 8;
      a_k[j] = k;
      a_ok[j] = m < g.M;   // chunks never straddle the row end: lda covers M rounded up to 8
      a_off[j] = (long)(k_lo + k) * g.lda + m;
    } else {
      const int r = q / 8, c = (q % 8) ^ rswz(r);
      a_k[j] = c * 8;
      a_ok[j] = true;      // rows past M repeat the last row; their outputs are masked
      a_off[j] = (long)min(m0 + r, g.M - 1) * g.lda + k_lo + c * 8;
    }
  }
#pragma unroll
  for (int j = 0; j < NLB; ++j) {
    const int q = j * GB_THREADS + tid;
    if constexpr (B_KMAJOR) {
      constexpr int CPR = GB_BN / 8;
      const int k = q / CPR, c = (q % CPR) ^ kswz<GB_BN>(k);
      const int n = n0 + c * 8;
      b_k[j] = k;
      b_ok[j] = n < g.N;
      b_off[j] = (long)(k_lo + k) * g.ldb + n;
    } else {
      const int r = q / 8, c = (q % 8) ^ rswz(r);
      b_k[j] = c * 8;
      b_ok[j] = true;
      b_off[j] = (long)min(n0 + r, g.N - 1) * g.ldb + k_lo + c * 8;
    }
  }
  const long a_step = A_KMAJOR ? (long)GB_BK * g.lda : (long)GB_BK;
  const long b_step = B_KMAJOR ? (long)GB_BK * g.ldb : (long)GB_BK;
  const int kspan = k_hi - k_lo;

  // branch-free choice between the chunk's address and the block of zeros
  auto pick = [&](bool ok, const bf16_t* p) -> const void* {
    const unsigned long long m = ok ? ~0ull : 0ull;
    return reinterpret_cast<const void*>((reinterpret_cast<unsigned long long>(p) & m) |
                                         (reinterpret_cast<unsigned long long>(zeros) & ~m));
  };
  auto issue = [&](int t) {
    unsigned char* st = gb_lds + (t % STAGES) * STAGE_BYTES + wave * 1024;
    const int kbase = t * GB_BK;
#pragma unroll
    for (int j = 0; j < NLA; ++j)
      glds16(pick(a_ok[j] && (kbase + a_k[j] < kspan), g.A + a_off[j] + (long)t * a_step),
             st + j * GB_THREADS * 16);
#pragma unroll
    for (int j = 0; j < NLB; ++j)
      glds16(pick(b_ok[j] && (kbase + b_k[j] < kspan), g.B + b_off[j] + (long)t * b_step),
             st + A_BYTES + j * GB_THREADS * 16);
  };

    // three k-steps in flight; k-step t has landed when at most min(2, T-1-t) younger k-steps of
    // this wave are outstanding.  After the barrier every consumer is done with k-step t-1, whose
    // ring slot the next fetch overwrites.
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
      if (s < T) issue(s);
    for (int t = 0; t < T; ++t) {
      const int younger = min(STAGES - 2, T - 1 - t);
      if (younger >= 2) wait_vmcnt<2 * LPS>();
      else if (younger == 1) wait_vmcnt<LPS>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      if (t + STAGES - 1 < T) issue(t + STAGES - 1);
    }
    return;
  }

  f32x16 acc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f32x16 csum[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) csum[i][r] = 0.f;
  gb_bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
  // backward GroupNorm epilogue: its operands (two 16-byte pieces per thread and row pass) are requested
  // now and used after the k loop
  u32x4 epi_y[2], epi_xh[2];
  if constexpr (EPI == 2) {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int mb = min(m0 + (tid >> 3) + 32 * pass, g.M - 1);
      const size_t off = (size_t)mb * g.N + n0 + 8 * (tid & 7);
      epi_xh[pass] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(epi->xhat_in) + off);
      epi_y[pass] = epi->y_in != nullptr ? *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(epi->y_in) + off)
                                         : u32x4{0u, 0u, 0u, 0u};
    }
  }

  // Fragment reads are inline assembly on purpose: for an LDS read it can see, hipcc waits until
  // EVERY outstanding LDS-DMA has landed (vmcnt(0)), which would drain the ring at each k-step.
  // The hand-placed vmcnt / barrier above each k-step is what orders these reads.
  const unsigned lds_base = (unsigned)reinterpret_cast<size_t>(
      (__attribute__((address_space(3))) unsigned char*)gb_lds);
  // k-major image of width W: byte address (within the image) of this lane's first transposing
  // read at k16 step 0; the second read is 4 k-rows further, k16 step ks adds 16 rows.  The swizzle
  // term depends on (k & 3) or ((k >> 1) & 1) only, so it is the same for every step.
  auto kmajor_addr = [&](auto wtag, int colbase) -> unsigned {
    constexpr int W = decltype(wtag)::value;
    const int i = lane & 15, g16 = lane >> 4;
    const int krow = 8 * (g16 >> 1) + (i >> 2);
    const int c0 = (colbase + 16 * (g16 & 1) + 4 * (i & 3)) * 2;  // byte in row; the XOR flips bits >= 6
    return (unsigned)(krow * (2 * W) + (c0 ^ (kswz<W>(krow) << 4)));
  };
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
  typedef std::integral_constant<int, BM> WA;
  typedef std::integral_constant<int, GB_BN> WB;
  unsigned a_addr[MT], b_addr;
  int a_row[MT], b_row;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    a_row[i] = wm * (32 * MT) + i * 32 + (lane & 31);
    if constexpr (A_KMAJOR) a_addr[i] = kmajor_addr(WA{}, wm * (32 * MT) + i * 32);
    else a_addr[i] = (unsigned)(a_row[i] * 128);
  }
  b_row = wn * 32 + (lane & 31);
  if constexpr (B_KMAJOR) b_addr = A_BYTES + kmajor_addr(WB{}, wn * 32);
  else b_addr = (unsigned)(A_BYTES + b_row * 128);

  struct Frags {
    u32x4 a[MT];
    u32x4 b;
  };
  // issues the LDS reads of k16 step ks of the image at LDS byte address `img`
  auto load_frags = [&](Frags& f, unsigned img, int ks) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      if constexpr (A_KMAJOR) {
        const unsigned ad = img + a_addr[i] + (unsigned)(ks * 16 * 2 * BM);
        const u32x2 lo = read_tr(ad), hi = read_tr(ad + 4 * 2 * BM);
        f.a[i] = u32x4{lo.x, lo.y, hi.x, hi.y};
      } else {
        f.a[i] = read_b128(img + a_addr[i] + (unsigned)((((2 * ks + (lane >> 5)) ^ rswz(a_row[i]))) << 4));
      }
    }
    if constexpr (B_KMAJOR) {
      const unsigned ad = img + b_addr + (unsigned)(ks * 16 * 2 * GB_BN);
      const u32x2 lo = read_tr(ad), hi = read_tr(ad + 4 * 2 * GB_BN);
      f.b = u32x4{lo.x, lo.y, hi.x, hi.y};
    } else {
      f.b = read_b128(img + b_addr + (unsigned)((((2 * ks + (lane >> 5)) ^ rswz(b_row))) << 4));
    }
  };
  constexpr int READS_PER_KS = (A_KMAJOR ? 2 * MT : MT) + (B_KMAJOR ? 2 : 1);
  auto as_bf16x8 = [](const u32x4& v) -> gb_bf16x8 {
    union { u32x4 u; gb_bf16x8 b; } c;
    c.u = v;
    return c.b;
  };
  auto mfma_step = [&](const Frags& f) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(f.a[i]), as_bf16x8(f.b), acc[i], 0, 0, 0);
      if constexpr (COLSUM)
        csum[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(f.a[i]), ones, csum[i], 0, 0, 0);
    }
  };

  for (int t = 0; t < T; ++t) {
    // the loaders waited for their parts of k-step t before this barrier
    __builtin_amdgcn_s_barrier();
    const unsigned img = lds_base + (unsigned)((t % STAGES) * STAGE_BYTES);
    // the reads of k16 step ks+1 are in flight while step ks multiplies
    Frags f0, f1;
    load_frags(f0, img, 0);
    load_frags(f1, img, 1);
    wait_lgkmcnt<READS_PER_KS>();
    __builtin_amdgcn_sched_barrier(0);
    mfma_step(f0);
    __builtin_amdgcn_sched_barrier(0);
    load_frags(f0, img, 2);
    wait_lgkmcnt<READS_PER_KS>();
    __builtin_amdgcn_sched_barrier(0);
    mfma_step(f1);
    __builtin_amdgcn_sched_barrier(0);
    load_frags(f1, img, 3);
    wait_lgkmcnt<READS_PER_KS>();
    __builtin_amdgcn_sched_barrier(0);
    mfma_step(f0);
    wait_lgkmcnt<0>();
    __builtin_amdgcn_sched_barrier(0);
    mfma_step(f1);
  }

  // ---- epilogue.  C/D layout of a 32x32 tile: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  const int n = n0 + wn * 32 + (lane & 31);
  float bias_v = 0.f;
  if (g.bias != nullptr && n < g.N)
    bias_v = g.bias_bf16 ? bf16_to_f32(reinterpret_cast<const bf16_t*>(g.bias)[n])
                         : reinterpret_cast<const float*>(g.bias)[n];
  if constexpr (EPI == 3) {
    // AdamW epilogue of a clipped-sum tile (see AdamEpi).  The tile goes through LDS so that a thread owns
    // 8 consecutive weights of a row: two generator calls, 16-byte moves of parameters and moments.
    static_assert(MT == 1 && A_KMAJOR && B_KMAJOR, "AdamW epilogue: clipped-sum tiles of 64 x 64");
    constexpr int TP = 68;
    float* T = reinterpret_cast<float*>(gb_lds);
    float* Tb = T + 64 * TP;  // the tile's 64 row sums (bias gradients)
    const AdamEpi& e = *adam;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      T[row * TP + wn * 32 + (lane & 31)] = acc[0][r];
      if (COLSUM && wn == 0 && (lane & 31) == 0) Tb[row] = csum[0][r];
    }
    __syncthreads();
    const bool draw = e.noise_std != 0.f;
    const unsigned long long seed = draw ? e.rng[0] : 0ull, counter = draw ? e.rng[1] : 0ull;
    const VqafAdamConsts ac = vqaf_adam_consts(e.step[1], e.noise_std, e.scale, e.lr, e.beta1, e.beta2, e.eps, e.weight_decay);
    const int chunk = tid & 7, nn = n0 + 8 * chunk;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int row = (tid >> 3) + 32 * pass, m = m0 + row;
      if (m < g.M && nn < g.N) {
        const long flat = adam_w_off + (long)m * g.N + nn;
        float sv[8], z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(T + row * TP + 8 * chunk);
        const f32x4 t1 = *reinterpret_cast<const f32x4*>(T + row * TP + 8 * chunk + 4);
        sv[0] = t0[0]; sv[1] = t0[1]; sv[2] = t0[2]; sv[3] = t0[3];
        sv[4] = t1[0]; sv[5] = t1[1]; sv[6] = t1[2]; sv[7] = t1[3];
        if (draw) {
          float za[4], zb[4];
          const unsigned long long call = (unsigned long long)((flat - e.noise_lo) >> 2);
          vqaf_normal4(seed, counter, call, za);
          vqaf_normal4(seed, counter, call + 1ull, zb);
#pragma unroll
          for (int j = 0; j < 4; ++j) { z[j] = za[j]; z[4 + j] = zb[j]; }
        }
        float pv[8], mv[8], vv[8], gv[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          // (requesting these before the k loop was measured: 48 more registers, two workgroups per CU
          // instead of three, 68 instead of 56 us)
          const f32x4 p4 = *reinterpret_cast<const f32x4*>(e.P + flat + 4 * h);
          const f32x4 m4 = *reinterpret_cast<const f32x4*>(e.M + flat + 4 * h);
          const f32x4 v4 = *reinterpret_cast<const f32x4*>(e.V + flat + 4 * h);
#pragma unroll
          for (int j = 0; j < 4; ++j) { pv[4 * h + j] = p4[j]; mv[4 * h + j] = m4[j]; vv[4 * h + j] = v4[j]; }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) vqaf_adamw1(ac, sv[j], z[j], pv[j], mv[j], vv[j], gv[j]);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          // next touched one whole step later: streamed past the caches, as the stand-alone update does
          store_nt16(e.P + flat + 4 * h, u32x4{__float_as_uint(pv[4 * h]), __float_as_uint(pv[4 * h + 1]), __float_as_uint(pv[4 * h + 2]), __float_as_uint(pv[4 * h + 3])});
          store_nt16(e.M + flat + 4 * h, u32x4{__float_as_uint(mv[4 * h]), __float_as_uint(mv[4 * h + 1]), __float_as_uint(mv[4 * h + 2]), __float_as_uint(mv[4 * h + 3])});
          store_nt16(e.V + flat + 4 * h, u32x4{__float_as_uint(vv[4 * h]), __float_as_uint(vv[4 * h + 1]), __float_as_uint(vv[4 * h + 2]), __float_as_uint(vv[4 * h + 3])});
          if (e.G != nullptr)
            store_nt16(e.G + flat + 4 * h, u32x4{__float_as_uint(gv[4 * h]), __float_as_uint(gv[4 * h + 1]), __float_as_uint(gv[4 * h + 2]), __float_as_uint(gv[4 * h + 3])});
        }
        if (e.shadow != nullptr) {  // (bucket offsets are multiples of 4 elements: 8-byte pieces of the bf16 copy)
          *reinterpret_cast<u32x2*>(e.shadow + flat) = u32x2{pack_bf16x2(pv[0], pv[1]), pack_bf16x2(pv[2], pv[3])};
          *reinterpret_cast<u32x2*>(e.shadow + flat + 4) = u32x2{pack_bf16x2(pv[4], pv[5]), pack_bf16x2(pv[6], pv[7])};
        }
      }
    }
    if constexpr (COLSUM) {
      const int m = m0 + tid;
      if (tid < 64 && m < g.M && adam_b_off >= 0) {  // the bias of output row m: its gradient sum is the row sum of G
        const long flat = adam_b_off + m, idx = flat - e.noise_lo;
        float z = 0.f;
        if (draw) {
          float z4[4];
          vqaf_normal4(seed, counter, (unsigned long long)(idx >> 2), z4);
          const int j = (int)(idx & 3);
          z = j == 0 ? z4[0] : j == 1 ? z4[1] : j == 2 ? z4[2] : z4[3];
        }
        float p = e.P[flat], mo = e.M[flat], v = e.V[flat], gg;
        vqaf_adamw1(ac, Tb[tid], z, p, mo, v, gg);
        e.P[flat] = p;
        e.M[flat] = mo;
        e.V[flat] = v;
        if (e.G != nullptr) e.G[flat] = gg;
        if (e.shadow != nullptr) e.shadow[flat] = f32_to_bf16(p);
      }
    }
    return;
  }
  if constexpr (EPI == 1) {
    // GroupNorm epilogue (64 x 64 tile, N a multiple of 64, 8 <= group size <= 64 dividing 64).  The tile
    // goes through LDS (the operand ring is free now; the fetching waves have left) so that a thread owns 8
    // consecutive channels of a row, as a lane of the stand-alone kernel does: the Linear output is rounded
    // to bf16 first and every sum is taken in that kernel's order, so both paths give the same bits.
    static_assert(MT == 1 && !A_KMAJOR && !B_KMAJOR && !COLSUM, "GroupNorm epilogue: forward tiles of 64 x 64");
    constexpr int TP = 68;  // floats per staged row (16-byte aligned rows)
    float* T = reinterpret_cast<float*>(gb_lds);
    const GnEpi& e = *epi;
    __syncthreads();  // every wave is done with the last k-step's fragments
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      T[row * TP + wn * 32 + (lane & 31)] = bf16_to_f32(f32_to_bf16(acc[0][r] + bias_v));
    }
    __syncthreads();
    const int C = g.N, gs = C / e.groups, lpg = gs / 8;
    const int chunk = tid & 7, c0 = n0 + 8 * chunk;
    float ga[8], be[8];
    if (e.param_bf16) { load8<true>(e.gamma, (size_t)c0, ga); load8<true>(e.beta, (size_t)c0, be); }
    else { load8<false>(e.gamma, (size_t)c0, ga); load8<false>(e.beta, (size_t)c0, be); }
    const bool drop = e.drop_counter != nullptr && e.drop_p > 0.f;
    const unsigned long long dcounter = drop ? e.drop_counter[1] : 0ull;
    const unsigned int drop_thr = (unsigned int)(e.drop_p * 65536.0f);
    const float keep_scale = drop ? 1.0f / (1.0f - e.drop_p) : 1.0f;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int row = (tid >> 3) + 32 * pass, m = m0 + row;
      const int mb = min(m, g.M - 1);  // rows past M repeat the last row (never stored)
      float v[8];
      const f32x4 t0 = *reinterpret_cast<const f32x4*>(T + row * TP + 8 * chunk);
      const f32x4 t1 = *reinterpret_cast<const f32x4*>(T + row * TP + 8 * chunk + 4);
      v[0] = t0[0]; v[1] = t0[1]; v[2] = t0[2]; v[3] = t0[3];
      v[4] = t1[0]; v[5] = t1[1]; v[6] = t1[2]; v[7] = t1[3];
      const float rs = vqaf_gn_fwd8(v, lpg, gs, e.eps);  // v is normalised on return
      float yo[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        yo[i] = fmaf(v[i], ga[i], be[i]);
        if (e.relu) yo[i] = fmaxf(yo[i], 0.f);
      }
      if (drop) {
        const Philox4 pr = philox4x32_10(e.drop_seed, ((size_t)mb * C + c0) >> 3, dcounter);
        const unsigned int w[4] = {pr.x, pr.y, pr.z, pr.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          yo[2 * i] = (w[i] & 0xffffu) >= drop_thr ? yo[2 * i] * keep_scale : 0.f;
          yo[2 * i + 1] = (w[i] >> 16) >= drop_thr ? yo[2 * i + 1] * keep_scale : 0.f;
        }
      }
      if (m < g.M) {
        if ((chunk % lpg) == 0) e.rstd[(size_t)m * e.groups + c0 / gs] = rs;
        store8<true>(e.y, (size_t)m * C + c0, yo);
        store8<true>(e.xhat, (size_t)m * C + c0, v);
      }
    }
    return;
  }
  if constexpr (EPI == 2) {
    // GroupNorm backward on the tile of dL/dy this GEMM just produced (same staging and thread layout as the
    // forward epilogue; the tile is rounded to bf16 first and the sums follow vqaf_groupnorm_bwd's order).
    static_assert(MT == 1 && !A_KMAJOR && B_KMAJOR && !COLSUM, "GroupNorm backward epilogue: dgrad tiles of 64 x 64");
    constexpr int TP = 68;
    float* T = reinterpret_cast<float*>(gb_lds);
    const GnEpi& e = *epi;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      T[row * TP + wn * 32 + (lane & 31)] = bf16_to_f32(f32_to_bf16(acc[0][r]));
    }
    __syncthreads();
    const int C = g.N, gs = C / e.groups, lpg = gs / 8;
    const int chunk = tid & 7, c0 = n0 + 8 * chunk;
    float ga[8];
    if (e.param_bf16) load8<true>(e.gamma, (size_t)c0, ga);
    else load8<false>(e.gamma, (size_t)c0, ga);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int row = (tid >> 3) + 32 * pass, m = m0 + row;
      const int mb = min(m, g.M - 1);
      float gy[8], xh[8], yo[8];
      const f32x4 t0 = *reinterpret_cast<const f32x4*>(T + row * TP + 8 * chunk);
      const f32x4 t1 = *reinterpret_cast<const f32x4*>(T + row * TP + 8 * chunk + 4);
      gy[0] = t0[0]; gy[1] = t0[1]; gy[2] = t0[2]; gy[3] = t0[3];
      gy[4] = t1[0]; gy[5] = t1[1]; gy[6] = t1[2]; gy[7] = t1[3];
      xh[0] = bf16_lo(epi_xh[pass].x); xh[1] = bf16_hi(epi_xh[pass].x); xh[2] = bf16_lo(epi_xh[pass].y); xh[3] = bf16_hi(epi_xh[pass].y);
      xh[4] = bf16_lo(epi_xh[pass].z); xh[5] = bf16_hi(epi_xh[pass].z); xh[6] = bf16_lo(epi_xh[pass].w); xh[7] = bf16_hi(epi_xh[pass].w);
      if (e.relu) {
        yo[0] = bf16_lo(epi_y[pass].x); yo[1] = bf16_hi(epi_y[pass].x); yo[2] = bf16_lo(epi_y[pass].y); yo[3] = bf16_hi(epi_y[pass].y);
        yo[4] = bf16_lo(epi_y[pass].z); yo[5] = bf16_hi(epi_y[pass].z); yo[6] = bf16_lo(epi_y[pass].w); yo[7] = bf16_hi(epi_y[pass].w);
#pragma unroll
        for (int i = 0; i < 8; ++i) gy[i] = yo[i] > 0.f ? gy[i] * e.out_scale : 0.f;
      }
      float dx[8];
      vqaf_gn_bwd8(gy, xh, ga, lpg, gs, e.rstd_in[(size_t)mb * e.groups + c0 / gs], dx);
      if (m < g.M) {
        store8<true>(e.g_pre, (size_t)m * C + c0, gy);
        store8<true>(e.grad_x, (size_t)m * C + c0, dx);
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * (32 * MT) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (m < g.M && n < g.N) {
        const float v = acc[i][r] + bias_v;
        if (g.out_mode == GB_OUT_BF16) {
          reinterpret_cast<bf16_t*>(g.C)[(size_t)m * g.ldc + n] = f32_to_bf16(v);
        } else {
          float* p = reinterpret_cast<float*>(g.C) + (size_t)m * g.ldc + n;
          if (g.out_mode == GB_OUT_F32_ATOMIC) atomicAdd(p, v);
          else if (g.out_mode == GB_OUT_F32_ACC) *p += v;
          else *p = v;
        }
      }
      if (COLSUM && wn == 0 && (lane & 31) == 0 && m < g.M) {
        float* p = g.colsum + m;
        if (g.out_mode == GB_OUT_F32_ATOMIC) atomicAdd(p, csum[i][r]);
        else if (g.out_mode == GB_OUT_F32_ACC) *p += csum[i][r];
        else *p = csum[i][r];
      }
    }
  }
}

__device__ __forceinline__ void tile_of(const GemmArgs& g, const int id, int& tile_m, int& tile_n) {
  if (g.reg_m > 0) {
    const int xcd = id & 7, slot = id >> 3;
    const int bm = g.tiles_m / g.reg_m, bn = g.tiles_n / g.reg_n;
    tile_m = (xcd / g.reg_n) * bm + slot / bn;
    tile_n = (xcd % g.reg_n) * bn + slot % bn;
  } else {
    tile_m = id / g.tiles_n;
    tile_n = id % g.tiles_n;
  }
}

template <int MT, bool A_KMAJOR, bool B_KMAJOR, int STAGES = GB_STAGES>
__global__ void __launch_bounds__(2 * GB_THREADS)
gemm_bf16_kernel(GemmArgs g) {
  // Workgroup -> output tile.  Workgroups are dealt round-robin over the 8 XCDs, each with its own
  // L2, so ids with equal (id % 8) share an L2.  When the tile grid splits into 8 equal rectangular
  // regions, each XCD gets one region: its workgroups then share a few operand panels that fit its
  // L2, instead of every XCD streaming all of A (speed only; any mapping is correct).
  int tile_m, tile_n;
  tile_of(g, (int)blockIdx.x, tile_m, tile_n);
  // the column-sum variant of the loop runs only in the workgroups of the first tile column, as a
  // separate instantiation: a runtime condition around its MFMAs would make the compiler shuttle
  // the extra accumulators between register files at every k-step of every workgroup
  if (g.colsum != nullptr && tile_n == 0) gemm_bf16_body<MT, A_KMAJOR, B_KMAJOR, true, STAGES>(g, tile_m, tile_n);
  else gemm_bf16_body<MT, A_KMAJOR, B_KMAJOR, false, STAGES>(g, tile_m, tile_n);
}

template <int STAGES>
__global__ void __launch_bounds__(2 * GB_THREADS)
gemm_bf16_gn_kernel(GemmArgs g, GnEpi e) {
  int tile_m, tile_n;
  tile_of(g, (int)blockIdx.x, tile_m, tile_n);
  gemm_bf16_body<1, false, false, false, STAGES, 1>(g, tile_m, tile_n, &e);
}

template <int STAGES>
__global__ void __launch_bounds__(2 * GB_THREADS)
gemm_bf16_gn_bwd_kernel(GemmArgs g, GnEpi e) {
  int tile_m, tile_n;
  tile_of(g, (int)blockIdx.x, tile_m, tile_n);
  gemm_bf16_body<1, false, true, false, STAGES, 2>(g, tile_m, tile_n, &e);
}

// Several weight-gradient-shaped products (both operands k-major, the same contraction length) in ONE
// launch: the clipped sums of the classifier's Linear layers are a few hundred tiles and sixteen k-steps
// each, so a launch of their own costs each of them as much as its arithmetic.
constexpr int GB_GROUP_STAGES = 3;  // 72 KiB per workgroup: two workgroups (of different products) share a CU
struct GemmGroup {
  GemmArgs p[GB_MAX_GROUP];
  int first[GB_MAX_GROUP + 1];  // first workgroup id of each problem; first[n] = grid size
  int n;
};

template <int MT>
__global__ void __launch_bounds__(2 * GB_THREADS)
gemm_bf16_wgrad_group_kernel(GemmGroup G) {
  const int id = blockIdx.x;
  int pi = 0;
#pragma unroll
  for (int i = 1; i < GB_MAX_GROUP; ++i)
    if (i < G.n && id >= G.first[i]) pi = i;
  // uniform selects (no indexed access to the by-value argument)
  GemmArgs g = G.p[0];
  int base = G.first[0];
  if (pi == 1) { g = G.p[1]; base = G.first[1]; }
  if (pi == 2) { g = G.p[2]; base = G.first[2]; }
  if (pi == 3) { g = G.p[3]; base = G.first[3]; }
  int tile_m, tile_n;
  tile_of(g, id - base, tile_m, tile_n);
  if (g.colsum != nullptr && tile_n == 0) gemm_bf16_body<MT, true, true, true, GB_GROUP_STAGES>(g, tile_m, tile_n);
  else gemm_bf16_body<MT, true, true, false, GB_GROUP_STAGES>(g, tile_m, tile_n);
}

// The grouped clipped sums with the AdamW epilogue (64 x 64 tiles).
__global__ void __launch_bounds__(2 * GB_THREADS)
gemm_bf16_wgrad_adamw_kernel(GemmGroup G, AdamEpi E) {
  const int id = blockIdx.x;
  int pi = 0;
#pragma unroll
  for (int i = 1; i < GB_MAX_GROUP; ++i)
    if (i < G.n && id >= G.first[i]) pi = i;
  GemmArgs g = G.p[0];
  int base = G.first[0];
  long w_off = E.w_off[0], b_off = E.b_off[0];
  if (pi == 1) { g = G.p[1]; base = G.first[1]; w_off = E.w_off[1]; b_off = E.b_off[1]; }
  if (pi == 2) { g = G.p[2]; base = G.first[2]; w_off = E.w_off[2]; b_off = E.b_off[2]; }
  if (pi == 3) { g = G.p[3]; base = G.first[3]; w_off = E.w_off[3]; b_off = E.b_off[3]; }
  int tile_m, tile_n;
  tile_of(g, id - base, tile_m, tile_n);
  if (b_off >= 0 && tile_n == 0) gemm_bf16_body<1, true, true, true, GB_GROUP_STAGES, 3>(g, tile_m, tile_n, nullptr, &E, w_off, b_off);
  else gemm_bf16_body<1, true, true, false, GB_GROUP_STAGES, 3>(g, tile_m, tile_n, nullptr, &E, w_off, -1);
}

constexpr int GB_MAX_DEVICES = 64;
std::atomic<const unsigned int*> g_zero_ptr[GB_MAX_DEVICES];
std::mutex g_zero_mu;

int zero_block_on_current_device(const unsigned int** out) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= GB_MAX_DEVICES) return VQAF_E_LAUNCH;
  const unsigned int* p = g_zero_ptr[dev].load(std::memory_order_acquire);
  if (p == nullptr) {
    std::lock_guard<std::mutex> lock(g_zero_mu);
    p = g_zero_ptr[dev].load(std::memory_order_relaxed);
    if (p == nullptr) {
      void* addr = nullptr;
      if (hipGetSymbolAddress(&addr, HIP_SYMBOL(gb_zero_block)) != hipSuccess || addr == nullptr) {
        (void)hipGetLastError();
        return VQAF_E_LAUNCH;
      }
      p = reinterpret_cast<const unsigned int*>(addr);
      g_zero_ptr[dev].store(p, std::memory_order_release);
    }
  }
  *out = p;
  return VQAF_OK;
}

template <int MT, bool AK, bool BK_, int STAGES = GB_STAGES>
int launch_gemm(const GemmArgs& g, dim3 grid, hipStream_t st) {
  constexpr size_t smem = (size_t)STAGES * (64 * MT + GB_BN) * GB_BK * 2;
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<MT, AK, BK_, STAGES>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  });
  hipLaunchKernelGGL((gemm_bf16_kernel<MT, AK, BK_, STAGES>), grid, dim3(2 * GB_THREADS), smem, st, g);
  return hipGetLastError() == hipSuccess ? VQAF_OK : VQAF_E_LAUNCH;
}

template <int STAGES>
int launch_gemm_gn(const GemmArgs& g, const GnEpi& e, dim3 grid, hipStream_t st) {
  constexpr size_t smem = (size_t)STAGES * (64 + GB_BN) * GB_BK * 2;
  static_assert(smem >= 64 * 68 * sizeof(float), "the staged output tile lives in the operand ring");
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_gn_kernel<STAGES>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  });
  hipLaunchKernelGGL((gemm_bf16_gn_kernel<STAGES>), grid, dim3(2 * GB_THREADS), smem, st, g, e);
  return hipGetLastError() == hipSuccess ? VQAF_OK : VQAF_E_LAUNCH;
}

template <int STAGES>
int launch_gemm_gn_bwd(const GemmArgs& g, const GnEpi& e, dim3 grid, hipStream_t st) {
  constexpr size_t smem = (size_t)STAGES * (64 + GB_BN) * GB_BK * 2;
  static_assert(smem >= 64 * 68 * sizeof(float), "the staged output tile lives in the operand ring");
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_gn_bwd_kernel<STAGES>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  });
  hipLaunchKernelGGL((gemm_bf16_gn_bwd_kernel<STAGES>), grid, dim3(2 * GB_THREADS), smem, st, g, e);
  return hipGetLastError() == hipSuccess ? VQAF_OK : VQAF_E_LAUNCH;
}

// 8 equal regions of the tile grid (one per XCD), as square as the tile counts allow; 0, 0 if none divides
void pick_regions(int tiles_m, int tiles_n, int BM, int& reg_m, int& reg_n) {
  reg_m = reg_n = 0;
  const int cand[4][2] = {{2, 4}, {4, 2}, {1, 8}, {8, 1}};
  long best = -1;
  for (const auto& c : cand) {
    if (tiles_m % c[0] != 0 || tiles_n % c[1] != 0) continue;
    const long cost = (long)(tiles_m / c[0]) * BM + (long)(tiles_n / c[1]) * GB_BN;
    if (best < 0 || cost < best) { best = cost; reg_m = c[0]; reg_n = c[1]; }
  }
}

}  // namespace

// The same block of zeros for the library's other translation units (declared in common.h).
int vqaf_internal_zero_block(const unsigned int** out) { return zero_block_on_current_device(out); }

extern "C" int vqaf_linear_groupnorm_supported(long ldx, long ldw, int B, int C, int K, int groups) {
  if (groups <= 0 || C % groups != 0 || C % 64 != 0) return 0;
  const int gs = C / groups;
  if (!(gs == 8 || gs == 16 || gs == 32 || gs == 64)) return 0;
  return vqaf_gemm_bf16_supported(ldx, 0, ldw, 0, B, C, K);
}

extern "C" int vqaf_linear_groupnorm_fwd(const void* x, long ldx, const void* W, long ldw, const void* bias,
                                         int bias_dtype, int B, int C, int K, const void* gamma, const void* beta,
                                         int param_dtype, int groups, float eps, int relu, float dropout_p,
                                         uint64_t dropout_seed, const uint64_t* dropout_counter, void* y, void* xhat,
                                         float* rstd, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(x && W && gamma && beta && y && xhat && rstd, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_dtype_ok(param_dtype) && (bias == nullptr || vqaf_dtype_ok(bias_dtype)), VQAF_E_DTYPE);
  VQAF_REQUIRE(B > 0 && C > 0 && K > 0, VQAF_E_SHAPE);
  VQAF_REQUIRE(vqaf_linear_groupnorm_supported(ldx, ldw, B, C, K, groups), VQAF_E_ALIGN);
  VQAF_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, VQAF_E_SHAPE);
  VQAF_REQUIRE(dropout_p == 0.f || (dropout_counter != nullptr && relu), VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_aligned16(x) && vqaf_aligned16(W) && vqaf_aligned16(y) && vqaf_aligned16(xhat) &&
                   vqaf_aligned16(gamma) && vqaf_aligned16(beta), VQAF_E_ALIGN);
  const unsigned int* zeros = nullptr;
  const int rc = zero_block_on_current_device(&zeros);
  if (rc != VQAF_OK) return rc;
  GemmArgs g;
  g.A = reinterpret_cast<const bf16_t*>(x);
  g.B = reinterpret_cast<const bf16_t*>(W);
  g.bias = bias;
  g.C = nullptr;
  g.colsum = nullptr;
  g.zeros = zeros;
  g.lda = ldx; g.ldb = ldw; g.ldc = C;
  g.M = B; g.N = C; g.K = K;
  g.k_per_split = (K + GB_BK - 1) / GB_BK * GB_BK;
  g.tiles_m = (B + 63) / 64; g.tiles_n = C / 64;
  pick_regions(g.tiles_m, g.tiles_n, 64, g.reg_m, g.reg_n);
  g.bias_bf16 = bias_dtype == VQAF_BF16 ? 1 : 0;
  g.out_mode = GB_OUT_BF16;
  GnEpi e;
  e.gamma = gamma; e.beta = beta; e.y = y; e.xhat = xhat; e.rstd = rstd;
  e.drop_counter = dropout_p > 0.f ? dropout_counter : nullptr;
  e.drop_seed = (unsigned long long)dropout_seed;
  e.eps = eps; e.drop_p = dropout_p; e.groups = groups; e.relu = relu;
  e.param_bf16 = param_dtype == VQAF_BF16 ? 1 : 0;
  const dim3 grid((unsigned)(g.tiles_m * g.tiles_n), 1, 1);
  // ring depth as the plain forward GEMM picks it: three stages (three workgroups per CU) for 200-512 tiles of 128
  const long tiles128 = (long)((B + 127) / 128) * (C / 64);
  if (tiles128 >= 200 && tiles128 <= 512 && K <= 2048 && B <= 4096) return launch_gemm_gn<3>(g, e, grid, (hipStream_t)stream);
  return launch_gemm_gn<GB_STAGES>(g, e, grid, (hipStream_t)stream);
}

extern "C" int vqaf_linear_dgrad_groupnorm_supported(long ldg, long ldw, int B, int C, int K, int groups) {
  if (groups <= 0 || C % groups != 0 || C % 64 != 0) return 0;
  const int gs = C / groups;
  if (!(gs == 8 || gs == 16 || gs == 32 || gs == 64)) return 0;
  return vqaf_gemm_bf16_supported(ldg, 0, ldw, 1, B, C, K);
}

extern "C" int vqaf_linear_dgrad_groupnorm_bwd(const void* gy, long ldg, const void* W, long ldw, int B, int C, int K,
                                               const void* y, const void* xhat, const float* rstd, const void* gamma,
                                               int param_dtype, int groups, int relu, float out_scale, void* g_pre,
                                               void* grad_x, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(gy && W && xhat && rstd && gamma && g_pre && grad_x, VQAF_E_NULL);
  VQAF_REQUIRE(!relu || y != nullptr, VQAF_E_NULL);
  VQAF_REQUIRE(relu || out_scale == 1.0f, VQAF_E_SHAPE);
  VQAF_REQUIRE(vqaf_dtype_ok(param_dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(B > 0 && C > 0 && K > 0, VQAF_E_SHAPE);
  VQAF_REQUIRE(vqaf_linear_dgrad_groupnorm_supported(ldg, ldw, B, C, K, groups), VQAF_E_ALIGN);
  VQAF_REQUIRE(vqaf_aligned16(gy) && vqaf_aligned16(W) && vqaf_aligned16(xhat) && vqaf_aligned16(g_pre) &&
                   vqaf_aligned16(grad_x) && vqaf_aligned16(gamma) && (y == nullptr || vqaf_aligned16(y)), VQAF_E_ALIGN);
  const unsigned int* zeros = nullptr;
  const int rc = zero_block_on_current_device(&zeros);
  if (rc != VQAF_OK) return rc;
  GemmArgs g;
  g.A = reinterpret_cast<const bf16_t*>(gy);
  g.B = reinterpret_cast<const bf16_t*>(W);
  g.bias = nullptr;
  g.C = nullptr;
  g.colsum = nullptr;
  g.zeros = zeros;
  g.lda = ldg; g.ldb = ldw; g.ldc = C;
  g.M = B; g.N = C; g.K = K;
  g.k_per_split = (K + GB_BK - 1) / GB_BK * GB_BK;
  g.tiles_m = (B + 63) / 64; g.tiles_n = C / 64;
  pick_regions(g.tiles_m, g.tiles_n, 64, g.reg_m, g.reg_n);
  g.bias_bf16 = 0;
  g.out_mode = GB_OUT_BF16;
  GnEpi e = {};
  e.gamma = gamma; e.groups = groups; e.relu = relu; e.param_bf16 = param_dtype == VQAF_BF16 ? 1 : 0;
  e.y_in = relu ? y : nullptr; e.xhat_in = xhat; e.rstd_in = rstd; e.g_pre = g_pre; e.grad_x = grad_x;
  e.out_scale = out_scale;
  const dim3 grid((unsigned)(g.tiles_m * g.tiles_n), 1, 1);
  const long tiles128 = (long)((B + 127) / 128) * (C / 64);
  if (tiles128 >= 200 && tiles128 <= 512 && K <= 2048 && B <= 4096) return launch_gemm_gn_bwd<3>(g, e, grid, (hipStream_t)stream);
  return launch_gemm_gn_bwd<GB_STAGES>(g, e, grid, (hipStream_t)stream);
}

extern "C" int vqaf_gemm_bf16_supported(long lda, int a_kmajor, long ldb, int b_kmajor, int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0 || lda <= 0 || ldb <= 0) return 0;
  if (lda % 8 != 0 || ldb % 8 != 0) return 0;                       // 16-byte chunks, 16-byte aligned rows
  // a k-contiguous operand is fetched in 16-byte chunks of 8 k: its rows must hold K rounded up to 8
  // elements (the caller keeps the elements past K zero; the other operand's rows past K come from zeros)
  if (a_kmajor ? ((M + 7) / 8 * 8 > lda) : ((K + 7) / 8 * 8 > lda)) return 0;
  if (b_kmajor ? ((N + 7) / 8 * 8 > ldb) : ((K + 7) / 8 * 8 > ldb)) return 0;
  if (!a_kmajor && !b_kmajor && K % 8 != 0) return 0;  // then neither operand's tail is known to be zero
  return 1;
}

namespace {
// tile plan of one product; returns its tile count
long plan_wgrad(GemmArgs& g, int& MT, int M, int N, int K) {
  MT = (M > 4096 || K > 2048) ? 2 : 1;  // short contractions: 64 x 64 tiles, three workgroups per CU
  const int BM = 64 * MT;
  g.tiles_m = (M + BM - 1) / BM;
  g.tiles_n = (N + GB_BN - 1) / GB_BN;
  g.reg_m = g.reg_n = 0;
  const int cand[4][2] = {{2, 4}, {4, 2}, {1, 8}, {8, 1}};
  long best = -1;
  for (const auto& c : cand) {
    if (g.tiles_m % c[0] != 0 || g.tiles_n % c[1] != 0) continue;
    const long cost = (long)(g.tiles_m / c[0]) * BM + (long)(g.tiles_n / c[1]) * GB_BN;
    if (best < 0 || cost < best) { best = cost; g.reg_m = c[0]; g.reg_n = c[1]; }
  }
  g.M = M; g.N = N; g.K = K;
  g.k_per_split = (K + GB_BK - 1) / GB_BK * GB_BK;
  return (long)g.tiles_m * g.tiles_n;
}
}  // namespace

extern "C" int vqaf_clipped_sum_adamw_group(int n, const void* const* Gp, const long* ldg, const void* const* Ap,
                                            const long* lda, const int* Out, const int* In, int B,
                                            const long* w_off, const long* b_off, long noise_lo,
                                            const uint64_t* rng_state, float noise_std, float scale, float* param,
                                            float* exp_avg, float* exp_avg_sq, void* shadow_bf16, float* grad_out,
                                            float lr, float beta1, float beta2, float eps, float weight_decay,
                                            const uint64_t* step, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(Gp && ldg && Ap && lda && Out && In && w_off && b_off && param && exp_avg && exp_avg_sq && step, VQAF_E_NULL);
  VQAF_REQUIRE(n >= 1 && n <= GB_MAX_GROUP && B > 0, VQAF_E_SHAPE);
  VQAF_REQUIRE(noise_std == 0.f || rng_state != nullptr, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_aligned16(param) && vqaf_aligned16(exp_avg) && vqaf_aligned16(exp_avg_sq) &&
                   (grad_out == nullptr || vqaf_aligned16(grad_out)) &&
                   (shadow_bf16 == nullptr || vqaf_aligned16(shadow_bf16)), VQAF_E_ALIGN);
  const unsigned int* zeros = nullptr;
  const int rc = zero_block_on_current_device(&zeros);
  if (rc != VQAF_OK) return rc;
  GemmGroup G;
  G.n = n;
  AdamEpi E = {};
  int next = 0;
  long total_tiles = 0;
  for (int i = 0; i < n; ++i) {
    VQAF_REQUIRE(Gp[i] && Ap[i], VQAF_E_NULL);
    VQAF_REQUIRE(Out[i] > 0 && In[i] > 0 && In[i] % 8 == 0, VQAF_E_SHAPE);
    VQAF_REQUIRE(vqaf_gemm_bf16_supported(ldg[i], 1, lda[i], 1, Out[i], In[i], B), VQAF_E_ALIGN);
    VQAF_REQUIRE(vqaf_aligned16(Gp[i]) && vqaf_aligned16(Ap[i]), VQAF_E_ALIGN);
    // eight consecutive weights of a row are one 32-byte piece of the buckets and two whole generator calls
    VQAF_REQUIRE(w_off[i] >= noise_lo && (w_off[i] - noise_lo) % 4 == 0 && w_off[i] % 4 == 0, VQAF_E_ALIGN);
    VQAF_REQUIRE(b_off[i] < 0 || b_off[i] >= noise_lo, VQAF_E_SHAPE);
    GemmArgs& g = G.p[i];
    int MT = 1;
    total_tiles += plan_wgrad(g, MT, Out[i], In[i], B);
    VQAF_REQUIRE(MT == 1, VQAF_E_SHAPE);  // the epilogue is written for 64 x 64 tiles
    g.A = reinterpret_cast<const bf16_t*>(Gp[i]);
    g.B = reinterpret_cast<const bf16_t*>(Ap[i]);
    g.bias = nullptr;
    g.bias_bf16 = 0;
    g.C = nullptr;
    g.colsum = nullptr;
    g.zeros = zeros;
    g.lda = ldg[i]; g.ldb = lda[i]; g.ldc = In[i];
    g.out_mode = GB_OUT_F32;
    G.first[i] = next;
    next += g.tiles_m * g.tiles_n;
    E.w_off[i] = w_off[i];
    E.b_off[i] = b_off[i];
  }
  VQAF_REQUIRE(total_tiles >= 192, VQAF_E_SHAPE);
  for (int i = n; i <= GB_MAX_GROUP; ++i) G.first[i] = next;
  for (int i = n; i < GB_MAX_GROUP; ++i) { G.p[i] = G.p[0]; E.w_off[i] = E.w_off[0]; E.b_off[i] = -1; }
  E.P = param; E.M = exp_avg; E.V = exp_avg_sq; E.G = grad_out;
  E.shadow = reinterpret_cast<bf16_t*>(shadow_bf16);
  E.rng = rng_state; E.step = step;
  E.noise_std = noise_std; E.scale = scale; E.lr = lr; E.beta1 = beta1; E.beta2 = beta2; E.eps = eps;
  E.weight_decay = weight_decay; E.noise_lo = noise_lo;
  constexpr size_t smem = (size_t)GB_GROUP_STAGES * (64 + GB_BN) * GB_BK * 2;
  static_assert(smem >= (64 * 68 + 64) * sizeof(float), "the staged tile and its row sums live in the operand ring");
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_wgrad_adamw_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  });
  hipLaunchKernelGGL(gemm_bf16_wgrad_adamw_kernel, dim3((unsigned)next, 1, 1), dim3(2 * GB_THREADS), smem,
                     (hipStream_t)stream, G, E);
  VQAF_LAUNCH_CHECK();
}

extern "C" int vqaf_gemm_bf16_wgrad_group(int n, const void* const* A, const long* lda, const void* const* B,
                                          const long* ldb, const int* M, const int* N, int K, float* const* C,
                                          const long* ldc, int accumulate, float* const* colsum,
                                          vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(A && lda && B && ldb && M && N && C && ldc, VQAF_E_NULL);
  VQAF_REQUIRE(n >= 1 && n <= GB_MAX_GROUP && K > 0, VQAF_E_SHAPE);
  const unsigned int* zeros = nullptr;
  const int rc = zero_block_on_current_device(&zeros);
  if (rc != VQAF_OK) return rc;
  GemmGroup G;
  G.n = n;
  int mt_all = 0, next = 0;
  long total_tiles = 0;
  for (int i = 0; i < n; ++i) {
    VQAF_REQUIRE(A[i] && B[i] && C[i], VQAF_E_NULL);
    VQAF_REQUIRE(M[i] > 0 && N[i] > 0 && ldc[i] >= N[i], VQAF_E_SHAPE);
    VQAF_REQUIRE(vqaf_gemm_bf16_supported(lda[i], 1, ldb[i], 1, M[i], N[i], K), VQAF_E_ALIGN);
    VQAF_REQUIRE(vqaf_aligned16(A[i]) && vqaf_aligned16(B[i]), VQAF_E_ALIGN);
    GemmArgs& g = G.p[i];
    int MT = 1;
    // products of different tile heights need different kernels: the caller launches those one by one
    total_tiles += plan_wgrad(g, MT, M[i], N[i], K);
    VQAF_REQUIRE(i == 0 || MT == mt_all, VQAF_E_SHAPE);
    mt_all = MT;
    g.A = reinterpret_cast<const bf16_t*>(A[i]);
    g.B = reinterpret_cast<const bf16_t*>(B[i]);
    g.bias = nullptr;
    g.bias_bf16 = 0;
    g.C = C[i];
    g.colsum = colsum ? colsum[i] : nullptr;
    g.zeros = zeros;
    g.lda = lda[i]; g.ldb = ldb[i]; g.ldc = ldc[i];
    g.out_mode = accumulate ? GB_OUT_F32_ACC : GB_OUT_F32;
    G.first[i] = next;
    next += g.tiles_m * g.tiles_n;
  }
  // a launch with too few tiles altogether wants its contractions split: left to the single-product entry
  VQAF_REQUIRE(total_tiles >= 192, VQAF_E_SHAPE);
  for (int i = n; i <= GB_MAX_GROUP; ++i) G.first[i] = next;
  for (int i = n; i < GB_MAX_GROUP; ++i) G.p[i] = G.p[0];
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)next, 1, 1);
  if (mt_all == 2) {
    constexpr size_t smem = (size_t)GB_GROUP_STAGES * (128 + GB_BN) * GB_BK * 2;
    static std::once_flag once;
    std::call_once(once, [] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_wgrad_group_kernel<2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    });
    hipLaunchKernelGGL(gemm_bf16_wgrad_group_kernel<2>, grid, dim3(2 * GB_THREADS), smem, st, G);
  } else {
    constexpr size_t smem = (size_t)GB_GROUP_STAGES * (64 + GB_BN) * GB_BK * 2;
    static std::once_flag once;
    std::call_once(once, [] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_wgrad_group_kernel<1>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    });
    hipLaunchKernelGGL(gemm_bf16_wgrad_group_kernel<1>, grid, dim3(2 * GB_THREADS), smem, st, G);
  }
  VQAF_LAUNCH_CHECK();
}

extern "C" int vqaf_gemm_bf16(const void* A, long lda, int a_kmajor, const void* B, long ldb, int b_kmajor,
                              int M, int N, int K, const void* bias, int bias_dtype, void* C, long ldc,
                              int c_dtype, int accumulate, float* colsum, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(A && B && C, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_dtype_ok(c_dtype) && (bias == nullptr || vqaf_dtype_ok(bias_dtype)), VQAF_E_DTYPE);
  VQAF_REQUIRE(M > 0 && N > 0 && K > 0 && ldc >= N, VQAF_E_SHAPE);
  VQAF_REQUIRE(vqaf_gemm_bf16_supported(lda, a_kmajor, ldb, b_kmajor, M, N, K), VQAF_E_ALIGN);
  VQAF_REQUIRE(vqaf_aligned16(A) && vqaf_aligned16(B), VQAF_E_ALIGN);
  VQAF_REQUIRE(!(c_dtype == VQAF_BF16 && accumulate), VQAF_E_DTYPE);
  VQAF_REQUIRE(colsum == nullptr || a_kmajor, VQAF_E_SHAPE);
  const unsigned int* zeros = nullptr;
  const int rc = zero_block_on_current_device(&zeros);
  if (rc != VQAF_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  // tile height: 128 rows unless that leaves the chip half empty
  const long tiles128 = (long)((M + 127) / 128) * ((N + GB_BN - 1) / GB_BN);
  // A few hundred tiles over a short contraction (the classifier's own GEMMs: sixteen k-steps): a workgroup's
  // life is mostly pipeline fill and epilogue, so 64 x 64 tiles on a three-deep ring (48 KiB) put three
  // workgroups on a CU and overlap one's fill with another's arithmetic (measured: 0.428 -> 0.421 ms per step).
  const bool small3 = tiles128 >= 200 && tiles128 <= 512 && K <= 2048 && M <= 4096;
  const int MT = small3 ? 1 : ((tiles128 >= 200 || M > 4096) ? 2 : 1);
  const int BM = 64 * MT;
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + GB_BN - 1) / GB_BN;
  VQAF_REQUIRE((long)tiles_m * tiles_n < (1l << 30), VQAF_E_SHAPE);
  dim3 grid((unsigned)(tiles_m * tiles_n), 1, 1);
  // 8 equal regions of the tile grid, as square as the tile counts allow
  int reg_m = 0, reg_n = 0;
  {
    const int cand[4][2] = {{2, 4}, {4, 2}, {1, 8}, {8, 1}};
    long best = -1;
    for (const auto& c : cand) {
      if (tiles_m % c[0] != 0 || tiles_n % c[1] != 0) continue;
      // operand bytes one region streams: its A panels + its B panels
      const long cost = (long)(tiles_m / c[0]) * BM + (long)(tiles_n / c[1]) * GB_BN;
      if (best < 0 || cost < best) { best = cost; reg_m = c[0]; reg_n = c[1]; }
    }
  }
  // few tiles and an f32 output: split the contraction over workgroups (f32 atomics into a zeroed
  // output); each split keeps at least four k-steps
  int splits = 1;
  if (c_dtype == VQAF_F32) {
    const long tiles = (long)grid.x;
    while (tiles * splits < 192 && splits < 8 && K / (splits * 2) >= 4 * GB_BK) splits *= 2;
  }
  const int k_per_split = ((K + splits - 1) / splits + GB_BK - 1) / GB_BK * GB_BK;
  grid.z = splits;
  GemmArgs g;
  g.A = reinterpret_cast<const bf16_t*>(A);
  g.B = reinterpret_cast<const bf16_t*>(B);
  g.bias = bias;
  g.C = C;
  g.colsum = colsum;
  g.zeros = zeros;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.M = M; g.N = N; g.K = K;
  g.k_per_split = k_per_split;
  g.tiles_m = tiles_m; g.tiles_n = tiles_n; g.reg_m = reg_m; g.reg_n = reg_n;
  g.bias_bf16 = bias_dtype == VQAF_BF16 ? 1 : 0;
  if (c_dtype == VQAF_BF16) g.out_mode = GB_OUT_BF16;
  else if (splits > 1) g.out_mode = GB_OUT_F32_ATOMIC;
  else g.out_mode = accumulate ? GB_OUT_F32_ACC : GB_OUT_F32;
  if (splits > 1 && !accumulate) {  // atomics add into the output: start it from zero
    if (ldc == N) {
      if (hipMemsetAsync(C, 0, (size_t)M * N * 4, st) != hipSuccess) return VQAF_E_LAUNCH;
    } else {
      if (hipMemset2DAsync(C, (size_t)ldc * 4, 0, (size_t)N * 4, (size_t)M, st) != hipSuccess) return VQAF_E_LAUNCH;
    }
    if (colsum != nullptr && hipMemsetAsync(colsum, 0, (size_t)M * 4, st) != hipSuccess) return VQAF_E_LAUNCH;
  }
  if (splits > 1 && bias != nullptr) return VQAF_E_SHAPE;  // a bias would be added once per split
#define VQAF_GB(MTv)                                                                 \
  do {                                                                               \
    if (a_kmajor && b_kmajor) return launch_gemm<MTv, true, true>(g, grid, st);      \
    if (!a_kmajor && b_kmajor) return launch_gemm<MTv, false, true>(g, grid, st);    \
    if (!a_kmajor && !b_kmajor) return launch_gemm<MTv, false, false>(g, grid, st);  \
    return launch_gemm<MTv, true, false>(g, grid, st);                               \
  } while (0)
  if (small3) {
    if (a_kmajor && b_kmajor) return launch_gemm<1, true, true, 3>(g, grid, st);
    if (!a_kmajor && b_kmajor) return launch_gemm<1, false, true, 3>(g, grid, st);
    if (!a_kmajor && !b_kmajor) return launch_gemm<1, false, false, 3>(g, grid, st);
  }
  if (MT == 2) VQAF_GB(2);
  VQAF_GB(1);
#undef VQAF_GB
}
