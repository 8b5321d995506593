// Embedding lookups of the text encoder (model/text_encoder.py:14-48 wraps BertModel / RobertaModel, whose
// embeddings layer adds a word, a position and a token-type row per token) and their gradients.
//   forward : out(t,:) = word[ids0(t),:] + pos[ids1(t),:] + type[ids2(t),:], summed in f32, rounded once;
//   backward: table_grad[r,:] = sum of g(t,:) over the tokens t with ids(t) == r, in token order, so the
//             result does not depend on scheduling.  The first token carrying an id owns that id's row: its
//             workgroup marks every later token with the same id in a bitmap and adds their rows in order
//             (f32), the other workgroups leave at once.  Tables with at most 8 rows (token types) would make
//             one workgroup add every token, so they use a column-sum kernel with a fixed two-level order.
#include "common.h"

namespace {

constexpr int EM_MAX_TOKENS = 32768;   // bitmap of the owner kernel: 4 KB of LDS

template <bool PBF16, bool OBF16>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const long* ids0, const void* t0, const long* ids1,
                                                        const void* t1, const long* ids2, const void* t2,
                                                        long rows0, long rows1, long rows2, int T, int E,
                                                        void* out) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= T) return;
  // an index outside its table reads the nearest row instead of memory outside the table
  const size_t r0 = (size_t)min(max(ids0[row], 0L), rows0 - 1) * E;
  const size_t r1 = t1 != nullptr ? (size_t)min(max(ids1[row], 0L), rows1 - 1) * E : 0;
  const size_t r2 = t2 != nullptr ? (size_t)min(max(ids2[row], 0L), rows2 - 1) * E : 0;
  for (int c = lane * 8; c < E; c += 512) {
    float a[8], b[8];
    load8<PBF16>(t0, r0 + c, a);
    if (t1 != nullptr) {
      load8<PBF16>(t1, r1 + c, b);
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += b[e];
    }
    if (t2 != nullptr) {
      load8<PBF16>(t2, r2 + c, b);
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += b[e];
    }
    store8<OBF16>(out, (size_t)row * E + c, a);
  }
}

template <bool GBF16, bool PBF16>
__global__ __launch_bounds__(256) void embed_bwd_owner_kernel(const long* ids, int T, const void* g, int E,
                                                              long rows, long pad_idx, void* table_grad) {
  __shared__ unsigned bits[EM_MAX_TOKENS / 32];
  const int t = blockIdx.x, tid = threadIdx.x;
  const long id = ids[t];
  if (id == pad_idx || id < 0 || id >= rows) return;   // the padding row keeps its zero gradient
  int earlier = 0;
  for (int u = tid; u < t; u += 256) earlier |= ids[u] == id;
  if (__syncthreads_or(earlier)) return;   // an earlier token owns this id
  const int w_lo = t >> 5, w_hi = (T + 31) >> 5;
  for (int w = w_lo + tid; w < w_hi; w += 256) bits[w] = 0u;
  __syncthreads();
  for (int u = t + tid; u < T; u += 256)
    if (ids[u] == id) atomicOr(&bits[u >> 5], 1u << (u & 31));
  __syncthreads();
  for (int c = tid * 8; c < E; c += 2048) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int w = w_lo; w < w_hi; ++w) {
      unsigned m = bits[w];
      while (m != 0u) {
        const int u = (w << 5) + __builtin_ctz(m);
        m &= m - 1u;
        float v[8];
        load8<GBF16>(g, (size_t)u * E + c, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += v[e];
      }
    }
    store8<PBF16>(table_grad, (size_t)id * E + c, acc);
  }
}

// grid (E / 16, rows): 64 token groups x 4 threads of 4 columns; group j adds tokens j, j+64, ... in order,
// then the 64 partial sums are added in group order
template <bool GBF16, bool PBF16>
__global__ __launch_bounds__(256) void embed_bwd_small_kernel(const long* ids, int T, const void* g, int E,
                                                              void* table_grad) {
  __shared__ float part[64][16];
  const int tid = threadIdx.x, grp = tid >> 2, cl = (tid & 3) * 4;
  const int col = blockIdx.x * 16 + cl;   // E % 8 == 0: a thread's 4 columns are inside the row or all outside
  const long r = blockIdx.y;
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  if (col < E) {
    for (int u = grp; u < T; u += 64) {
      if (ids[u] != r) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] += load1<GBF16>(g, (size_t)u * E + col + e);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) part[grp][cl + e] = a[e];
  __syncthreads();
  if (tid < 16 && blockIdx.x * 16 + tid < E) {
    float s = 0.f;
#pragma unroll 8
    for (int j = 0; j < 64; ++j) s += part[j][tid];
    store1<PBF16>(table_grad, (size_t)r * E + blockIdx.x * 16 + tid, s);
  }
}

}  // namespace

extern "C" int vqaf_embed_fwd(const long* ids0, const void* table0, long rows0, const long* ids1, const void* table1,
                              long rows1, const long* ids2, const void* table2, long rows2, int T, int E,
                              int param_dtype, void* out, int out_dtype, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(ids0 && table0 && out, VQAF_E_NULL);
  VQAF_REQUIRE((ids1 != nullptr) == (table1 != nullptr) && (ids2 != nullptr) == (table2 != nullptr), VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_dtype_ok(param_dtype) && vqaf_dtype_ok(out_dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(T > 0 && E > 0 && E % 8 == 0 && rows0 > 0 && (!table1 || rows1 > 0) && (!table2 || rows2 > 0),
               VQAF_E_SHAPE);
  VQAF_REQUIRE(vqaf_aligned16(table0) && vqaf_aligned16(out) && (!table1 || vqaf_aligned16(table1)) &&
               (!table2 || vqaf_aligned16(table2)), VQAF_E_ALIGN);
  const dim3 grid((unsigned)((T + 3) / 4));
  hipStream_t st = (hipStream_t)stream;
  const bool pb = param_dtype == VQAF_BF16, ob = out_dtype == VQAF_BF16;
#define EM_FWD(P, O) \
  hipLaunchKernelGGL((embed_fwd_kernel<P, O>), grid, dim3(256), 0, st, ids0, table0, ids1, table1, ids2, table2, rows0, rows1, rows2, T, E, out)
  if (pb && ob) EM_FWD(true, true);
  else if (pb) EM_FWD(true, false);
  else if (ob) EM_FWD(false, true);
  else EM_FWD(false, false);
#undef EM_FWD
  VQAF_LAUNCH_CHECK();
}

extern "C" int vqaf_embed_bwd(const long* ids, int T, const void* g, int g_dtype, int E, long rows, long pad_idx,
                              void* table_grad, int param_dtype, vqaf_stream_t stream) {
  VQAF_ENTRY();
  VQAF_REQUIRE(ids && g && table_grad, VQAF_E_NULL);
  VQAF_REQUIRE(vqaf_dtype_ok(g_dtype) && vqaf_dtype_ok(param_dtype), VQAF_E_DTYPE);
  VQAF_REQUIRE(T > 0 && T <= EM_MAX_TOKENS && E > 0 && E % 8 == 0 && rows > 0 && rows <= 0x7fffffffL, VQAF_E_SHAPE);
  VQAF_REQUIRE(vqaf_aligned16(g) && vqaf_aligned16(table_grad), VQAF_E_ALIGN);
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(table_grad, 0, (size_t)rows * E * vqaf_esize(param_dtype), st) != hipSuccess) return VQAF_E_LAUNCH;
  const bool gb = g_dtype == VQAF_BF16, pb = param_dtype == VQAF_BF16;
  if (rows <= 8) {
    const dim3 grid((unsigned)((E + 15) / 16), (unsigned)rows);
#define EM_SM(G, P) hipLaunchKernelGGL((embed_bwd_small_kernel<G, P>), grid, dim3(256), 0, st, ids, T, g, E, table_grad)
    if (gb && pb) EM_SM(true, true);
    else if (gb) EM_SM(true, false);
    else if (pb) EM_SM(false, true);
    else EM_SM(false, false);
#undef EM_SM
  } else {
#define EM_OW(G, P) \
  hipLaunchKernelGGL((embed_bwd_owner_kernel<G, P>), dim3((unsigned)T), dim3(256), 0, st, ids, T, g, E, rows, pad_idx, table_grad)
    if (gb && pb) EM_OW(true, true);
    else if (gb) EM_OW(true, false);
    else if (pb) EM_OW(false, true);
    else EM_OW(false, false);
#undef EM_OW
  }
  VQAF_LAUNCH_CHECK();
}
