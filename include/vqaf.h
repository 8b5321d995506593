/*
 * libvqafuse — C ABI of the MI355X (gfx950) kernels behind the VQA fusion hot path.
 *
 * The reference (hanleju/VQA) has no FFI: its boundary for this path is the Python
 * nn.Module protocol of utils/fusion.py and the Opacus call sites in train.py.  Each
 * entry point below names the reference lines whose arithmetic it replaces; the Python
 * host layer in vqa_amd/ binds these symbols with ctypes and re-creates the reference's
 * module interface on top of them.
 *
 * Conventions (all entry points):
 *   - return 0 on success, a negative VQAF_E_* code otherwise; never throw, never exit;
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - tensors are dense row-major; the caller owns every buffer, workspaces included;
 *     the library allocates nothing.  Between calls it keeps only read-mostly process state:
 *     tuning knobs read once from the environment (VQAF_HS_TOKB, VQAF_GM_LPR), a per-kernel
 *     "dynamic LDS limit raised" flag, and the per-device address of the fixed scratch array of
 *     vqaf_score_noalloc -- all initialised thread-safely on first use;
 *   - work is only enqueued on `stream` (a hipStream_t passed as void*); no entry point
 *     synchronises with the host, so all of them may be captured into a HIP graph;
 *   - `dtype` is the storage type of token / activation tensors (VQAF_F32 or VQAF_BF16);
 *     accumulation is always f32;
 *   - index tensors are int64 (torch.long), as torch.topk / torch.gather use.
 */
#ifndef VQAF_H_
#define VQAF_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VQAF_VERSION 100 /* major*10000 + minor*100 + patch */

enum vqaf_status {
  VQAF_OK = 0,
  VQAF_E_SHAPE = -1,  /* a size is zero/negative or outside what the kernels support */
  VQAF_E_DTYPE = -2,  /* dtype code is not VQAF_F32 / VQAF_BF16 */
  VQAF_E_ALIGN = -3,  /* a pointer or row pitch is not 16-byte aligned */
  VQAF_E_LAUNCH = -4, /* the HIP runtime refused the launch (hipGetLastError != 0) */
  VQAF_E_NULL = -5,   /* a required pointer is NULL */
  VQAF_E_SCRATCH = -6 /* the sizes need more than the fixed device scratch of a *_noalloc entry point */
};

enum vqaf_dtype { VQAF_F32 = 0, VQAF_BF16 = 1 };

enum vqaf_noise_kind { VQAF_NOISE_NONE = 0, VQAF_NOISE_GAUSSIAN = 1, VQAF_NOISE_UNIFORM = 2 };

typedef void* vqaf_stream_t; /* hipStream_t */

int vqaf_version(void);
const char* vqaf_status_string(int status);

/* ------------------------------------------------------------------------------------
 * Token scoring (single-query multi-head attention weights, head-averaged).
 * Replaces the `nn.MultiheadAttention(query=q_global[:,None], key=seq, value=seq)` calls
 * whose OUTPUT is discarded and only the (B,1,N) weights are used:
 *   utils/fusion.py:234-238 (text->vision), :243-247 (vision->text), :159-163 (Attention).
 *
 *   w[b,n] = (1/H) sum_h softmax_n( (Wq_h q_b + bq_h) . (Wk_h x_bn + bk_h) / sqrt(D/H) )
 *
 * The key bias contributes a per-(b,h) constant that cancels in the softmax, and the
 * value / output projections never reach the weights, so the kernels evaluate
 *   u[b,h,:] = Wk_h^T (Wq_h q_b + bq_h) / sqrt(D/H),   s[b,n,h] = x_bn . u[b,h,:]
 * in one pass over the tokens.
 * ------------------------------------------------------------------------------------ */

/* u (B,H,D) f32  <-  q (B,D) `dtype`, in_proj_weight (3D,D) / in_proj_bias (3D) `w_dtype`.
 * `qproj_ws` is a (B,D) f32 workspace. */
int vqaf_fold_query(const void* q, const void* in_proj_weight, const void* in_proj_bias,
                    int B, int D, int H, int dtype, int w_dtype, float* qproj_ws, float* u,
                    vqaf_stream_t stream);

/* vqaf_fold_query for n <= 4 (query, in_proj) pairs of one shape in two launches instead of 2n
 * (CoAttention folds its text->vision and vision->text queries together). */
typedef struct vqaf_fold_desc {
  const void* q;              /* (B,D) `dtype` */
  const void* in_proj_weight; /* (3D,D) `w_dtype` */
  const void* in_proj_bias;   /* (3D) `w_dtype` */
  float* qproj_ws;            /* (B,D) f32 workspace */
  float* u;                   /* (B,H,D) f32 out */
} vqaf_fold_desc;
int vqaf_fold_query_multi(const vqaf_fold_desc* descs_host, int n, int B, int D, int H, int dtype,
                          int w_dtype, vqaf_stream_t stream);
/* The same call that also adds `delta` to count_state[1] ({seed, counter} uint64 pair in device memory) on
 * the side: the generator positions the launch BEHIND it is about to consume (vqaf_fused_prune_pool_multi
 * with rng_advance = -delta draws from the position the counter had before), so that no launch of its own
 * is spent on the counter.  utils/fusion.py:48-65 draws fresh positions in every forward; this is that
 * bookkeeping for a replayed HIP graph.  The folds themselves never read count_state. */
int vqaf_fold_query_multi_advance(const vqaf_fold_desc* descs_host, int n, int B, int D, int H, int dtype,
                                  int w_dtype, uint64_t* count_state, uint64_t delta, vqaf_stream_t stream);

/* s (B,N,H) f32  <-  seq (B,N,D) `dtype`, u (B,H,D) f32.  Raw per-head scaled scores.
 * With bf16 tokens, H <= 8 and D in {64,128,256,512,768,1024} the products run on bf16 MFMA:
 * tokens are exact, u enters as two bf16 terms bf16(u) + bf16(u - bf16(u)) (accurate to ~2^-17);
 * otherwise on the f32 vector ALU. */
int vqaf_head_scores(const void* seq, const float* u, int B, int N, int D, int H, int dtype,
                     float* s, vqaf_stream_t stream);

/* vqaf_head_scores for n <= 4 token sets (same B, D, H, dtype; different N) in one launch. */
typedef struct vqaf_scores_desc {
  const void* seq; /* (B,N,D) */
  const float* u;  /* (B,H,D) */
  float* s;        /* (B,N,H) out */
  int N;
} vqaf_scores_desc;
int vqaf_head_scores_multi(const vqaf_scores_desc* descs_host, int n, int B, int D, int H, int dtype,
                           vqaf_stream_t stream);

/* w (B,N) f32  <-  s (B,N,H) f32: softmax over n per head, then mean over heads. */
int vqaf_attn_weights(const float* s, int B, int N, int H, float* w, vqaf_stream_t stream);

/* Bytes of workspace vqaf_score needs (qproj + u + s), 256-byte aligned pieces. */
size_t vqaf_score_workspace_bytes(int B, int N, int D, int H);

/* The three steps above in one call: w (B,N) f32 <- q, seq, in_proj_{weight,bias}. */
int vqaf_score(const void* q, const void* seq, const void* in_proj_weight,
               const void* in_proj_bias, int B, int N, int D, int H, int dtype, int w_dtype,
               void* workspace, size_t workspace_bytes, float* w, vqaf_stream_t stream);

/* vqaf_score for callers that cannot allocate device memory and have no workspace to hand over
 * (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  It has
 * no counterpart in the reference.  Same arithmetic and same results as vqaf_score; qproj, u and
 * s live in a fixed VQAF_SCORE_SCRATCH_BYTES array inside the library's code object (one copy
 * per GPU, held until the process ends).  The call allocates and frees nothing.
 *   - sizes with vqaf_score_workspace_bytes(B,N,D,H) > VQAF_SCORE_SCRATCH_BYTES are refused with
 *     VQAF_E_SCRATCH before any device work (32 MiB covers B=1024, N=197 and B=512, N=577 at
 *     D=512, H=8);
 *   - the device that owns `stream` must be current; the first call on a device resolves the
 *     array's address (make that call outside stream capture);
 *   - every call on a GPU uses that GPU's one array: calls that overlap on one GPU (two streams,
 *     two threads) race on it.  Calls on one stream are ordered and safe. */
#define VQAF_SCORE_SCRATCH_BYTES (32u << 20)
int vqaf_score_noalloc(const void* q, const void* seq, const void* in_proj_weight,
                       const void* in_proj_bias, int B, int N, int D, int H, int dtype, int w_dtype,
                       float* w, vqaf_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Keep-index selection.  Replaces torch.topk(attn_weights, k) (utils/fusion.py:44) and
 * torch.topk(attn_weights, N-k, largest=False) (:51).
 *   top    (B,K)   int64: indices of the K largest weights, weight descending;
 *   bottom (B,N-K) int64: indices of the N-K smallest, weight ascending (may be NULL).
 * Order is total and deterministic: (weight desc, index asc); `bottom` is the tail of that
 * same ranking reversed, so top and bottom always partition [0,N).  NaN ranks above +inf
 * (as torch.topk does); -0.0 ties with +0.0.  1 <= K <= N <= 4096.
 * ------------------------------------------------------------------------------------ */
int vqaf_topk(const float* w, int B, int N, int K, int64_t* top, int64_t* bottom,
              vqaf_stream_t stream);

/* Adversarial index mixing, batched.  Replaces the per-sample Python loop
 * utils/fusion.py:56-65:  top[b, pos[b,j]] = bottom[b, src[b,j]]  for j < k_adv.
 * pos (B,k_adv) int64 holds distinct positions in [0,K); src (B,k_adv) int64 distinct
 * positions in [0,R).  The caller skips the call when R < k_adv, as the reference does
 * (:57).  Out-of-range pos/src entries are ignored. */
int vqaf_adv_swap(int64_t* top, const int64_t* bottom, const int64_t* pos, const int64_t* src,
                  int B, int K, int R, int k_adv, vqaf_stream_t stream);

/* Same, drawing pos/src inside the kernel: per sample, the first k_adv entries of a uniform
 * random permutation of [0,K) and of [0,R) (what randperm(k)[:k_adv] / randperm(R)[:k_adv]
 * at utils/fusion.py:59,62 produce), from the counter-based generator keyed by
 * rng_state = {seed, counter} (two uint64 in device memory; not modified).
 * Optionally records the draws in pos_out/src_out (B,k_adv) int64 (NULL to skip). */
int vqaf_adv_swap_random(int64_t* top, const int64_t* bottom, int B, int K, int R, int k_adv,
                         const uint64_t* rng_state, int64_t* pos_out, int64_t* src_out,
                         vqaf_stream_t stream);

/* vqaf_attn_weights (optional) + vqaf_topk + vqaf_adv_swap[_random] in ONE launch, one workgroup per
 * sample; weights and index lists stay in LDS in between.  Exactly one of `s` (B,N,H head scores)
 * and `w_in` (B,N weights) is given.  adv_mode: 0 = no swap, 1 = injected pos/src (B,k_adv),
 * 2 = in-kernel draws from rng_state.  w_out (B,N), bottom (B,N-K), pos_out/src_out may be NULL.
 * Same results as the separate entry points (same ranking, same draws for the same rng_state).
 * `salt` is added to the seed: callers that select for several modalities from one rng_state give
 * each its own salt and advance the counter once afterwards. */
int vqaf_select(const float* s, const float* w_in, int B, int N, int H, int K, int adv_mode, int k_adv,
                const int64_t* pos, const int64_t* src, const uint64_t* rng_state, uint64_t salt,
                float* w_out, int64_t* top, int64_t* bottom, int64_t* pos_out, int64_t* src_out,
                vqaf_stream_t stream);

/* vqaf_select for n <= 4 modalities (same B, H and rng_state; per-modality fields as the arguments
 * of vqaf_select) in one launch. */
typedef struct vqaf_select_desc {
  const float* s;
  const float* w_in;
  const int64_t* pos;
  const int64_t* src;
  float* w_out;
  int64_t* top;
  int64_t* bottom;
  int64_t* pos_out;
  int64_t* src_out;
  uint64_t salt;
  int N, K, adv_mode, k_adv;
} vqaf_select_desc;
int vqaf_select_multi(const vqaf_select_desc* descs_host, int n, int B, int H, const uint64_t* rng_state,
                      vqaf_stream_t stream);

/* counter += delta on the device (stream-ordered; graph-capturable). */
int vqaf_rng_advance(uint64_t* rng_state, uint64_t delta, vqaf_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Gather + optional noise + "remaining mean" mix token.  Replaces
 * utils/fusion.py:67-92 (gather, _add_privacy_noise, mask/scatter_/masked mean/cat):
 *   out[b,j,:] = seq[b, idx[b,j], :] (+ noise)                         j < K
 *   out[b,K,:] = alpha * sum_{n not in idx[b]} seq[b,n,:] / (R_b + 1e-10)   if alpha > 0
 * where R_b = N - |set(idx[b])|.  out is (B, K+1, D) when alpha > 0, else (B, K, D).
 * Gathered rows are bit-exact copies when no noise is added.  The mix token is summed in
 * f32 in a fixed order (deterministic run to run).
 * Noise: `noise` (B,K,D) f32 is added if non-NULL; else if noise_kind != NONE, N(0,scale)
 * or U(-scale,scale) is generated in the kernel from rng_state {seed, counter}.
 * out_dtype: storage type of out (VQAF_F32 is always allowed; VQAF_BF16 only with
 * dtype == VQAF_BF16).  Indices outside [0,N) produce a zero row and are not masked.
 * ------------------------------------------------------------------------------------ */
int vqaf_gather_mix(const void* seq, const int64_t* idx, const float* noise, int noise_kind,
                    float noise_scale, const uint64_t* rng_state, float alpha, int B, int N,
                    int K, int D, int dtype, int out_dtype, void* out, vqaf_stream_t stream);

/* Backward of vqaf_gather_mix w.r.t. seq (autograd of utils/fusion.py:67-92):
 *   grad_seq[b,n,:] = sum_{j: idx[b,j]==n} grad_out[b,j,:]
 *                   + [n not in idx[b]] * alpha/(R_b+1e-10) * grad_out[b,K,:]
 * grad_out is (B,K+1,D) (or (B,K,D) when alpha <= 0) of `g_dtype`; grad_seq (B,N,D) of
 * `dtype`. */
int vqaf_gather_mix_bwd(const void* grad_out, const int64_t* idx, float alpha, int B, int N,
                        int K, int D, int g_dtype, int dtype, void* grad_seq,
                        vqaf_stream_t stream);

/* Mean over the token axis written into a column block of the fused feature:
 *   out[b, col0 + d] = (1/M) sum_j x[b,j,d];   x (B,M,D) `dtype`, out (B,ld_out) `out_dtype`.
 * Replaces `.mean(dim=1)` + torch.cat at utils/fusion.py:253-254,274 (:170,182). */
int vqaf_mean_pool(const void* x, int B, int M, int D, int dtype, void* out, int ld_out,
                   int col0, int out_dtype, vqaf_stream_t stream);

/* Backward of vqaf_mean_pool: grad_x[b,j,d] = grad_out[b, col0+d] / M. */
int vqaf_mean_pool_bwd(const void* grad_out, int ld_out, int col0, int g_dtype, int B, int M,
                       int D, int dtype, void* grad_x, vqaf_stream_t stream);

/* Pruning branch of one modality with the pooled output only (the (B,K+1,D) tensor is
 * never written):  out[b, col0+d] = mean_j PrivacyAwareTokenPruning(seq, idx)[b,j,d].
 * Same arguments as vqaf_gather_mix; utils/fusion.py:240,253 (and :249,254). */
int vqaf_gather_mix_pool(const void* seq, const int64_t* idx, const float* noise,
                         int noise_kind, float noise_scale, const uint64_t* rng_state,
                         float alpha, int B, int N, int K, int D, int dtype, void* out,
                         int ld_out, int col0, int out_dtype, vqaf_stream_t stream);

/* Backward of vqaf_gather_mix_pool w.r.t. seq. */
int vqaf_gather_mix_pool_bwd(const void* grad_out, int ld_out, int col0, int g_dtype,
                             const int64_t* idx, float alpha, int B, int N, int K, int D,
                             int dtype, void* grad_seq, vqaf_stream_t stream);

/* Several modalities pooled by one launch (the vision and the text half of CoAttention's fused
 * feature, utils/fusion.py:240-254, 274): same arithmetic as one vqaf_gather_mix_pool call per
 * entry of `mods_host` (a HOST array of n_mod <= 4 descriptors; all modalities share B, D, dtype
 * and the output (B, ld_out)).  The backward twin writes grad_seq of every modality whose
 * pointer is non-NULL. */
typedef struct vqaf_pool_mod {
  const void* seq;           /* (B,N,D) */
  const int64_t* idx;        /* (B,K)   */
  const float* noise;        /* (B,K,D) f32 or NULL */
  const uint64_t* rng_state; /* {seed, counter} or NULL */
  float noise_scale, alpha;
  int noise_kind, N, K, col0;
} vqaf_pool_mod;
typedef struct vqaf_pool_bwd_mod {
  const int64_t* idx; /* (B,K) */
  void* grad_seq;     /* (B,N,D) or NULL */
  float alpha;
  int N, K, col0;
} vqaf_pool_bwd_mod;
int vqaf_gather_mix_pool_multi(const vqaf_pool_mod* mods_host, int n_mod, int B, int D, int dtype,
                               void* out, int ld_out, int out_dtype, vqaf_stream_t stream);
int vqaf_gather_mix_pool_bwd_multi(const void* grad_out, int ld_out, int g_dtype,
                                   const vqaf_pool_bwd_mod* mods_host, int n_mod, int B, int D,
                                   int dtype, vqaf_stream_t stream);

/* PrivacyAwareTokenPruning.forward (utils/fusion.py:26-94) in ONE launch: vqaf_select on given
 * weights followed by vqaf_gather_mix, without the gather waiting for a ranking launch: every lane
 * issues all row loads of its (sample, 128-column slice) first and the ranking runs while they are
 * in flight.  w (B,N) f32 attention weights; out (B, K + (alpha > 0), D); idx_out (B,K) int64 keep
 * indices.  Same results as vqaf_select + vqaf_gather_mix: same ranking, same draws for the same
 * rng states (noise_counter_offset as in vqaf_fused_mod), kept rows bit-exact copies.
 * vqaf_prune_mix_supported (1 / 0): D % 8 == 0, a sample below 2 GiB, and the per-sample selection
 * state within 64 KiB of LDS for EVERY K <= N and k_adv <= N / 2 (8 * nextpow2(N) + 20 * N + about
 * 16.5 KiB: N up to 1630), so a supported size never turns into a shape error at the launch; other
 * sizes return VQAF_E_SHAPE and callers use the two entry points (vqaf_select + vqaf_gather_mix). */
int vqaf_prune_mix_supported(int N, int D, int dtype);
int vqaf_prune_mix(const void* seq, const float* w, int B, int N, int K, int D, int dtype, int adv_mode,
                   int k_adv, const int64_t* pos, const int64_t* src, const uint64_t* sel_rng_state,
                   uint64_t salt, const float* noise, int noise_kind, float noise_scale,
                   const uint64_t* noise_rng_state, int noise_counter_offset, float alpha, int out_dtype,
                   void* out, int64_t* idx_out, vqaf_stream_t stream);

/* The whole pruning branch of a modality -- vqaf_head_scores, vqaf_select and
 * vqaf_gather_mix_pool -- in ONE launch, one workgroup per sample (utils/fusion.py:234-240 with
 * :253; :243-249 with :254; :159-170).  The second pass over a sample's tokens follows the first
 * within microseconds, so it is served from cache and HBM sees each token once.
 * Same keep indices as vqaf_select (same ranking code, same draws for the same rng state and salt);
 * the pooled features equal vqaf_gather_mix_pool's up to f32 summation order (noise is summed per
 * keep slot after the rows instead of row by row).  Per modality: seq (B,N,D) bf16, u (B,H,D) f32
 * from vqaf_fold_query, idx_out (B,K) int64 receives the keep indices, w_out (B,N) f32 the
 * attention weights (may be NULL); the other fields as in vqaf_select_desc / vqaf_pool_mod.
 * `sel_rng_state` serves adv_mode 2 of every modality (salted), as in vqaf_select_multi.
 * Limits: dtype VQAF_BF16, H <= 8, D in {64,128,256,512,1024}, N <= 4096 and about 64 KiB of LDS
 * per sample (N*H*4 + 8*nextpow2(N) + 12*N + 32*(D+8) bytes); other shapes return VQAF_E_SHAPE /
 * VQAF_E_DTYPE and the caller uses the separate entry points. */
typedef struct vqaf_fused_mod {
  const void* seq;
  const float* u;
  const int64_t* pos;
  const int64_t* src;
  const float* noise;
  const uint64_t* noise_rng_state;
  int64_t* idx_out;
  float* w_out;
  uint64_t salt;
  float alpha, noise_scale;
  int noise_kind, N, K, adv_mode, k_adv, col0;
  int noise_counter_offset; /* added to the noise stream's counter: 1 when the selection draws from the
                               same rng state and the caller advances it once for them (what the
                               separate launches see: select, advance, pool), else 0 */
} vqaf_fused_mod;
/* rng_advance > 0: `sel_rng_state` is a FOUR-word state {seed, counter, ticket, reserved} (ticket zero)
 * and the last workgroup of the launch adds rng_advance to its counter, which replaces the
 * vqaf_rng_advance launch(es) that would follow; every noise_rng_state must then be the same state.
 * rng_advance == 0: two words suffice and the caller advances.
 * rng_advance < 0: the counter of `sel_rng_state` (two words) was ALREADY advanced by -rng_advance before
 * this launch (vqaf_fold_query_multi_advance); selection and any noise stream that is the same state draw from
 * the positions counter + rng_advance ..., i.e. the values of the rng_advance == 0 call followed by the
 * caller's advance. */
int vqaf_fused_prune_pool_multi(const vqaf_fused_mod* mods_host, int n_mod, int B, int D, int H, int dtype,
                                uint64_t* sel_rng_state, int rng_advance, void* out, int ld_out,
                                int out_dtype, vqaf_stream_t stream);
/* The same call on a ONE-PASS kernel (utils/fusion.py:234-254, 274, as above): a sample's rows are read
 * from memory once and stay in registers / LDS of the compute unit between scoring and pooling (a wave
 * keeps three 16-token tiles as raw bf16 in registers, a thirteenth tile lives in LDS), and the pooled
 * sums are formed on the matrix cores (tiles pass through a small LDS image and come back through the
 * transposing LDS read as the operand of weights x tokens; every product is exact).  Identical keep
 * indices, random streams and output layout; the pooled features agree with the two-pass kernel's to
 * f32 summation order.  It halves the kernel's traffic over the memory fabric; at 256 registers per
 * lane only two workgroups fit a compute unit, and on MI355X its launch takes about as long as the
 * two-pass one, which is why vqaf_fused_prune_pool_multi stays on the two-pass kernel and this one is
 * asked for by name.  Limits: D == 512, every N <= 208, H <= 8 (vqaf_fused_prune_pool_onepass_supported:
 * 1 / 0); other shapes return VQAF_E_SHAPE. */
int vqaf_fused_prune_pool_onepass_supported(int n_mod, const int* N, int D, int H);
int vqaf_fused_prune_pool_onepass_multi(const vqaf_fused_mod* mods_host, int n_mod, int B, int D, int H,
                                        int dtype, uint64_t* sel_rng_state, int rng_advance, void* out,
                                        int ld_out, int out_dtype, vqaf_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Dense single-query cross-attention (non-pruning branch): the attention OUTPUT of
 * nn.MultiheadAttention(query=q[:,None], key=seq, value=seq), utils/fusion.py:257-268 and
 * :173-179.  With one query per sample both projections fold out of the token loop:
 *   s[b,n,h] = x_bn . u[b,h,:]   (vqaf_head_scores),   p[b,h,n] = softmax_n s[b,n,h],
 *   z[b,h,:] = sum_n p[b,h,n] x_bn,   attn_h = Wv_h z_h + bv_h (sum_n p = 1),  out = out_proj(attn)
 * and the backward re-uses the same two token passes plus one rank-2H update:
 *   dp[b,n,h] = x_bn . t[b,h,:] (t_h = Wv_h^T d_attn_h),  ds = p * (dp - sum_n p dp),
 *   du[b,h,:] = sum_n ds[b,h,n] x_bn,   grad_seq[b,n,:] = sum_h ds[b,h,n] u_h + p[b,h,n] t_h.
 * The (B,H,D)-sized projections around these kernels are small GEMMs (vqaf_gemm_f32 or any BLAS).
 * ------------------------------------------------------------------------------------ */

/* p (B,H,N) f32 <- s (B,N,H) f32: softmax over n for every head. */
int vqaf_softmax_heads(const float* s, int B, int N, int H, float* p, vqaf_stream_t stream);

/* ds (B,H,N) <- p (B,H,N), dp (B,N,H): softmax backward per head. */
int vqaf_softmax_heads_bwd(const float* p, const float* dp, int B, int N, int H, float* ds,
                           vqaf_stream_t stream);

/* z (B,H,D) f32 <- seq (B,N,D) `dtype`, c (B,H,N) f32:  z[b,h,:] = sum_n c[b,h,n] seq[b,n,:].
 * Fixed summation order.  H <= 16 and H*N*4 + 8 KiB <= 64 KiB of LDS. */
int vqaf_weighted_token_sum(const void* seq, const float* c, int B, int N, int D, int H, int dtype,
                            float* z, vqaf_stream_t stream);

/* out (B,N,D) `dtype` <- out[b,n,:] = sum_h ca[b,h,n] va[b,h,:] + cb[b,h,n] vb[b,h,:]
 * (ca, cb (B,H,N) f32; va, vb (B,H,D) f32).  H <= 8 and 2*H*N*4 <= 64 KiB. */
int vqaf_token_outer(const float* ca, const float* va, const float* cb, const float* vb, int B,
                     int N, int D, int H, int dtype, void* out, vqaf_stream_t stream);

/* Small strided-batched f32 GEMM used for the per-head projections around the kernels
 * above:  C[g][m][n] = alpha * sum_k A[g][m][k] * Bm[g][k][n] (+ bias[g][n]) (+ C if beta_one)
 * with element strides (s_g, s_m/s_k, s_k/s_n).  A and Bm may be f32 or bf16; C is f32. */
int vqaf_gemm_f32(const void* A, int a_dtype, long a_sg, long a_sm, long a_sk, const void* Bm,
                  int b_dtype, long b_sg, long b_sk, long b_sn, const void* bias, int bias_dtype,
                  long bias_sg, float* C, long c_sg, long c_sm, int G, int M, int N, int K,
                  float alpha, int beta_one, vqaf_stream_t stream);

/* The same product, free to split a long contraction over workgroups that add their parts into C with
 * f32 atomics when the output has only a few tiles (M, N small next to K: the (B,E)-sized projections
 * of the dense branch and their weight gradients at small batches).  The sum then depends on the order
 * the parts arrive in: equal to vqaf_gemm_f32 within f32 rounding, not bit for bit, so it serves
 * gradients and activations, never the scoring path whose results rank tokens. */
int vqaf_gemm_f32_atomic(const void* A, int a_dtype, long a_sg, long a_sm, long a_sk, const void* Bm,
                  int b_dtype, long b_sg, long b_sk, long b_sn, const void* bias, int bias_dtype,
                  long bias_sg, float* C, long c_sg, long c_sm, int G, int M, int N, int K,
                  float alpha, int beta_one, vqaf_stream_t stream);

/* vqaf_gemm_f32_atomic for n_sets <= 4 problems of ONE shape, stride set and dtype set in one launch (HOST
 * arrays of n_sets device pointers; bias_host may be NULL or hold NULLs): the two attention branches of
 * CoAttention (utils/fusion.py:257-268) have equal-shaped projections, which then share their launches. */
int vqaf_gemm_f32_atomic_sets(int n_sets, const void* const* A_host, int a_dtype, long a_sg, long a_sm,
                              long a_sk, const void* const* Bm_host, int b_dtype, long b_sg, long b_sk,
                              long b_sn, const void* const* bias_host, int bias_dtype, long bias_sg,
                              float* const* C_host, long c_sg, long c_sm, int G, int M, int N, int K,
                              float alpha, int beta_one, vqaf_stream_t stream);

/* n <= 12 DIFFERENT strided-batched products (each as vqaf_gemm_f32_atomic describes it: own shape, strides,
 * operand types) in ONE launch, each ADDED into its C, which the caller provides zero-filled (long
 * contractions are split over workgroups with f32 atomics; the sum order is not fixed).  For groups of short
 * products that do not depend on each other -- in the backward of the attention modules
 * (utils/fusion.py:194-200, 257-268 under autograd) the input gradient through the value projection and two
 * weight gradients all follow from one incoming gradient -- and would otherwise each occupy the stream for
 * a launch of their own.  With store = 1 it is also the f32 tier's clipped sums of the DP step (Opacus
 * clip_and_accumulate behind utils/src.py:104): S_W = G^T A and S_b = 1^T G of every Linear in one launch. */
typedef struct vqaf_gemm_desc {
  const void* A;    /* element (g, m, k) at A + g*a_sg + m*a_sm + k*a_sk (elements of a_dtype) */
  const void* Bm;   /* element (g, k, n) at Bm + g*b_sg + k*b_sk + n*b_sn */
  const void* bias; /* element (g, n) at bias + g*bias_sg + n, or NULL */
  void* C;          /* element (g, m, n) at C + g*c_sg + m*c_sm + n (elements of c_dtype) */
  long a_sg, a_sm, a_sk, b_sg, b_sk, b_sn, bias_sg, c_sg, c_sm;
  int a_dtype, b_dtype, bias_dtype;
  int G, M, N, K;
  float alpha;
  int store;        /* 0: C += result (C zero-filled by the caller; long contractions may be split, f32 atomics);
                       1: C = result, plain stores, the contraction is never split (deterministic) */
  int c_dtype;      /* VQAF_F32; VQAF_BF16 is allowed with store = 1: the product leaves in the type its consumer
                       wants (parameter gradients of a bf16 module) and no cast launch follows it */
} vqaf_gemm_desc;
int vqaf_gemm_f32_group(int n, const vqaf_gemm_desc* probs_host, vqaf_stream_t stream);

/* Both branches of the dense co-attention (utils/fusion.py:257-268) in the same launches, with the
 * per-sample softmax steps fused into the token passes; n <= 4 problems of one batch size, width and head
 * count, each with its own token count and buffers.  Results are bit-identical to the separate launches
 * (vqaf_softmax_heads + vqaf_weighted_token_sum; vqaf_softmax_heads_bwd + vqaf_weighted_token_sum +
 * vqaf_token_outer): same arithmetic, same order of every sum.
 *   vqaf_softmax_wsum_multi:  p[b,h,n] = softmax_n s[b,n,h] (written out for the backward),
 *                             z[b,h,:] = sum_n p[b,h,n] x_bn
 *   vqaf_token_bwd_multi:     ds = p * (dp - sum_n p dp) (kept on chip),  du[b,h,:] = sum_n ds[b,h,n] x_bn,
 *                             grad_seq[b,n,:] = sum_h ds[b,h,n] u[b,h,:] + p[b,h,n] t[b,h,:]
 * vqaf_softmax_wsum_multi also clears zero_count f32 at zero_ptr on the side (may be 0 / NULL; nothing the launch
 * reads or writes): the outputs that the projections behind it build with atomic adds.
 * du or grad_seq may be NULL (for every problem alike) when that gradient is not needed.  H <= 8,
 * D % 8 == 0, and the per-sample coefficients must fit 128 KiB of LDS ((H+4)*N*4, resp. (2*H+4)*N*4 bytes, H rounded
 * up to a power of two). */
typedef struct vqaf_dense_fwd_mod {
  const void* seq; /* (B,N,D) `dtype` */
  const float* s;  /* (B,N,H) raw head scores */
  float* p;        /* (B,H,N) out */
  float* z;        /* (B,H,D) out */
  int N;
} vqaf_dense_fwd_mod;
int vqaf_softmax_wsum_multi(const vqaf_dense_fwd_mod* mods_host, int n, int B, int D, int H, int dtype,
                            float* zero_ptr, size_t zero_count, vqaf_stream_t stream);
typedef struct vqaf_dense_bwd_mod {
  const void* seq; /* (B,N,D) `dtype` */
  const float* p;  /* (B,H,N) */
  const float* dp; /* (B,N,H) = seq . t */
  const float* u;  /* (B,H,D) folded queries */
  const float* t;  /* (B,H,D) = Wv_h^T d_attn_h */
  float* du;       /* (B,H,D) out or NULL */
  void* grad_seq;  /* (B,N,D) `dtype` out or NULL */
  int N;
} vqaf_dense_bwd_mod;
int vqaf_token_bwd_multi(const vqaf_dense_bwd_mod* mods_host, int n, int B, int D, int H, int dtype,
                         vqaf_stream_t stream);

/* bf16 GEMM on the matrix cores for the classifier's three uses of a Linear under DP-SGD
 * (model/model.py:54-63 wrapped at train.py:124-141):
 *   C[m][n] = sum_k A(m,k) B(k,n) (+ bias[n]),   f32 accumulation,
 *   A(m,k) = a_kmajor ? A[k*lda + m] : A[m*lda + k],   B(k,n) = b_kmajor ? B[k*ldb + n] : B[n*ldb + k]
 *   forward  y = x W^T + b : A = x,  B = W (out,in) k-contiguous, bias, C bf16
 *   dgrad    gx = gy W     : A = gy, B = W (out,in) k-major, C bf16
 *   clipped sum S = G^T A  : A = G (batch,out) k-major, B = A (batch,in) k-major, C f32, colsum =
 *                            sum_b G[b][m] (the bias gradient); rows of G already carry the clip factor.
 * A, B bf16; C f32 or bf16 (c_dtype); bias f32 or bf16 (bias_dtype) or NULL; colsum is f32 with either
 * C; accumulate != 0 adds into an f32 C (and colsum).  Few output tiles with an f32 C split the contraction over workgroups
 * and add partial tiles with f32 atomics (summation order then varies from run to run).
 * vqaf_gemm_bf16_supported tells whether the layout is covered: row pitches multiples of 8
 * elements; a k-contiguous operand needs its row pitch to cover K rounded up to 8, and when K is not
 * a multiple of 8 its elements between K and that bound must be ZERO and the other operand must be
 * k-major (its rows past K are never read); a k-major operand needs its row pitch to cover the row
 * count rounded up to 8; otherwise VQAF_E_ALIGN (use another GEMM). */
int vqaf_gemm_bf16_supported(long lda, int a_kmajor, long ldb, int b_kmajor, int M, int N, int K);
int vqaf_gemm_bf16(const void* A, long lda, int a_kmajor, const void* B, long ldb, int b_kmajor, int M, int N,
                   int K, const void* bias, int bias_dtype, void* C, long ldc, int c_dtype, int accumulate,
                   float* colsum, vqaf_stream_t stream);

/* The clipped sums of n <= 4 Linear layers AND the DP step's update of their parameters in ONE launch
 * (Opacus clip_and_accumulate + add_noise + scale_grad + the wrapped AdamW's step, train.py:130-153,
 * utils/src.py:104): each 64 x 64 tile of S_W = G^T A (G (B,Out), A (B,In) bf16, rows already scaled by the
 * clip factor) goes through noise, scaling and AdamW in registers and is never written; the tiles of the
 * first column also update the bias from the row sums of G.  param / exp_avg / exp_avg_sq / grad_out (f32)
 * and shadow_bf16 are the FLAT buckets of vqaf_dp_adamw; w_off[i] / b_off[i] are the element offsets of
 * layer i's weight (Out_i x In_i, dense) and bias (b_off < 0: none) in them.  Element e of the buckets draws
 * value (e - noise_lo) & 3 of generator call (e - noise_lo) >> 2 of rng_state {seed, counter}: the values
 * one vqaf_dp_adamw launch over [noise_lo, ...) would draw, so updating some layers here and the rest with
 * vqaf_dp_adamw_ranges equals that launch bit for bit.  `step` {unused, updates applied} is read, not
 * advanced.  VQAF_E_SHAPE when the launch would have fewer than 192 tiles or a product wants 128-row tiles
 * (use vqaf_gemm_bf16_wgrad_group + vqaf_dp_adamw); In_i a multiple of 8, offsets multiples of 4. */
int vqaf_clipped_sum_adamw_group(int n, const void* const* G, const long* ldg, const void* const* A,
                                 const long* lda, const int* Out, const int* In, int B, const long* w_off,
                                 const long* b_off, long noise_lo, const uint64_t* rng_state, float noise_std,
                                 float scale, float* param, float* exp_avg, float* exp_avg_sq,
                                 void* shadow_bf16, float* grad_out, float lr, float beta1, float beta2,
                                 float eps, float weight_decay, const uint64_t* step, vqaf_stream_t stream);

/* vqaf_dp_adamw over n_ranges <= 8 ranges [offset, offset + count) of the flat buckets in one launch, S
 * (the clipped sums) addressed by the same offsets, noise addressed as described above (offsets are
 * multiples of 4 past noise_lo).  The parameters that vqaf_clipped_sum_adamw_group does not reach
 * (normalisation layers) take this call. */
int vqaf_dp_adamw_ranges(int n_ranges, const long* offsets_host, const size_t* counts_host, const float* S,
                         const uint64_t* rng_state, long noise_lo, float noise_std, float scale, float* param,
                         float* exp_avg, float* exp_avg_sq, void* shadow_bf16, float* grad_out, float lr,
                         float beta1, float beta2, float eps, float weight_decay, const uint64_t* step,
                         vqaf_stream_t stream);

/* Linear -> GroupNorm (-> ReLU -> Dropout) of the DP classifier (model/model.py:55-58 after
 * ModuleValidator.fix, train.py:124) in ONE launch: the normalisation runs on the Linear's output tile
 * before it leaves the workgroup.  x (B,K) bf16 rows ldx apart, W (C,K) bf16 rows ldw apart, bias (C) or
 * NULL; gamma, beta (C) `param_dtype`; outputs as vqaf_groupnorm_fwd writes them: y (B,C) bf16 (the
 * activation), xhat (B,C) bf16, rstd (B,groups) f32; the dropout arguments have that call's meaning.  The
 * Linear output is rounded to bf16 before it is normalised and every sum is taken in vqaf_groupnorm_fwd's
 * order, so the results equal vqaf_gemm_bf16 followed by vqaf_groupnorm_fwd bit for bit.  Covered
 * (vqaf_linear_groupnorm_supported): C a multiple of 64, group size 8, 16, 32 or 64, and the operand rules
 * of vqaf_gemm_bf16 for two k-contiguous operands; otherwise VQAF_E_ALIGN (use the two calls). */
int vqaf_linear_groupnorm_supported(long ldx, long ldw, int B, int C, int K, int groups);
int vqaf_linear_groupnorm_fwd(const void* x, long ldx, const void* W, long ldw, const void* bias,
                              int bias_dtype, int B, int C, int K, const void* gamma, const void* beta,
                              int param_dtype, int groups, float eps, int relu, float dropout_p,
                              uint64_t dropout_seed, const uint64_t* dropout_counter, void* y, void* xhat,
                              float* rstd, vqaf_stream_t stream);

/* The backward of the same pair seen from the NEXT Linear: its input gradient dL/dy = gy W (gy (B,K) bf16
 * rows ldg apart, K need not be a multiple of 8 when the rows are zero-padded to one; W (K,C) bf16 rows ldw
 * apart) and, on the output tile, the GroupNorm backward of vqaf_groupnorm_bwd with that call's arguments
 * (y, xhat, rstd, gamma, relu, out_scale) and outputs (g_pre, grad_x), in ONE launch; dL/dy itself is not
 * stored.  Equal to vqaf_gemm_bf16 followed by vqaf_groupnorm_bwd bit for bit.  Covered
 * (vqaf_linear_dgrad_groupnorm_supported): C a multiple of 64, group size 8, 16, 32 or 64, operand rules of
 * vqaf_gemm_bf16 (k-contiguous gy, k-major W); otherwise VQAF_E_ALIGN. */
int vqaf_linear_dgrad_groupnorm_supported(long ldg, long ldw, int B, int C, int K, int groups);
int vqaf_linear_dgrad_groupnorm_bwd(const void* gy, long ldg, const void* W, long ldw, int B, int C, int K,
                                    const void* y, const void* xhat, const float* rstd, const void* gamma,
                                    int param_dtype, int groups, int relu, float out_scale, void* g_pre,
                                    void* grad_x, vqaf_stream_t stream);

/* n <= 4 products of the clipped-sum shape in ONE launch: C_i (M_i,N_i) f32 = A_i^T B_i with A_i (K,M_i) and
 * B_i (K,N_i) bf16, both k-major, the same K (the batch); colsum (array of n pointers, or NULL, entries may be
 * NULL) receives the column sums of A_i; accumulate != 0 adds into C_i / colsum_i.  The clipped sums of the
 * classifier's Linear layers are launch-bound on their own.  Returns VQAF_E_SHAPE when the launch would have
 * fewer than 192 output tiles altogether (such products want their contraction split) or the products need
 * different tile heights: the caller then uses vqaf_gemm_bf16 per product. */
int vqaf_gemm_bf16_wgrad_group(int n, const void* const* A, const long* lda, const void* const* B,
                               const long* ldb, const int* M, const int* N, int K, float* const* C,
                               const long* ldc, int accumulate, float* const* colsum, vqaf_stream_t stream);

/* ------------------------------------------------------------------------------------
 * DP-SGD step for the classifier.  The reference delegates this to Opacus 1.4.0
 * (GradSampleModule hooks + DPOptimizer.clip_and_accumulate / add_noise / scale_grad),
 * reached from loss.backward() and optimizer['classifier'].step() at utils/src.py:99,104
 * and configured at train.py:124-153.  Per-sample gradients are never materialised:
 * for a Linear on 2-D input g_i^W = b_i a_i^T, so ||g_i^W||^2 = ||a_i||^2 ||b_i||^2.
 * ------------------------------------------------------------------------------------ */

/* sq[b] (+)= ||a_b||^2 * ||g_b||^2 + (has_bias ? ||g_b||^2 : 0)
 * a (B,In), g (B,Out) of `dtype`; sq (B) f32; accumulate != 0 adds to sq. */
int vqaf_ghost_sqnorm_linear(const void* a, const void* g, int B, int In, int Out, int dtype,
                             int has_bias, int accumulate, float* sq, vqaf_stream_t stream);

/* Per-sample squared gradient norm of an affine normalisation layer (GroupNorm weight and
 * bias): sq[b] (+)= sum_c (xhat[b,c] g[b,c])^2 + sum_c g[b,c]^2.  xhat, g (B,C). */
int vqaf_ghost_sqnorm_affine(const void* xhat, const void* g, int B, int C, int dtype,
                             int accumulate, float* sq, vqaf_stream_t stream);

/* coef[b] = scale * min(1, max_norm / (scale*sqrt(sq[b]) + 1e-6));  norms[b] = scale*sqrt(sq[b])
 * (norms may be NULL).  `scale` undoes the 1/B of a mean-reduced loss, so coef multiplies
 * the raw backprops.  Opacus DPOptimizer.clip_and_accumulate. */
int vqaf_clip_coef(const float* sq, int B, float max_norm, float scale, float* coef,
                   float* norms, vqaf_stream_t stream);

/* The ghost norms of every wrapped layer and the clip coefficient in ONE launch (one workgroup
 * per sample): same arithmetic as vqaf_ghost_sqnorm_linear / _affine accumulated over the layers
 * followed by vqaf_clip_coef.  The *_host arguments are HOST arrays of n_layers entries (device
 * pointers a[l], g[l]; in_dim / out_dim; kind 0 = Linear, 1 = Linear with bias, 2 = affine
 * normalisation with a = xhat, in_dim ignored).  n_layers <= 8.  norms, sq (B) may be NULL.
 * With scale_g_inplace != 0 every g[l] row is then multiplied by its sample's coef IN PLACE
 * (g is not const in that mode); the clipped-sum entry points are then called with coef == NULL. */
int vqaf_ghost_clip(int n_layers, const void* const* a_host, const void* const* g_host,
                    const int* in_dim_host, const int* out_dim_host, const int* kind_host, int B,
                    int dtype, float max_norm, float scale, int scale_g_inplace, float* coef,
                    float* norms, float* sq, vqaf_stream_t stream);

/* vqaf_ghost_clip that also clears n_zero <= 8 ranges of f32 (HOST arrays of device pointers and element
 * counts) in the same launch: the sums the following kernels build with atomic adds (what a separate
 * vqaf_zero_multi launch would do).  The ranges must not overlap anything the launch reads. */
int vqaf_ghost_clip_zero(int n_layers, const void* const* a_host, const void* const* g_host,
                         const int* in_dim_host, const int* out_dim_host, const int* kind_host, int B,
                         int dtype, float max_norm, float scale, int scale_g_inplace, float* coef,
                         float* norms, float* sq, float* const* zero_ptrs_host,
                         const size_t* zero_counts_host, int n_zero, vqaf_stream_t stream);

/* vqaf_ghost_clip_zero that also adds 1 to count_state[1] ({unused, count} uint64 pair in device memory, or
 * NULL): the update count of the step that follows (vqaf_dp_adamw with step_advance = -1), so that no launch
 * of its own is spent on it.  Nothing this launch reads may alias count_state.  The DP optimizer's per-step
 * bookkeeping around train.py:130-141 / utils/src.py:104. */
int vqaf_ghost_clip_zero_bump(int n_layers, const void* const* a_host, const void* const* g_host,
                              const int* in_dim_host, const int* out_dim_host, const int* kind_host, int B,
                              int dtype, float max_norm, float scale, int scale_g_inplace, float* coef,
                              float* norms, float* sq, float* const* zero_ptrs_host,
                              const size_t* zero_counts_host, int n_zero, uint64_t* count_state,
                              vqaf_stream_t stream);

/* Clipped gradient sum of a Linear:  SW (Out,In) f32 = sum_b coef[b] g_b a_b^T,
 * Sb (Out) f32 = sum_b coef[b] g_b (Sb may be NULL).  bf16 inputs use bf16 MFMA, f32
 * inputs the exact f32 MFMA.  accumulate != 0 adds into SW / Sb (micro-batching).
 * coef == NULL means g is already scaled (vqaf_ghost_clip with scale_g_inplace). */
int vqaf_clipped_sum_linear(const void* a, const void* g, const float* coef, int B, int In,
                            int Out, int dtype, int accumulate, float* SW, float* Sb,
                            vqaf_stream_t stream);

/* Clipped gradient sums of an affine normalisation layer:
 * Sgamma[c] = sum_b coef[b] xhat[b,c] g[b,c];  Sbeta[c] = sum_b coef[b] g[b,c]  (coef == NULL: 1). */
int vqaf_clipped_sum_affine(const void* xhat, const void* g, const float* coef, int B, int C,
                            int dtype, int accumulate, float* Sgamma, float* Sbeta,
                            vqaf_stream_t stream);

/* The same sums for n <= 8 affine layers in one launch (the GroupNorm layers of the classifier): arrays
 * of n pointers / widths; coef (B) f32 or NULL is shared; the sums are ACCUMULATED into Sgamma_i /
 * Sbeta_i (atomics), which the caller zeroes (vqaf_zero_multi). */
int vqaf_clipped_sum_affine_multi(int n, const void* const* xhat, const void* const* g, const float* coef,
                                  int B, const int* C, int dtype, float* const* Sgamma,
                                  float* const* Sbeta, vqaf_stream_t stream);

/* grad[i] = (S[i] + noise_std * z_i) * scale  over n f32 elements
 * (Opacus DPOptimizer.add_noise + scale_grad; noise_std = noise_multiplier * max_grad_norm,
 * scale = 1/expected_batch_size).  z = noise[i] if `noise` is non-NULL, else N(0,1) drawn in
 * the kernel from rng_state {seed, counter}; element i always uses draw (counter, i), so the
 * result does not depend on the launch geometry.  grad may alias S. */
int vqaf_noise_scale(const float* S, const float* noise, const uint64_t* rng_state,
                     float noise_std, float scale, size_t n, float* grad, vqaf_stream_t stream);

/* vqaf_noise_scale followed by one AdamW update (torch.optim.AdamW semantics: decoupled weight
 * decay, bias-corrected moments), fused over flat f32 buckets of n elements:
 *   g = (S + noise_std z) * scale;  p *= 1 - lr*wd;  m += (1-b1)(g - m);  v = b2 v + (1-b2) g^2;
 *   p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps),   t = step[1] + 1.
 * `step` = uint64 words in device memory {unused, updates applied so far}: with step_advance == 0
 * two words, advanced by the caller (vqaf_rng_advance) after the call; with step_advance > 0 FOUR
 * words {unused, updates, ticket, reserved} (ticket zero) and the last workgroup adds step_advance
 * to the count itself; with step_advance == -1 the count was ALREADY advanced by one for this update before
 * the launch (vqaf_ghost_clip_zero_bump): t = step[1], and a rng_state that is the same pair draws from
 * counter step[1] - 1, i.e. exactly what the launch with step_advance == 0 followed by the caller's advance
 * computes.  `step` may double as rng_state.  shadow_bf16 (n bf16, may be NULL) receives the updated
 * parameters rounded to bf16; grad_out (n f32, may be NULL, may alias S) receives g.
 * This is the step train.py:130-141 configures: AdamW on the classifier wrapped by the DP
 * optimizer (utils/src.py:104). */
int vqaf_dp_adamw(const float* S, const float* noise, const uint64_t* rng_state, float noise_std,
                  float scale, float* param, float* exp_avg, float* exp_avg_sq, void* shadow_bf16,
                  float* grad_out, size_t n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, uint64_t* step, int step_advance, vqaf_stream_t stream);

/* Zeroes up to 8 float ranges (pointer, element count) in one launch: the accumulators of the
 * clipped sums that add atomically (then called with accumulate = 1), instead of one memset each. */
int vqaf_zero_multi(float* const* ptrs_host, const size_t* counts_host, int n_ranges, vqaf_stream_t stream);

/* GroupNorm (+ optional ReLU) on (B,C) input, as the classifier holds after
 * ModuleValidator.fix swaps BatchNorm1d for GroupNorm(min(32,C), C) (train.py:124;
 * model/model.py:56-61).  Saves xhat (B,C) for the backward and the ghost norms.
 * dropout_p > 0 (needs relu): the nn.Dropout that follows the ReLU (model/model.py:57-58) applied to
 * y in the same pass -- inverted dropout; the keep decision of element e is drawn from stream
 * (dropout_seed, dropout_counter[1]).  dropout_counter is any {unused, count} pair in device memory
 * that changes between training steps (the step count vqaf_dp_adamw keeps serves: no launch is spent
 * on a counter of its own, and a replayed HIP graph draws fresh masks); dropout_seed tells apart
 * the call sites within one step. */
int vqaf_groupnorm_fwd(const void* x, const void* gamma, const void* beta, int B, int C,
                       int groups, float eps, int relu, int dtype, void* y, void* xhat,
                       float* rstd, float dropout_p, uint64_t dropout_seed, const uint64_t* dropout_counter,
                       vqaf_stream_t stream);

/* Backward of the above w.r.t. x only (parameter gradients come from the clipped sums).
 * y is the forward output (for the ReLU mask); writes g_pre (B,C), the gradient at the
 * GroupNorm output before the affine (what the ghost norms and clipped sums consume),
 * and grad_x (B,C).  out_scale = 1/(1 - dropout_p) when the forward applied dropout (y > 0 is then both
 * the ReLU gate and the keep mask), 1 otherwise. */
int vqaf_groupnorm_bwd(const void* grad_y, const void* y, const void* xhat, const float* rstd,
                       const void* gamma, int B, int C, int groups, int relu, float out_scale, int dtype,
                       void* g_pre, void* grad_x, vqaf_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Membership-inference scoring: per-sample cross-entropy and max-softmax confidence.
 * Replaces nn.CrossEntropyLoss(reduction='none') (attack/metric_src.py:313-314, 335, 354)
 * and softmax -> max (attack/metric_src.py:376-377, 415).  logits (B,C) `dtype`,
 * targets (B) int64; loss, conf (B) f32 and pred (B) int64 may each be NULL.
 * grad_logits (B,C) `grad_dtype`, if non-NULL, receives grad_scale * (softmax - onehot): the
 * backward of the mean-reduced training loss (utils/src.py:96,99) with grad_scale = 1/B; written in
 * the logits' own dtype it IS the gradient loss.backward() hands to the classifier (incoming
 * gradient 1), with no further launch.
 * ------------------------------------------------------------------------------------ */
int vqaf_ce_conf(const void* logits, const int64_t* targets, int B, int C, int dtype,
                 float* loss, float* conf, int64_t* pred, void* grad_logits, int grad_dtype,
                 float grad_scale, vqaf_stream_t stream);

/* The training loss of utils/src.py:96 (nn.CrossEntropyLoss(), mean over the batch) in ONE launch:
 * vqaf_ce_conf's per-row losses `loss` (B) and logits gradient, plus their mean in loss_mean[0].  The
 * workgroup that finishes last adds the B values in a fixed order (the result is the same whichever
 * workgroup that is).  `ticket` is one zero-initialised 32-bit word of device memory owned by the caller;
 * the launch leaves it zero, so it can be reused by the next call on the same stream (calls that overlap
 * on different streams need a word each).  A target outside [0, C) makes the mean NaN.  grad_logits rows
 * are ld_grad >= C elements apart and the launch writes zeros to the elements between C and ld_grad: with
 * ld_grad = C rounded up to 8 the gradient of a 430-class head is a legal operand of vqaf_gemm_bf16 and of
 * vqaf_gemm_bf16_wgrad_group. */
int vqaf_ce_mean(const void* logits, const int64_t* targets, int B, int C, int dtype, float* loss,
                 void* grad_logits, long ld_grad, int grad_dtype, float grad_scale, float* loss_mean,
                 unsigned int* ticket, vqaf_stream_t stream);

/* out[i] = in[i] * scale[0] (scale in device memory), written as `out_dtype`: the backward of the
 * mean cross-entropy of utils/src.py:96 -- the logits gradient vqaf_ce_conf produced in the forward
 * pass times the incoming scalar gradient, rounded once into the logits' dtype. */
int vqaf_scale_cast(const float* in, const float* scale, size_t n, int out_dtype, void* out,
                    vqaf_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Encoder blocks: the transformer layers of the vision and text encoders whose outputs feed the
 * fusion (model/vision_encoder.py:83-127 and model/text_encoder.py:14-48 wrap Hugging Face
 * models; these are the per-layer operations of those models that are not plain GEMMs).
 * ------------------------------------------------------------------------------------ */

/* Multi-head self-attention, bf16, head size dh = 64:
 *   out[b,n,64h:64h+64] = softmax_m(scale * q_bnh . k_bmh + key_bias[b,m]) v_bmh.
 * q, k, v: (B, N, ld_qkv) bf16 with head h in columns [64h, 64h+64) (three tensors, or three column
 * blocks of one packed projection output); key_bias (B,N) f32 additive or NULL (the extended
 * attention mask of BertModel: 0 for tokens, a large negative number for padding); out (B, N, ld_o);
 * lse (B,H,N) f32 receives the base-2 logarithm of the softmax denominator in the scaled domain
 * (what vqaf_attn_bwd needs).  Row pitches are in elements and multiples of 8.
 * dropout_p > 0 drops attention probabilities after the softmax (attention_probs_dropout_prob of the
 * BERT config, active in train mode): probability (b,h,n,m) is kept, and scaled by 1/(1-p), iff word
 * m%4 of the Philox4x32-10 call (seed, counter, ((b*H+h)*N+n)*ceil(N/4) + m/4) is >= p*2^32, with
 * {seed, counter} = dropout_state[0..1] in device memory (read at run time; not advanced). */
int vqaf_attn_fwd(const void* q, const void* k, const void* v, long ld_qkv, const float* key_bias,
                  int B, int H, int N, int dh, float scale, float dropout_p,
                  const uint64_t* dropout_state, void* out, long ld_o, float* lse, vqaf_stream_t stream);

/* Backward of vqaf_attn_fwd: dq, dk, dv (B, N, ld_d) bf16 from dout (B, N, ld_o) and the forward's
 * lse; the forward output itself is not needed (the row term of the softmax backward is recomputed
 * from the probabilities, which keeps sum_m ds[n,m] = 0 to f32 rounding).  dvec (B,H,N) f32 is
 * workspace.  dropout_p / dropout_state: the values the forward call saw (a copy of its state). */
int vqaf_attn_bwd(const void* q, const void* k, const void* v, long ld_qkv, const float* key_bias,
                  const void* dout, long ld_o, const float* lse, int B, int H, int N,
                  int dh, float scale, float dropout_p, const uint64_t* dropout_state, void* dq, void* dk,
                  void* dv, long ld_d, float* dvec, vqaf_stream_t stream);

/* LayerNorm over the last dimension with the residual add in front of it (nn.LayerNorm of the
 * BertSelfOutput / BertOutput / ViTLayer blocks): s = x (+ residual), y = LN(s) * gamma + beta.
 * x, residual (may be NULL), sum_out (may be NULL; receives s), y: (rows, D) `dtype`; gamma, beta
 * (D) `param_dtype`; mean, rstd (rows) f32 for the backward.  D <= 2048, D % 8 == 0. */
int vqaf_layernorm_fwd(const void* x, const void* residual, const void* gamma, const void* beta,
                       int param_dtype, long rows, int D, float eps, int dtype, void* sum_out, void* y,
                       float* mean, float* rstd, vqaf_stream_t stream);

/* Backward: dx (rows, D) = LayerNorm backward of dy at input s (+ dsum_in, the gradient that reaches
 * s directly through a residual branch; may be NULL); dgamma, dbeta (D) `param_dtype` are written
 * (fixed summation order).  workspace: vqaf_layernorm_bwd_workspace_bytes(rows, D) bytes of device
 * memory, 16-byte aligned (per-workgroup column sums). */
size_t vqaf_layernorm_bwd_workspace_bytes(long rows, int D);
int vqaf_layernorm_bwd(const void* dy, const void* s, const void* gamma, int param_dtype,
                       const float* mean, const float* rstd, const void* dsum_in, long rows, int D,
                       int dtype, void* dx, void* dgamma, void* dbeta, void* workspace,
                       size_t workspace_bytes, vqaf_stream_t stream);

/* GELU in its exact (erf) form, the "gelu" activation of both encoder configs; n % 8 == 0. */
int vqaf_gelu_fwd(const void* x, size_t n, int dtype, void* y, vqaf_stream_t stream);
int vqaf_gelu_bwd(const void* x, const void* dy, size_t n, int dtype, void* dx, vqaf_stream_t stream);

/* Embedding lookups of the text encoder (model/text_encoder.py:14-48: BertModel / RobertaModel add a word, a
 * position and a token-type row per token).  Forward: out (T,E) `out_dtype` = table0[ids0] + table1[ids1] +
 * table2[ids2] summed in f32 and rounded once; tables (rows_k, E) `param_dtype`, ids int64 (T); the second and
 * third table may be NULL.  An index outside its table reads the nearest row.  E % 8 == 0.
 * Backward: table_grad (rows, E) `param_dtype` = for every row r the sum of g (T,E) `g_dtype` over the tokens
 * with ids == r, added in token order (reproducible); rows no token names, and row pad_idx (-1: none), are
 * zero.  T <= 32768. */
int vqaf_embed_fwd(const long* ids0, const void* table0, long rows0, const long* ids1, const void* table1,
                   long rows1, const long* ids2, const void* table2, long rows2, int T, int E, int param_dtype,
                   void* out, int out_dtype, vqaf_stream_t stream);
int vqaf_embed_bwd(const long* ids, int T, const void* g, int g_dtype, int E, long rows, long pad_idx,
                   void* table_grad, int param_dtype, vqaf_stream_t stream);

/* Encoder-size Linear on the library's own large-tile bf16 MFMA kernel (model/vision_encoder.py:83-127 and
 * model/text_encoder.py:14-48 run these layers through torch.nn.Linear inside the pretrained stacks):
 * y (M,N) bf16 = x (M,K) op(w) (+ bias (N), `bias_dtype`), f32 accumulation.  w is (N,K) with k contiguous
 * (w_kmajor 0: the forward y = x W^T + b) or (K,N) (w_kmajor 1: the input gradient gx = gy W straight from
 * the (out,in) weight, no transposed copy); ldx, ldw are row pitches in elements.  epilogue: 0 = bias only;
 * 1 = also writes aux (M,N) = gelu(y) (y stays the pre-activation the backward needs); 2 = GELU backward,
 * y = round_bf16(x op(w)) * gelu'(aux) with aux (M,N) the forward's pre-activation and bias NULL.
 * Needs K % 64 == 0, N % 8 == 0, pitches that are multiples of 8 and 16-byte aligned pointers:
 * vqaf_linear_nt_supported tells (1/0); other shapes go to vqaf_gemm_bf16. */
int vqaf_linear_nt_supported(long ldx, long ldw, int w_kmajor, long ldy, int M, int N, int K);
int vqaf_linear_nt(const void* x, long ldx, const void* w, long ldw, int w_kmajor, const void* bias,
                   int bias_dtype, int M, int N, int K, void* y, long ldy, int epilogue, void* aux, long ld_aux,
                   vqaf_stream_t stream);
/* Weight and bias gradient of such a layer: gw (M,N) `gw_dtype` = gy^T x with gy (K,M) and x (K,N) bf16, K the
 * token count (the slow index of both: no transposed copies are made); gb (M) `gb_dtype` = column sums of gy,
 * or NULL.  The token range is split over workgroups when the output alone has too few tiles to fill the
 * chip; the f32 partial products go to `workspace` (vqaf_wgrad_tn_workspace_bytes(M,N,K) bytes of device
 * memory, 16-byte aligned) and a second launch adds them in a fixed order, so results are reproducible.
 * Needs M % 8 == 0, N % 8 == 0, pitches that are multiples of 8, 16-byte aligned pointers. */
int vqaf_wgrad_tn_supported(long ldg, long ldx, long ld_out, int M, int N, int K);
size_t vqaf_wgrad_tn_workspace_bytes(int M, int N, int K);
int vqaf_wgrad_tn(const void* gy, long ldg, const void* x, long ldx, int M, int N, int K, void* gw, long ld_out,
                  int gw_dtype, void* gb, int gb_dtype, void* workspace, size_t workspace_bytes,
                  vqaf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VQAF_H_ */
