/* C ABI of libscmomer_hip.so: gfx950 (MI355X) kernels for the scMomer Performer RNA-encoder path.
 *
 * The reference (nobody927/scMomer) is pure Python/PyTorch and has no FFI of its own; each entry
 * point below replaces the aten-op group of the cited reference lines.  A binding only needs
 * device pointers, sizes and a stream (bound with ctypes in scmomer_amd/_lib.py).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless marked "host"; buffers are caller-owned, nothing
 *     is allocated or freed by the library, and nothing synchronises the device;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream);
 *   - activations fed to matrix products are bf16 (raw uint16 storage); the residual stream,
 *     statistics, gradients of parameters and all accumulations are f32;
 *   - every function returns 0 on success, a negative SCM_E* code on a rejected argument, or a
 *     positive hipError_t from the launch; scm_last_error() gives the text (host, thread-local);
 *   - T = B*N tokens, D = model dim (multiple of 8; <= 256 for the row-wise kernels, <= 224 / 220 for the products
 *     with LayerNorm epilogues, 200 for the chained FeedForward; the module API accepts dim <= 200), head dim is 64
 *     and the number of random features is 266 (compile-time constants of the path);
 *   - no tensor size limit other than T < 2^31: the kernels address with 32-bit offsets inside 2 GiB windows, and
 *     the entry points walk tall inputs in row slabs (products) or re-base per token slice / per (cell, head).
 */
#ifndef SCMOMER_HIP_H
#define SCMOMER_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCM_EINVAL (-1)
#define SCM_NB_FEATURES 266
#define SCM_NB_FEATURES_PAD 288
#define SCM_DIM_HEAD 64

/* bf16 tensor view [B,H,N,64]: element (b,h,n,d) at p + b*sb + h*sh + n*sn + d (strides in elements,
 * d contiguous, sb/sh/sn multiples of 8).  Covers both the reference's 'b h n d' layout
 * (performer_pytorch.py:376) and head slices of a fused [B*N, 3*H*64] projection buffer. */
typedef struct { void* p; int64_t sb, sh, sn; } scm_view4;

int scm_version(void);
const char* scm_last_error(void); /* host */
/* Diagnostic: how many products have run on the general register-staged kernel since the library was loaded (shapes
 * that none of the asynchronous-copy kernels takes; the encoder's own shapes never do, at any batch size). */
int64_t scm_gemm_general_launches(void);

/* ---- parameter preparation ------------------------------------------------------------- */
/* f32 weight [rows,cols] -> bf16 copy and/or bf16 transposed copy [cols,rows] (either may be NULL). */
int scm_cast_weight(const float* src, int rows, int cols, void* dst_bf16, void* dstT_bf16, void* stream);
/* The same for a table of n weights in ONE launch (after an optimizer step).  The table lives in DEVICE memory;
 * entry i casts src [R,C] f32 into dst[r * ldd + c] and dstT[c * ldt + r] (bf16; either may be NULL), so that
 * several sources can fill slices of one fused matrix (q/k/v weights -> the fused [3*inner, dim] operand). */
typedef struct {
  const float* src;
  int32_t R, C;
  void* dst;
  int64_t ldd;
  void* dstT;
  int64_t ldt;
  int32_t chain; /* bit 0: dst columns, bit 1: dstT columns are written in the chain order of scm_ffn_chain_fwd */
  int32_t group_cells;               /* > 0: the batch is B / group_cells micro-batches stacked (B a multiple), one set
                                      * of loss means per micro-batch; 0: one group */
} scm_cast_item;
int scm_cast_weights_batch(const scm_cast_item* items /* device */, int n, void* stream);
/* FastAttention.projection_matrix f32 [266,64] (performer_pytorch.py:209) -> zero-padded bf16 [288,64]. */
int scm_prep_omega(const float* omega, void* omega_b, void* stream);

/* ---- embeddings: PerformerLM.forward performer_pytorch.py:604-607, Gene2VecPositionalEmbedding :444-446 -- */
/* out[b,n,:] = f32( f64(E[tok[b,n],:]) + G[n,:] ); G (f64 [>=N, D]) may be NULL (no positional table).
 * *err (device int, caller-zeroed) is set to 1 if a token id is outside [0,V). */
int scm_embed_fwd(const int64_t* tok, const float* E, const double* G, int B, int N, int D, int V, float* out,
                  int* err, void* stream);
/* dE[v,:] (+)= sum of dx rows whose token id is v.  ws: scm_embed_bwd_ws_floats(T,D,V) floats.
 * dE may be NULL: then the per-workgroup partial rows [blocks][V*D] stay in ws for scm_reduce_batch. */
int64_t scm_embed_bwd_ws_floats(int64_t T, int D, int V);
int scm_embed_bwd(const int64_t* tok, const float* dx, int64_t T, int D, int V, float* ws, float* dE, int accumulate,
                  void* stream);

/* ---- LayerNorm of PreLayerNorm: performer_pytorch.py:289-295 (eps 1e-5, affine) ----------- */
int scm_ln_fwd(const float* x, const float* gamma, const float* beta, int64_t T, int D, void* y_bf16, float* mean,
               float* rstd, void* stream);
/* dx = dres + dLN(dy) (dres may be NULL); dgamma/dbeta (+)= ; dres_colsum (+)= column sums of dres
 * (the bias gradient of the Linear that produced the residual branch; may be NULL).
 * dx_bf16 (may be NULL): the same dx rounded to bf16, the operand the next data-/weight-gradient products read.
 * ws: scm_ln_bwd_ws_floats(T,D) floats. */
int64_t scm_ln_bwd_ws_floats(int64_t T, int D);
int scm_ln_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
               const float* dres, int64_t T, int D, float* dx, void* dx_bf16, float* ws, float* dgamma,
               float* dbeta, float* dres_colsum, int accumulate, void* stream);

/* The same LayerNorm backward for callers that cannot allocate or carry a workspace (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  It has no counterpart in the reference.  The column
 * partials live in an array built into the library's code object (3 MiB per GPU, kept until the process ends)
 * instead of `ws`; nothing is allocated or freed, on the device or through any library.  Limits: D <= 256 and
 * D % 4 == 0 as above, any T (the partial count is capped at 1024 workgroups); a size beyond the array is rejected
 * with SCM_EINVAL.  The first call on a device looks the array's address up (make that call outside a graph
 * capture).  All calls on one GPU share the array: calls that overlap on one GPU (different streams) race; calls
 * on one stream are ordered and safe.  The stream must belong to the CURRENT device (each GPU has its own copy of
 * the array at its own address, and the current device's is used): a stream of another device is rejected with
 * SCM_EINVAL.  Results are bit-identical to scm_ln_bwd's. */
int scm_ln_bwd_noalloc(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                       const float* dres, int64_t T, int D, float* dx, void* dx_bf16, float* dgamma, float* dbeta,
                       float* dres_colsum, int accumulate, void* stream);

/* ---- nn.Linear products: performer_pytorch.py:374 (to_q/k/v), :402 (to_out), :323,330 (FeedForward) ---- */
/* Y[M,N] = A[M,K] * W[N,K]^T with a fused epilogue; A is bf16 or (a_f32) f32, W bf16, K % 8 == 0, N % 8 == 0,
 * leading dimensions multiples of 8 elements and 16-byte aligned bases.
 *   SCM_EPI_BF16          out0 bf16 = acc
 *   SCM_EPI_F32_BIAS_RES  out0 f32  = acc + bias[n] + resid[m,n]     (bias/resid may be NULL; reversible.py:144-145)
 *   SCM_EPI_GELU_DUAL     out0 bf16 = acc + bias ; out1 bf16 = gelu_erf(acc + bias)   (nn.GELU, :313)
 *   SCM_EPI_DGELU         out0 bf16 = acc * gelu_erf'(pre[m,n])
 *   SCM_EPI_F32           out0 f32  = acc
 *   SCM_EPI_BF16_BIAS     out0 bf16 = acc + bias[n] */
enum { SCM_EPI_BF16 = 0, SCM_EPI_F32_BIAS_RES = 1, SCM_EPI_GELU_DUAL = 2, SCM_EPI_DGELU = 3, SCM_EPI_F32 = 4,
       SCM_EPI_BF16_BIAS = 5 };
int scm_gemm_nt(int epi, int a_f32, const void* A, int64_t lda, const void* W_bf16, int64_t ldw, int M, int N, int K,
                void* out0, int64_t ldo0, void* out1, int64_t ldo1, const float* bias, const float* resid,
                int64_t ldr, const void* pre_bf16, int64_t ldp, void* stream);
/* ---- the data-gradient product that ends a sublayer's backward, with the LayerNorm backward as its epilogue:
 *      dx = dres + dLN(A W^T) for the PreLayerNorm of performer_pytorch.py:289-295 wrapped around the to_q/to_k/to_v
 *      (:374-376) or FeedForward.w1 (:323) Linear.  It replaces scm_gemm_nt(SCM_EPI_F32) followed by scm_ln_bwd: the
 *      f32 product rows stay on the chip.  A [M,K] bf16, W [D,K] bf16; x, dres, dx: [M,D] f32, contiguous rows;
 *      other arguments as scm_ln_bwd (dgamma, dbeta, dres_colsum may all be NULL: then the column partials are left
 *      in ws, [scm_gemm_ln_bwd_ws_floats(M,D) / (3 D)][3 D], for a later scm_reduce_batch).
 *      Sizes: scm_gemm_ln_bwd_supported (D <= 224, D % 4 == 0, K >= 128, K % 8 == 0, lda <= 2340, any M: inputs taller
 *      than 458,752 rows run as row slabs of that height); other sizes are rejected with SCM_EINVAL and take the
 *      two separate calls. */
int scm_gemm_ln_bwd_supported(int64_t M, int D, int K, int64_t lda, int64_t ldw);
int64_t scm_gemm_ln_bwd_ws_floats(int64_t M, int D);
int scm_gemm_ln_bwd(const void* A_bf16, int64_t lda, const void* W_bf16, int64_t ldw, int64_t M, int D, int K,
                    const float* x, const float* mean, const float* rstd, const float* gamma, const float* dres,
                    float* dx, void* dx_bf16, float* ws, float* dgamma, float* dbeta, float* dres_colsum,
                    int accumulate, void* stream);

/* ---- the output product of a sublayer with the NEXT sublayer's LayerNorm as its epilogue:
 *      y = A W^T + bias + resid (SelfAttention.to_out, performer_pytorch.py:402, or FeedForward.w2, :330, with the
 *      residual add of reversible.py:144-145), then h = LN(y) for the PreLayerNorm (:289-295) that reads y next.
 *      It replaces scm_gemm_nt(SCM_EPI_F32_BIAS_RES) followed by scm_ln_fwd: y is not read back.
 *      A [M,K] bf16, W [D,K] bf16; resid (may be NULL), y: [M,D] f32, contiguous rows; h [M,D] bf16; mean, rstd [M].
 *      Sizes: scm_gemm_res_ln_supported (D <= 220, D % 4 == 0, K >= 128, K % 8 == 0, lda <= 2340, any M). */
int scm_gemm_res_ln_supported(int64_t M, int D, int K, int64_t lda, int64_t ldw);
int scm_gemm_res_ln(const void* A_bf16, int64_t lda, const void* W_bf16, int64_t ldw, int64_t M, int D, int K,
                    const float* bias, const float* resid, float* y, const float* gamma, const float* beta,
                    void* h_bf16, float* mean, float* rstd, void* stream);

/* ---- forward of the FeedForward sublayer in one launch: PreLayerNorm(Chunk(FeedForward)),
 *      performer_pytorch.py:289-331, with the residual add of reversible.py:145.  Model dim D = 200 and a hidden
 *      width F % 8 == 0, F <= 2048 (scm_ffn_chain_supported); other sizes use scm_ln_fwd + scm_gemm_nt.
 *   y = x + w2 gelu(w1 LN(x) + b1) + b2; optionally h_next = LN_next(y) for the following sublayer.
 * The hidden activations stay on the chip between the two products.  The second weight is read in "chain order":
 * inside every block of 64 along its F index, unit 32 s + 16 e + 8 h + 4 t + i sits at position
 * 32 t + 16 s + 8 h + 4 e + i (bits s,e,h,t; i = 0..3) -- the order in which the first product's accumulators become
 * the second product's operand.  scm_cast_weights_batch writes such copies (scm_cast_item.chain).
 * Written when the pointers are given: h = LN(x) (bf16), mean, rstd, pre, act (the tape of a backward that runs on
 * scm_gemm_nt / scm_wgrad / scm_ln_bwd).  The product path uses this entry point where no tape is kept. */
typedef struct {
  int64_t M; int32_t D, F;
  const float* x; int64_t ldx;                 /* f32 [M, D] */
  const float* gamma; const float* beta;       /* LayerNorm of the sublayer */
  const void* w1; int64_t ldw1;                /* bf16 [F, D] */
  const float* b1;                             /* [F], may be NULL */
  const void* w2c; int64_t ldw2c;              /* bf16 [D, Fp], Fp = F rounded up to 64, chain order along F, zeros at unused positions */
  const float* b2;                             /* [D], may be NULL */
  void* h; int64_t ldh; float* mean; float* rstd;          /* bf16 [M, D], f32 [M] x 2; all NULL when nothing is saved */
  void* pre; int64_t ldpre; void* act; int64_t ldact;      /* bf16 [M, F]; may be NULL */
  float* y; int64_t ldy;                       /* f32 [M, D] */
  const float* gamma_next; const float* beta_next;         /* optional LayerNorm of y: all five NULL to skip */
  void* h_next; int64_t ldh_next; float* mean_next; float* rstd_next;
} scm_ffn_fwd_args;
int scm_ffn_chain_supported(int64_t M, int D, int F);
int scm_ffn_chain_fwd(const scm_ffn_fwd_args* args /* host */, void* stream);

/* Weight gradient: G[P,Q] = sum_t Pm[t,P] * Qm[t,Q]; Pm bf16 (P % 8 == 0), Qm bf16 or (q_f32) f32, Q <= 200,
 * Q % 8 == 0.  grad (+)= G, or G^T when trans (grad is [Q,P]).  bias_grad (may be NULL) (+)= column sums
 * of Pm.  ws: scm_wgrad_ws_floats(P, slices) floats; slices token slices are reduced in a fixed order. */
int64_t scm_wgrad_ws_floats(int P, int slices);
/* grad may be NULL: then only the slice partials are left in ws ([slices][P][208] f32, the bias column at index Q
 * when bias_grad is not NULL) for a later scm_reduce_batch. */
int scm_wgrad(const void* Pm_bf16, int64_t ldp, const void* Qm, int64_t ldq, int q_f32, int64_t T, int P, int Q,
              int slices, float* ws, float* grad, float* bias_grad, int trans, int accumulate, void* stream);

/* Up to four weight gradients over the same T tokens in one launch (the Linear layers of a layer: to_out and
 * to_q/k/v, performer_pytorch.py:374-376,402; w2 and w1, :323,330): bf16 operands only.  Each item's slice partials
 * are left in its ws ([slices][P][208] f32, scm_wgrad_ws_floats(P, slices) floats, the bias column at index Q when
 * with_bias) for scm_reduce_batch, exactly as scm_wgrad leaves them with grad == NULL.  The P tiles (128 wide) of all
 * items share one round of workgroups, so `slices` is chosen for their total (about 256 / total tiles): fewer slices
 * per product than separate calls would use, so fewer partials written and read back, and one launch. */
typedef struct scm_wgrad_item {
  const void* Pm; int64_t ldp;   /* [T,P] bf16 */
  const void* Qm; int64_t ldq;   /* [T,Q] bf16 */
  int P, Q, with_bias;
  float* ws;
} scm_wgrad_item;
int scm_wgrad_multi(const scm_wgrad_item* items, int n, int64_t T, int slices, void* stream);

/* ---- deferred reductions: one launch for the small reductions that end the gradient kernels ----------------
 * A backward pass runs some forty of them (24 weight-gradient slab sums, 14 column sums); each is a few microseconds
 * of launch and drain.  The producing entry points can leave their partials in the workspace (scm_wgrad with
 * grad == NULL; scm_ln_bwd / scm_head_bwd / scm_embed_bwd with NULL gradient outputs) and one call sums them all, each
 * element in the same fixed order as the immediate reduction (bit-identical results).
 *   kind 0  weight-gradient slabs: src = ws of scm_wgrad, S slices, out0 (+)= sum (transposed when trans),
 *           out1 (may be NULL) (+)= the bias column;
 *   kind 1  column partials: src = [S][P] rows (P = row width); out_k[i] (+)= sum_s src[s][lo[k] + i], i < hi[k] - lo[k],
 *           for the outputs that are not NULL. */
typedef struct {
  int32_t kind, accumulate;
  const float* src;
  int32_t S, P, Q, trans;
  float* out[4];
  int32_t lo[4], hi[4];
} scm_reduce_job;
int scm_reduce_batch(const scm_reduce_job* jobs /* host */, int n, void* stream);

/* ---- FAVOR+ attention: FastAttention.forward performer_pytorch.py:233-250 =
 *      softmax_kernel (:67-92, query and key modes) + linear_attention (:144-149) --------------- */
typedef struct {
  int B, H, N;
  int kmax_chunks; /* key-stabiliser partials per (b,h): 9 per chunk of the K'^T V pass */
  int ctx_chunks;  /* token chunks per (b,h) of the K'^T V pass (partials in ws) */
  int apply_chunks;/* token chunks per (b,h) of the output pass */
  int group_cells; /* cells per key-stabiliser group: the key features of performer_pytorch.py:89-90 are stabilised by
                    * ONE max over everything in a call, so a launch that batches several reference calls (the
                    * micro-batches of one gradient-accumulation window, s1_pretrain_rna.py:294-305) must keep one max
                    * per call: cells [g*group_cells, (g+1)*group_cells) share a stabiliser (the last group may be
                    * smaller).  scm_favor_make_plan sets it to B (one call); the caller may lower it (1 .. B) */
} scm_favor_plan;
/* fills the chunk counts for the MI355X (256 CUs); group_cells = B */
int scm_favor_make_plan(int B, int H, int N, scm_favor_plan* plan /* host */);
int64_t scm_favor_ws_floats(const scm_favor_plan* plan /* host */);
/* out = attention output; out_centered = out - obar[b,h,:] where obar is the (k'-weighted) mean value vector
 * of the head: kept for the backward, whose denominator path needs it at full relative precision.
 * Saved state of one forward call (caller-allocated):
 *   ctx_bf16 [B*H,288,64] bf16 (centred K'^T V), z [B*H,288] f32, sums [B*H,80] f32 (obar, sum z, key stabiliser of
 *   the (b,h)'s group),
 *   rowmax [B*H*N] f32, den [B*H*N] f32,
 *   kmax [kmax_chunks*B*H] f32 (partial maxima; their max over a stabiliser group is the key stabiliser of
 *   performer_pytorch.py:90),
 *   kmax_idx [kmax_chunks*B*H] i64 (where each partial maximum sits: (bh*N + n)*512 + feature).
 * When no backward follows, out_centered.p and rowmax / den (both) may be NULL: they are then not written
 * (a third less traffic in the output pass). */
int scm_favor_fwd(const scm_favor_plan* plan, scm_view4 q, scm_view4 k, scm_view4 v, const void* omega_b,
                  scm_view4 out, scm_view4 out_centered, void* ctx_bf16, float* z, float* sums, float* rowmax, float* den, float* kmax,
                  int64_t* kmax_idx, float* ws, void* stream);
/* Backward of the above: dq, dk, dv from dout, including the gradient that autograd routes through both
 * stabilisers (per-token argmax for queries, the single argmax of each stabiliser group for keys).  `o_centered` is the
 * second forward output.
 * Scratch: dctx_bf16/dz/dsums shaped like ctx_bf16/z/sums, dden [B*H*N]. */
int scm_favor_bwd(const scm_favor_plan* plan, scm_view4 q, scm_view4 k, scm_view4 v, scm_view4 o_centered,
                  scm_view4 dout,
                  const void* omega_b, const void* ctx_bf16, const float* z, const float* sums, const float* rowmax,
                  const float* den, const float* kmax, const int64_t* kmax_idx, scm_view4 dq, scm_view4 dk,
                  scm_view4 dv, void* dctx_bf16, float* dz, float* dsums, float* dden, float* ws, void* stream);

/* ---- the reference's free functions with MATERIALISED feature maps, exact f32 arithmetic (the model path above never
 *      writes q'/k' to memory; these serve callers of the functions themselves) -------------------------------
 * softmax_kernel(data, projection_matrix, is_query, normalize_data, eps), performer_pytorch.py:67-92:
 *   data f32 [B*H, N, 64] contiguous, projection f32 [M, 64], M <= 320 -> out f32 [B*H, N, M] =
 *   M^-1/2 * (exp((c data) proj^T - c^2 |data|^2 / 2 - stab) + eps), c = 64^-1/4 (1 when !normalize_data);
 *   stab = per-token max of the scores (is_query) or ONE max over the whole tensor (keys: two passes, ws of
 *   scm_feature_map_ws_floats(B*H, N) floats; ws may be NULL for queries). */
int64_t scm_feature_map_ws_floats(int64_t BH, int N);
int scm_feature_map(const float* data, const float* projection, int64_t BH, int N, int M, int is_query,
                    int normalize_data, float eps, float* out, float* ws, void* stream);
/* linear_attention(q, k, v), performer_pytorch.py:144-149: q', k' f32 [B*H, N, M], v f32 [B*H, N, 64] ->
 * out f32 [B*H, N, 64] = (q' (k'^T v)) / (q' . sum_n k').  ws: scm_linear_attention_ws_floats(B*H, N, M) floats. */
int64_t scm_linear_attention_ws_floats(int64_t BH, int N, int M);
int scm_linear_attention(const float* qf, const float* kf, const float* v, int64_t BH, int N, int M, float* out,
                         float* ws, void* stream);

/* ---- final LayerNorm + head: performer_pytorch.py:628-634 (Linear(dim,num_tokens)); rna_encoder.py:5-16
 *      (1x200 conv == dot product, + ReLU; C = 1, relu = 1) -------------------------------------- */
/* enc (may be NULL) f32 [T,D] = LN(x); out (may be NULL when C == 0) f32 [T,C]. */
int scm_head_fwd(const float* x, const float* gamma, const float* beta, const float* W, const float* bias, int64_t T,
                 int D, int C, int relu, float* enc, float* out, float* mean, float* rstd, void* stream);
int64_t scm_head_bwd_ws_floats(int64_t T, int D, int C);
/* dout [T,C] and/or denc [T,D] (either may be NULL) -> dx (and its bf16 rounding dx_bf16, may be NULL);
 * dgamma, dbeta, dW [C,D], db [C] (+)= . */
int scm_head_bwd(const float* dout, const float* denc, const float* out, const float* x, const float* mean,
                 const float* rstd, const float* gamma, const float* beta, const float* W, int64_t T, int D, int C,
                 int relu, float* dx, void* dx_bf16, float* ws, float* dgamma, float* dbeta, float* dW, float* db,
                 int accumulate, void* stream);

/* ---- S1 masked-autoencoding step tail: s1_pretrain_rna.py:88-121 (data_mask, get_mask_subset_with_prob,
 *      mask_with_tokens, prob_mask_like) and :268,297 (CrossEntropyLoss(ignore_index, mean)) ------------------ */
/* Per cell (row of tokens [B,N], int64): tokens 0 and pad_id are never maskable; the ceil(f32(count) * mask_prob)
 * maskable positions with the largest rand_pick draws are picked (at most max_masked = ceil(mask_prob * N); equal
 * draws at the boundary: lowest positions first); picked positions with rand_replace < replace_prob become
 * mask_id in masked_input; labels = token at picked positions, pad_id elsewhere.  Draws are f32 in [0,1).
 * N <= 20480. */
int scm_s1_mask(const int64_t* tokens, const float* rand_pick, const float* rand_replace, int B, int N, float mask_prob,
                float replace_prob, int max_masked, int64_t pad_id, int64_t mask_id, int64_t* masked_input,
                int64_t* labels, void* stream);
/* The tokens form loss groups of group_tokens consecutive tokens (the last may be shorter; group_tokens <= 0 or
 * >= T: one group).  One group = one call of the reference's loss: a launch that batches the micro-batches of a
 * gradient-accumulation window (s1_pretrain_rna.py:294-305) keeps one mean per micro-batch.  For group g:
 * loss_count[2g] = mean over its tokens with label != ignore_index of -log softmax(logits[t,:])[label[t]],
 * loss_count[2g+1] = number of such tokens (f32).  logits f32 [T,C], C <= 8.
 * ws: scm_ce_ws_floats(T, group_tokens) floats; loss_count: 2 * scm_ce_groups(T, group_tokens) floats. */
int64_t scm_ce_ws_floats(int64_t T, int64_t group_tokens);
int64_t scm_ce_groups(int64_t T, int64_t group_tokens);
int scm_ce_fwd(const float* logits, const int64_t* labels, int64_t T, int C, int64_t ignore_index, int64_t group_tokens,
               float* ws, float* loss_count, void* stream);
/* dlogits[t,:] = d loss_g / d logits[t,:] * dloss[g] for the tokens t of group g (dloss: device vector with one
 * incoming gradient per group). */
int scm_ce_bwd(const float* logits, const int64_t* labels, int64_t T, int C, int64_t ignore_index, int64_t group_tokens,
               const float* dloss, const float* loss_count, float* dlogits, void* stream);

/* Accuracy of the S1 loop (s1_pretrain_rna.py:308-312): per cell b, counts[b] = {number of positions with
 * label != pad_id whose prediction 1 + argmax(logits[b,n,1:C-1]) equals the label, number of such positions}
 * (int32 [B,2]) and ratio[b] = correct / counted (f32 [B]; nan when nothing is counted).  logits f32 [B,N,C].
 * ws: scm_s1_accuracy_ws_ints(B, N) int32 of scratch (per-workgroup partial counts, summed in a fixed order by a
 * second small launch; nothing needs clearing between calls). */
int64_t scm_s1_accuracy_ws_ints(int B, int N);
int scm_s1_accuracy(const float* logits, const int64_t* labels, int B, int N, int C, int64_t pad_id, float* ratio,
                    int* counts, int* ws, void* stream);

/* ---- the tail of scMomer.get_cell_embedding after the encoder: scmomer.py:96-113 (rna_ln, encoder, to_out on the
 *      concatenation) with _get_rna_features :124-130 (rna_projection, L2 normalisation) and the student encoder of
 *      extract_cell_emb.py:27-40 (Linear, LayerNorm, GELU, Dropout (eval: off), Linear).  Forward only, f32, two
 *      launches (one split-K kernel for the two G-wide products, one workgroup per cell for the narrow rest):
 *        out = to_out(cat(W2 gelu(LN_h(W1 x + b1)) + b2,  LN_p(normalised(Wp genes))))
 *      genes [B,G] f32 = the encoder's per-gene output (rna_encoder.py:11-16), x [B,G] f32 = the expression values.
 *      All weights f32, row-major as nn.Linear holds them.  ws: scm_cell_tail_ws_floats(B, G, P, Hd) floats. */
typedef struct {
  int32_t P, Hd, O, normalize;       /* projection_dim, student hidden width, output width, config.normalize */
  const float* Wp;                   /* rna_projection.weight [P, G] */
  const float* ln_p_w; const float* ln_p_b;   /* rna_ln [P] */
  const float* W1; const float* b1;  /* encoder.encoder.0 [Hd, G], [Hd] */
  const float* ln_h_w; const float* ln_h_b;   /* encoder.encoder.1 [Hd] */
  const float* W2; const float* b2;  /* encoder.encoder.4 [P, Hd], [P] */
  const float* Wo; const float* bo;  /* to_out [O, 2 P], [O] */
} scm_cell_tail_args;
int64_t scm_cell_tail_ws_floats(int B, int G, int P, int Hd);
int scm_cell_tail_fwd(const float* genes, const float* x, int B, int G, const scm_cell_tail_args* a /* host */,
                      float* ws, float* out, void* stream);

/* ---- the trainable tail of the S3 distillation step: s3_pretrain_missing.py:291-307 (BCEWithLogitsLoss with
 *      pos_weight on the binarised ATAC target :262,292-294, MSE on the RNA reconstruction, distill_weight * MSE
 *      between the student's embedding and the teacher's, backward :299) over scMomer.forward(mode='one',
 *      reconstruct=True) (scmomer.py:41-94) with the student encoder of extract_cell_emb.py:27-40 and the linear
 *      decoders of utils.py:168-179.  The Performer, rna_projection and rna_ln are frozen (:233-246): `genes` is the
 *      encoder's per-gene output and gets no gradient.  f32 throughout.
 *      scm_distill_fwd: four launches -> per loss group g (see group_cells; one group by default) losses[4g .. 4g+3] =
 *        {BCE, RNA MSE, distillation MSE, BCE + RNA MSE + distill_weight * distillation MSE} (means over the group's
 *        cells, as torch forms them; the last is the sum the script backpropagates, :295), and the tape the backward reads: the student's normalised hidden row, its dropped GELU output,
 *        cat(student embedding, LayerNorm(normalised projection)), to_out's output, the LayerNorm's 1/std.
 *      scm_distill_bwd: four launches -> the gradients of the sum over the loss groups of gloss[0] BCE + gloss[1]
 *        MSE_rna + gloss[2] MSE_distill (gloss: three DEVICE floats; the micro-batches of an accumulation window,
 *        s3_pretrain_missing.py:296-299, in one call) for the twelve trainable tensors, written (accumulate = 0) or added
 *        (accumulate != 0: gradient accumulation, s3_pretrain_missing.py:296-299) to the given buffers.
 *      drop: the dropout layer's multipliers (0 or 1 / (1 - p)) per hidden unit [B, Hd], drawn by the caller; NULL =
 *        dropout off.  pos_weight: DEVICE, 1 value or one per peak.  B <= scm_distill_max_cells(); O must be 128.
 *      tape: scm_distill_tape_floats(B, P, Hd, O) floats; ws: scm_distill_ws_floats(B, G, P, Hd, O, Na) floats, the
 *      same buffer in the forward and its backward. */
typedef struct {
  scm_cell_tail_args tail;           /* the weights up to to_out, and config.normalize */
  int32_t B, G, Na;                  /* cells, genes (= RNA decoder rows), ATAC peaks (= ATAC decoder rows) */
  int32_t pos_weight_n;              /* 1 or Na */
  float distill_weight;              /* weight of the distillation term in losses[3] */
  int32_t group_cells;               /* > 0: the batch is B / group_cells micro-batches stacked (B a multiple), one set
                                      * of loss means per micro-batch; 0: one group */
  const float* Wa; const float* ba;  /* ATAC.output_mean.0 [Na, O], [Na] */
  const float* Wr; const float* br;  /* RNA.output_mean.0 [G, O], [G] */
  const float* genes;                /* [B, G] the encoder's per-gene output */
  const float* x;                    /* [B, G] expression values: student input and RNA target */
  const float* atac_raw;             /* [B, Na]; the target is atac_raw > 0 */
  const float* latent;               /* [B, P] teacher embedding */
  const float* drop;                 /* [B, Hd] or NULL */
  const float* pos_weight;           /* device */
} scm_distill_args;
typedef struct {
  float* dW1; float* db1; float* dln_w; float* dln_b; float* dW2; float* db2;   /* student encoder */
  float* dWo; float* dbo;                                                       /* to_out */
  float* dWa; float* dba; float* dWr; float* dbr;                               /* decoders */
} scm_distill_grads;
int scm_distill_max_cells(void);
int64_t scm_distill_tape_floats(int B, int P, int Hd, int O);
int64_t scm_distill_ws_floats(int B, int G, int P, int Hd, int O, int Na);
int scm_distill_fwd(const scm_distill_args* a /* host */, float* tape, float* ws, float* losses, void* stream);
int scm_distill_bwd(const scm_distill_args* a /* host */, const float* tape, float* ws, const float* gloss,
                    const scm_distill_grads* g /* host */, int accumulate, void* stream);

/* ---- the optimizer step of the S1 loop: s1_pretrain_rna.py:303-305 (clip_grad_norm_(model.parameters(), 100),
 *      optimizer.step() of the Adam built at :247, optimizer.zero_grad()) over every parameter tensor in THREE launches.
 *      The table (DEVICE memory) lists chunks of at most scm_adam_chunk_elems() consecutive elements, each inside one
 *      parameter tensor; the chunks of a tensor are consecutive and carry the tensor's number.  Per element, with
 *      g = grad * grad_scale (grad_scale = 1 / world size after the gradient all-reduce):
 *        norm = sqrt(sum over tensors of |g_tensor|^2) as torch forms it, coef = min(1, max_norm / (norm + 1e-6))
 *               (max_norm <= 0: no clipping), g *= coef;
 *        exp_avg += (g - exp_avg) (1 - beta1);  exp_avg_sq = exp_avg_sq beta2 + (1 - beta2) g g;
 *        param -= lr / (1 - beta1^step) * exp_avg / (sqrt(exp_avg_sq) / sqrt(1 - beta2^step) + eps);  grad = 0
 *      i.e. torch.optim.Adam (no weight decay, no amsgrad) operation by operation; `step` is the step number AFTER
 *      the increment (>= 1); the hyper-parameters are doubles, as torch holds them (1 - beta is formed in double).
 *      ws: n_chunks + 2 floats.  norm_out (may be NULL): the total norm before clipping. */
typedef struct {
  float* p;        /* parameter */
  float* g;        /* its gradient (accumulator) */
  float* m;        /* exp_avg */
  float* v;        /* exp_avg_sq */
  int32_t n;       /* elements of this chunk */
  int32_t tensor;  /* number of the parameter tensor the chunk belongs to */
} scm_adam_item;
int scm_adam_chunk_elems(void);
int scm_adam_step(const scm_adam_item* items /* device */, int n_chunks, float* ws, double grad_scale, double max_norm,
                  double lr, double beta1, double beta2, double eps, int64_t step, float* norm_out, void* stream);

/* ---- data ingest: SCDataset.__getitem__ (s1_pretrain_rna.py:133-137: data[index].toarray()[0], values above
 *      class_limit-2 set to it, .long()) and the scMomer rule (scmomer.py:56,108: round().long().clamp(0,5)) -----
 * The expression matrix is CSR in device memory (indptr int64 [n_rows+1], indices int32, values f32; duplicate
 * column indices of a row add up).  Cell b of the batch is row rows[b] (rows == NULL: row b).  tokens int64 [B,N].
 * rule 0: v > limit -> limit, then truncation toward zero; rule 1: round half to even, clamp to [0, limit].
 * *err_flag (device int, caller zeroes it) gets bit 0 for a row index outside [0,n_rows) and bit 1 for a column
 * index outside [0,N); such entries are skipped.  N <= scm_bin_max_genes() (the dense row is built in LDS).
 * Bit-exact with the reference's bins for canonical CSR (no duplicate column in a row) and for pairs of
 * duplicates; with three or more duplicates of one gene their f32 sum is taken in another order than scipy's
 * toarray(), which can move a value across a bin edge by one ulp. */
int scm_bin_max_genes(void);
int scm_bin_csr(const int64_t* indptr, const int32_t* indices, const float* values, const int64_t* rows,
                int64_t n_rows, int B, int N, int rule, int limit, int64_t* tokens, int* err_flag, void* stream);
/* The same two rules on dense f32 values [total]. */
int scm_bin_dense(const float* x, int64_t total, int rule, int limit, int64_t* tokens, void* stream);

/* ---- projection redraw: gaussian_orthogonal_random_matrix + orthogonal_matrix_chunk
 *      (performer_pytorch.py:110-139) for nmat matrices at once, from Gaussian draws the caller made --------------
 * gauss_blocks f32 [nmat, ceil(nb_rows/64), 64, 64]: the unstructured blocks; each is QR-factorised (Householder,
 * LAPACK conventions) and contributes the rows of Q^T (the last block only nb_rows % 64 of them).
 * scaling 0: row r is multiplied by the 2-norm of gauss_mult[m, r, :] (f32 [nmat, nb_rows, 64]); scaling 1: by 8.
 * omega f32 [nmat, nb_rows, 64]. */
int scm_orthogonal_omega(const float* gauss_blocks, const float* gauss_mult, int nmat, int nb_rows, int scaling,
                         float* omega, void* stream);

/* ---- measurement aid (no reference counterpart): per-kernel durations from HIP events recorded on the
 *      launch stream around each launch of the selected kernel.  scm_trace_enable(kid): -1 off,
 *      scm_trace_kernel_count() = every kernel, else one kernel id.  scm_trace_collect waits for the recorded
 *      events and returns, per kernel id, summed milliseconds, summed algorithmic GEMM-term FLOPs
 *      (recomputation not counted), summed algorithmic HBM bytes (each operand read once, each result written
 *      once) and launch counts, then clears the log.  Arrays are host memory of
 *      scm_trace_kernel_count() entries.  Tracing creates events: keep it off in production loops. */
int scm_trace_kernel_count(void);
const char* scm_trace_kernel_name(int kid); /* host */
int scm_trace_enable(int kid);
int scm_trace_collect(double* ms /* host */, double* flops /* host */, double* bytes /* host */,
                      int64_t* launches /* host */);

#ifdef __cplusplus
}
#endif
#endif
