// Host-side launch interface shared by the kernel translation units and the C ABI wrappers.
#pragma once
#include "scm_common.hpp"
#include "../../include/scmomer_hip.h"
#include <atomic>

namespace scm {

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: set it once on every device
// the kernel is launched on (bit d of `done` = device d is set up; device numbers fold modulo 64, where a repeat
// is harmless), from any thread (setting it twice is idempotent), and report a failure to the caller.
inline hipError_t ensure_dynamic_lds(const void* kernel, size_t bytes, std::atomic<unsigned long long>& done) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const unsigned long long bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
  return e;
}

struct GemmEpi {
  void* out0; long ldo0;
  void* out1; long ldo1;
  const float* bias;
  const float* resid; long ldr;
  const bf16* pre; long ldp;
  // LayerNorm-backward epilogue (gemm_ln_bwd): out0 = dx f32 [M,N], resid = dres (row pitch N), and
  const float* ln_x; const float* ln_mean; const float* ln_rstd; const float* ln_gamma;
  bf16* ln_dxb;        // dx rounded to bf16 (may be null)
  float* ln_partial;   // [workgroups][3 N] column partials: sum dy*xhat, sum dy, sum dres
  // LayerNorm-forward epilogue (gemm_res_ln): out0 = y f32 [M,N] = A W^T + bias + resid, and h = LN(y) with
  // ln_gamma / ln_beta into ln_dxb (bf16 [M,N]), the row statistics into ln_mean_out / ln_rstd_out
  const float* ln_beta; float* ln_mean_out; float* ln_rstd_out;
};

struct TensorBHND {  // element (b,h,n,d) at p + b*sb + h*sh + n*sn + d   (d contiguous)
  const bf16* p; long sb, sh, sn;
};
struct TensorBHND_out { bf16* p; long sb, sh, sn; };

constexpr int CTX_PART = NFP * DH + NFP + DH + 16;  // floats per (bh, chunk) partial record
constexpr int SUMS_STRIDE = 80;                     // Csum[64], zsum at [64]

struct CtxArgs {
  TensorBHND x;          // MODE 0: k ; MODE 1: q
  TensorBHND y;          // MODE 0: v ; MODE 1: dO
  const bf16* omega_b;   // [288][64]
  const float* stab;     // MODE 1: rowmax [B*H*N]
  int nstab;
  float* stab_out;       // MODE 0: [B*H][chunks][9] running maxima of k.w, one per wave (its 32 features)
  long long* key_out;    // MODE 0: where each was attained: (bh*N + n)*512 + feature
  const float* den;      // MODE 1: [B*H*N]
  const float* dden;     // MODE 1: [B*H*N]
  int H, N, chunk_len;   // chunk_len % 64 == 0
  float* part;
};

struct ApplyArgs {
  TensorBHND x;            // MODE 0,1: q ; MODE 2: k
  TensorBHND y;            // MODE 1: dO ; MODE 2: v
  TensorBHND y2;           // MODE 1: o - obar (saved centred forward output)
  const bf16* omega_b;     // [288][64]
  const bf16* Mb;          // [bh][288][64]  MODE 0,1: centred C ; MODE 2: dC
  const float* zf;         // [bh][288]      MODE 0,1: z ; MODE 2: dz
  const float* sums;       // [bh][80]       MODE 0: obar[64], sum(z) ; MODE 2: colsum(dC)
  const float* colpart;    // MODE 2, optional: [bh][9][80] tile partials whose sum is colsum(dC) (then sums is not read)
  const float* stab;       // MODE 1: rowmax [B*H*N] ; MODE 2: key stabiliser partials of (b,h) at stab + bh * stab_stride
  int nstab;               // MODE 2: partials per (b,h)
  long stab_stride;
  const float* den_in;     // MODE 1
  TensorBHND_out out0;     // MODE 0: o ; MODE 1: dq ; MODE 2: dk
  TensorBHND_out out1;     // MODE 0: o - obar ; MODE 2: dv
  float* rowmax_out;       // MODE 0
  float* den_out;          // MODE 0
  float* dden_out;         // MODE 1
  float* ksum_part;        // MODE 2: [workgroups] partial sums of E * dE' (key stabiliser gradient)
  int H, N, chunk_len;     // chunk_len % 32 == 0
};

constexpr int W_LDSLAB = 208;   // floats per wgrad slab row

// Optional per-kernel timing with HIP events on the launch stream (scm_trace_* in the C ABI).
enum KernelId {   // one id per kernel symbol family that a profiler lists separately
  K_GEMM_NT = 0,            // gemm_nt_kernel, + epilogue id (6 consecutive ids)
  K_GEMM_SKINNY = 6,        // gemm_skinny_kernel, + epilogue id (6 consecutive ids)
  K_WGRAD = 12, K_WGRAD_BIAS, K_WGRAD_REDUCE, K_CAST, K_EMBED_FWD, K_EMBED_BWD, K_LN_FWD, K_LN_BWD, K_HEAD_FWD,
  K_HEAD_BWD, K_COLSUM, K_PREP_OMEGA, K_CTX_FWD, K_CTX_BWD, K_CTX_FINALIZE, K_APPLY_FWD, K_APPLY_BWD_Q,
  K_APPLY_BWD_K, K_KFIX, K_S1_MASK, K_CE, K_BIN, K_OMEGA_QR, K_FEATURE_MAP, K_LINEAR_ATTENTION, K_FFN_CHAIN_FWD,
  K_REDUCE_BATCH, K_GEMM_LN_BWD, K_GEMM_RES_LN, K_ADAM, K_CELL_TAIL, K_DISTILL_TAIL, K_COUNT
};
void trace_begin(int kid, double flops, double bytes, hipStream_t s);
void trace_end(int kid, hipStream_t s);
struct Traced {  // brackets one launch
  int kid; hipStream_t s;
  // flops: algorithmic GEMM-term FLOPs of the launch; bytes: its algorithmic HBM bytes (operands read once,
  // results written once)
  Traced(int k, double flops, hipStream_t st, double bytes = 0) : kid(k), s(st) { trace_begin(k, flops, bytes, st); }
  ~Traced() { trace_end(kid, s); }
};

// gemm.hip
hipError_t gemm_nt(int epi, int a_f32, const void* A, long lda, const void* W, long ldw, int M, int N, int K,
                   GemmEpi e, hipStream_t s);
long long gemm_general_launches();   // launches of the general (register-staged) product kernel since load
// dx = dres + LN'(A W^T): the data-gradient product that ends a sublayer's backward with the LayerNorm backward
// as its epilogue.  Supported: bf16 A, N <= 224 with N % 4 == 0, K >= 128, lda <= 2340 (any M: tall inputs run as
// row slabs of 458,752 rows).
bool gemm_ln_bwd_supported(long M, int N, int K, long lda, long ldw);
int gemm_ln_bwd_blocks(long M);
hipError_t gemm_ln_bwd(const void* A, long lda, const void* W, long ldw, int M, int N, int K, GemmEpi e, hipStream_t s);
// y = A W^T + bias + resid (f32) and the following sublayer's LayerNorm h = LN(y) (bf16), mean, rstd in one launch.
// Supported as gemm_ln_bwd, with N <= 220.
bool gemm_res_ln_supported(long M, int N, int K, long lda, long ldw);
hipError_t gemm_res_ln(const void* A, long lda, const void* W, long ldw, int M, int N, int K, GemmEpi e, hipStream_t s);
hipError_t wgrad(const void* Pm, long ldp, const void* Qm, long ldq, int q_f32, int ones, int T, int P, int Q,
                 int slices, float* slab, hipStream_t s);
// up to four weight gradients over the same T tokens in one launch (bf16 operands; each token slice below 2 GiB),
// slice partials left in each product's slab
struct WgradDesc { const void* Pm; long ldp; const void* Qm; long ldq; float* slab; int P, Q, ones; };
hipError_t wgrad_multi(const WgradDesc* d, int n, int T, int slices, hipStream_t s);
hipError_t wgrad_reduce(const float* slab, int S, int P, int Q, float* grad, float* bias, int trans, int accumulate,
                        hipStream_t s);
hipError_t cast_weight(const float* src, int R, int C, void* dst, void* dstT, hipStream_t s);
using CastItem = ::scm_cast_item;   // the C ABI's table entry
hipError_t cast_weights_batch(const CastItem* items, int n, hipStream_t s);

// ffn_chain.hip: forward of the FeedForward sublayer as one kernel (model dim 200, hidden width % 8 == 0)
struct FfnChainArgs {
  long M; int F;
  const bf16* W1; long ldw1;      // first product: w1 [F, 200]
  const bf16* W2c; long ldw2;     // second product: w2 [200, F padded to 64] in chain k order
  const float* x; long ldx;       // f32 [M, 200]: input = residual
  const float* gamma; const float* beta;   // LayerNorm of the sublayer
  float* mean; float* rstd;       // out (may be null)
  const float* bias1; const float* bias2;
  bf16* h; long ldh;              // LayerNorm output (may be null)
  bf16* pre; bf16* act; long ldo0, ldo1;   // hidden pre-activation / activation (may be null)
  float* y; long ldy;
  const float* gamma_n; const float* beta_n; bf16* h_n; long ldhn; float* mean_n; float* rstd_n;  // optional LN of y
};
int ffn_chain_grid(long M);
bool ffn_chain_supported(long M, int D, int F);
hipError_t ffn_chain(FfnChainArgs a, hipStream_t s);

// norm_embed.hip
hipError_t embed_fwd(const int64_t* tok, const float* E, const double* G, int B, int N, int D, int V, float* out,
                     int* err, hipStream_t s);
int embed_bwd_blocks(long T);
hipError_t embed_bwd(const int64_t* tok, const float* dx, long T, int D, int V, float* partial, hipStream_t s);
hipError_t ln_fwd(const float* x, const float* g, const float* b, long T, int D, void* y, float* mean, float* rstd,
                  hipStream_t s);
int ln_bwd_blocks(long T);
constexpr int LN_STATIC_SCRATCH_FLOATS = 1024 * 3 * 256;   // ln_bwd_blocks(T) <= 1024, D <= 256
hipError_t ln_bwd_static_scratch(float** device_ptr);       // this device's address of the built-in scratch array
hipError_t ln_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                  const float* dres, long T, int D, float* dx, void* dx_b, float* partial, hipStream_t s);
hipError_t head_fwd(const float* x, const float* g, const float* b, const float* W, const float* bias, long T, int D,
                    int C, int relu, float* enc, float* out, float* mean, float* rstd, hipStream_t s);
hipError_t head_bwd(const float* dout, const float* denc, const float* out, const float* x, const float* mean,
                    const float* rstd, const float* gamma, const float* beta, const float* W, long T, int D, int C,
                    int relu, float* dx, void* dx_b, float* partial, hipStream_t s);
struct ColsumSegs { float* out[4]; int lo[4], hi[4]; };   // column segments [lo,hi) of a partial row -> outputs
hipError_t colsum_reduce(const float* partial, int nblocks, int width, const ColsumSegs& seg, int nseg, int accumulate,
                         hipStream_t s);
hipError_t colsum_reduce(const float* partial, int nblocks, int width, int lo, int hi, float* out, int accumulate,
                         hipStream_t s);

using ReduceJob = ::scm_reduce_job;
constexpr int RB_MAX = 40;                      // jobs per launch (they travel as kernel arguments)
hipError_t reduce_batch(const ReduceJob* jobs, int n, hipStream_t s);

// optim.hip
using AdamItem = ::scm_adam_item;
struct AdamScalars { float grad_scale, max_norm, one_minus_beta1, beta2, one_minus_beta2, eps, step_size, bc2_sqrt; };
int adam_chunk_elems();
hipError_t adam_step(const AdamItem* items, int nchunks, float* partial, AdamScalars k, float* norm_out, hipStream_t s);

// cell_tail.hip
using CellTailArgs = ::scm_cell_tail_args;
long cell_tail_ws_floats(int B, int G, int P, int Hd);
hipError_t cell_tail(const float* genes, const float* x, int B, int G, CellTailArgs a, float* ws, float* out,
                     hipStream_t s);
hipError_t cell_tail_products(const float* genes, const float* x, int B, int G, const CellTailArgs& a, float* ws,
                              int* nchunks, hipStream_t s);

// distill_tail.hip
using DistillArgs = ::scm_distill_args;
using DistillGrads = ::scm_distill_grads;
long distill_tape_floats(int B, int P, int Hd, int O);
long distill_ws_floats(int B, int G, int P, int Hd, int O, int Na);
int distill_max_cells();
hipError_t distill_fwd(const DistillArgs& a, float* tape, float* ws, float* losses, hipStream_t s);
hipError_t distill_bwd(const DistillArgs& a, const float* tape, float* ws, const float* gloss, const DistillGrads& gr,
                       int accumulate, hipStream_t s);

// s1_tail.hip
hipError_t s1_mask(const int64_t* tok, const float* rand_pick, const float* rand_replace, int B, int N, float mask_prob,
                   float replace_prob, int max_masked, long long pad_id, long long mask_id, int64_t* masked,
                   int64_t* labels, hipStream_t s);
int ce_blocks(long T);
int ce_groups(long T, long group_tokens);
hipError_t ce_fwd(const float* logits, const int64_t* labels, long T, int C, long long ignore, long group_tokens,
                  float* partial, float* out, hipStream_t s);
hipError_t ce_bwd(const float* logits, const int64_t* labels, long T, int C, long long ignore, long group_tokens,
                  const float* dloss, const float* loss_count, float* dlogits, hipStream_t s);
int s1_accuracy_chunks(int N);
hipError_t s1_accuracy(const float* logits, const int64_t* labels, int B, int N, int C, long long pad_id, float* ratio,
                       int* counts, int* ws, hipStream_t s);

// ingest.hip
constexpr int BIN_MAX_GENES = 36864;   // dense f32 row of one cell in LDS (144 KB)
hipError_t bin_csr(const int64_t* indptr, const int32_t* indices, const float* values, const int64_t* rows, int n_rows,
                   int B, int N, int rule, int limit, int64_t* tokens, int* err, hipStream_t s);
hipError_t bin_dense(const float* x, long total, int rule, int limit, int64_t* tokens, hipStream_t s);
hipError_t orthogonal_omega(const float* gauss_blocks, const float* gauss_mult, int nmat, int nb_rows, int scaling,
                            float* omega, hipStream_t s);

// favor.hip
hipError_t favor_prep_omega(const float* omega, void* out, hipStream_t s);
hipError_t favor_kfix(const float* part_val, const long long* part_key, int stab_per_bh, const float* ksum_part,
                      int sum_per_bh, int BH, int group_bh, const void* omega_b, TensorBHND_out dk, int H, int N,
                      hipStream_t s);
hipError_t favor_ctx(int mode, CtxArgs a, int B, int nchunks, hipStream_t s);
constexpr int CTX_FIN_SCRATCH = NFP * DH + NFP + NFT * SUMS_STRIDE;  // floats per (b,h) of finalize scratch
float* favor_fin_colpart(float* scratch, int BH);   // where favor_ctx_finalize leaves the column partials
hipError_t favor_ctx_finalize(const float* part, int nchunks, int BH, int mode, const float* stab, int stab_per_bh,
                              int group_bh, const bf16* omega_b, const float* sums_fwd, void* Mb, float* zf, float* sums, float* scratch, hipStream_t s);
hipError_t favor_apply(int mode, ApplyArgs a, int B, int nchunks, hipStream_t s);


// favor_free.hip: the reference's free functions with materialised f32 feature maps
constexpr int FREE_MAX_FEATURES = 320;
long feature_map_ws_floats(long BH, int N);
hipError_t feature_map(const float* data, const float* proj, long BH, int N, int M, int is_query, float normalizer,
                       float norm_sq, float ratio, float eps, float* out, float* ws, hipStream_t s);
long linear_attention_ws_floats(long BH, int N, int M);
hipError_t linear_attention(const float* qf, const float* kf, const float* v, long BH, int N, int M, float* out,
                            float* ws, hipStream_t s);

}  // namespace scm
