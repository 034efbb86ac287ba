// extern "C" entry points of libscmomer_hip.so (declared in include/scmomer_hip.h):
// argument validation + kernel sequencing.  No allocation, no device synchronisation.
#include "scm_launch.hpp"
#include "../../include/scmomer_hip.h"

#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
#include <vector>
#include <math.h>

using namespace scm;

static thread_local char g_err[512] = "";

static int fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return SCM_EINVAL;
}
static int hipfail(hipError_t e, const char* what) {
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
  return (int)e;
}
#define SCM_HIP(call, what)                          \
  do {                                               \
    hipError_t e__ = (call);                         \
    if (e__ != hipSuccess) return hipfail(e__, what); \
  } while (0)
#define SCM_REQ(cond, ...) \
  do {                     \
    if (!(cond)) return fail(__VA_ARGS__); \
  } while (0)

static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

static bool view_ok(const scm_view4& v) {
  return v.p && aligned16(v.p) && v.sb % 8 == 0 && v.sh % 8 == 0 && v.sn % 8 == 0;
}
static TensorBHND cview(const scm_view4& v) { return {(const bf16*)v.p, (long)v.sb, (long)v.sh, (long)v.sn}; }
static TensorBHND_out oview(const scm_view4& v) { return {(bf16*)v.p, (long)v.sb, (long)v.sh, (long)v.sn}; }

// ---------------------------------------------------------------------------- per-kernel event tracing
namespace scm {
static const char* const kKernelNames[K_COUNT] = {
    "gemm_nt/bf16", "gemm_nt/f32_bias_res", "gemm_nt/gelu_dual", "gemm_nt/dgelu", "gemm_nt/f32", "gemm_nt/bf16_bias",
    "gemm_skinny/bf16", "gemm_skinny/f32_bias_res", "gemm_skinny/gelu_dual", "gemm_skinny/dgelu", "gemm_skinny/f32",
    "gemm_skinny/bf16_bias", "wgrad", "wgrad/bias", "wgrad_reduce", "cast_weight", "embed_fwd", "embed_bwd", "ln_fwd", "ln_bwd", "head_fwd",
    "head_bwd", "colsum_reduce", "prep_omega", "favor_ctx_fwd", "favor_ctx_bwd", "favor_ctx_finalize",
    "favor_apply_fwd", "favor_apply_bwd_q", "favor_apply_bwd_k", "favor_kfix", "s1_mask", "cross_entropy", "bin_cells", "omega_qr", "feature_map", "linear_attention", "ffn_chain_fwd", "reduce_batch", "gemm_ln_bwd", "gemm_res_ln", "adam_step", "cell_tail", "distill_tail"};
struct TraceRec { int kid; double flops, bytes; hipEvent_t t0, t1; };
static int g_trace_kid = -1;  // -1 off, K_COUNT = every kernel, else one kernel id
static std::vector<TraceRec> g_trace;
static bool trace_on(int kid) { return g_trace_kid == K_COUNT || g_trace_kid == kid; }
void trace_begin(int kid, double flops, double bytes, hipStream_t s) {
  if (!trace_on(kid)) return;
  TraceRec r{kid, flops, bytes, nullptr, nullptr};
  if (hipEventCreate(&r.t0) != hipSuccess || hipEventCreate(&r.t1) != hipSuccess) return;
  (void)hipEventRecord(r.t0, s);
  g_trace.push_back(r);
}
void trace_end(int kid, hipStream_t s) {
  if (!trace_on(kid) || g_trace.empty()) return;
  (void)hipEventRecord(g_trace.back().t1, s);
}
}  // namespace scm

extern "C" {

int scm_version(void) { return 1; }

int scm_trace_kernel_count(void) { return K_COUNT; }
const char* scm_trace_kernel_name(int kid) { return (kid >= 0 && kid < K_COUNT) ? kKernelNames[kid] : ""; }
int scm_trace_enable(int kid) {
  SCM_REQ(kid >= -1 && kid <= K_COUNT, "scm_trace_enable: kernel id out of range");
  g_trace_kid = kid;
  return 0;
}
int scm_trace_collect(double* ms, double* flops, double* bytes, int64_t* launches) {
  SCM_REQ(ms && flops && bytes && launches, "scm_trace_collect: null pointer");
  for (int i = 0; i < K_COUNT; ++i) { ms[i] = 0; flops[i] = 0; bytes[i] = 0; launches[i] = 0; }
  for (auto& r : g_trace) {
    float dt = 0.f;
    if (hipEventSynchronize(r.t1) == hipSuccess && hipEventElapsedTime(&dt, r.t0, r.t1) == hipSuccess) {
      ms[r.kid] += dt; flops[r.kid] += r.flops; bytes[r.kid] += r.bytes; launches[r.kid] += 1;
    }
    (void)hipEventDestroy(r.t0);
    (void)hipEventDestroy(r.t1);
  }
  g_trace.clear();
  return 0;
}
const char* scm_last_error(void) { return g_err; }
int64_t scm_gemm_general_launches(void) { return (int64_t)gemm_general_launches(); }

int scm_cast_weights_batch(const scm_cast_item* items, int n, void* stream) {
  SCM_REQ(items && n > 0 && n <= 65535, "scm_cast_weights_batch: bad arguments");
  SCM_HIP(cast_weights_batch(items, n, (hipStream_t)stream), "scm_cast_weights_batch");
  return 0;
}

int scm_cast_weight(const float* src, int rows, int cols, void* dst, void* dstT, void* stream) {
  SCM_REQ(src && rows > 0 && cols > 0 && (dst || dstT), "scm_cast_weight: bad arguments");
  SCM_HIP(cast_weight(src, rows, cols, dst, dstT, (hipStream_t)stream), "scm_cast_weight");
  return 0;
}

int scm_prep_omega(const float* omega, void* omega_b, void* stream) {
  SCM_REQ(omega && omega_b && aligned16(omega_b), "scm_prep_omega: null or unaligned pointer");
  SCM_HIP(favor_prep_omega(omega, omega_b, (hipStream_t)stream), "scm_prep_omega");
  return 0;
}

int scm_embed_fwd(const int64_t* tok, const float* E, const double* G, int B, int N, int D, int V, float* out,
                  int* err, void* stream) {
  SCM_REQ(tok && E && out && err, "scm_embed_fwd: null pointer");
  SCM_REQ(B > 0 && N > 0 && D > 0 && D <= 256 && D % 4 == 0 && V > 0 && V <= 8,
          "scm_embed_fwd: need 0<D<=256, D%%4==0, 0<V<=8 (got D=%d V=%d)", D, V);
  SCM_REQ(aligned16(E) && aligned16(out), "scm_embed_fwd: E/out must be 16-byte aligned");
  SCM_HIP(embed_fwd(tok, E, G, B, N, D, V, out, err, (hipStream_t)stream), "scm_embed_fwd");
  return 0;
}

int64_t scm_embed_bwd_ws_floats(int64_t T, int D, int V) { return (int64_t)embed_bwd_blocks(T) * V * D; }

int scm_embed_bwd(const int64_t* tok, const float* dx, int64_t T, int D, int V, float* ws, float* dE, int accumulate,
                  void* stream) {
  SCM_REQ(tok && dx && ws, "scm_embed_bwd: null pointer");
  SCM_REQ(T > 0 && D > 0 && D <= 256 && D % 4 == 0 && V > 0 && V <= 8, "scm_embed_bwd: bad sizes");
  SCM_HIP(embed_bwd(tok, dx, T, D, V, ws, (hipStream_t)stream), "scm_embed_bwd");
  if (dE)
    SCM_HIP(colsum_reduce(ws, embed_bwd_blocks(T), V * D, 0, V * D, dE, accumulate, (hipStream_t)stream),
            "scm_embed_bwd(reduce)");
  return 0;
}

int scm_ln_fwd(const float* x, const float* gamma, const float* beta, int64_t T, int D, void* y, float* mean,
               float* rstd, void* stream) {
  SCM_REQ(x && gamma && beta && y && mean && rstd, "scm_ln_fwd: null pointer");
  SCM_REQ(T > 0 && D > 0 && D <= 256 && D % 4 == 0, "scm_ln_fwd: need 0<D<=256, D%%4==0 (got %d)", D);
  SCM_REQ(aligned16(x) && aligned16(gamma) && aligned16(beta) && aligned16(y), "scm_ln_fwd: unaligned pointer");
  SCM_HIP(ln_fwd(x, gamma, beta, T, D, y, mean, rstd, (hipStream_t)stream), "scm_ln_fwd");
  return 0;
}

int64_t scm_ln_bwd_ws_floats(int64_t T, int D) { return (int64_t)ln_bwd_blocks(T) * 3 * D; }

int scm_ln_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
               const float* dres, int64_t T, int D, float* dx, void* dx_bf16, float* ws, float* dgamma,
               float* dbeta, float* dres_colsum, int accumulate, void* stream) {
  SCM_REQ(dy && x && mean && rstd && gamma && dx && ws, "scm_ln_bwd: null pointer");
  SCM_REQ(T > 0 && D > 0 && D <= 256 && D % 4 == 0, "scm_ln_bwd: bad sizes");
  SCM_REQ(aligned16(dy) && aligned16(x) && aligned16(gamma) && aligned16(dx) && (!dres || aligned16(dres)) &&
              (!dx_bf16 || aligned16(dx_bf16)),
          "scm_ln_bwd: unaligned pointer");
  hipStream_t s = (hipStream_t)stream;
  SCM_HIP(ln_bwd(dy, x, mean, rstd, gamma, dres, T, D, dx, dx_bf16, ws, s), "scm_ln_bwd");
  const int nb = ln_bwd_blocks(T);
  const ColsumSegs seg{{dgamma, dbeta, dres_colsum, nullptr}, {0, D, 2 * D, 0}, {D, 2 * D, 3 * D, 0}};
  SCM_HIP(colsum_reduce(ws, nb, 3 * D, seg, 3, accumulate, s), "scm_ln_bwd(column sums)");
  return 0;
}

int scm_ln_bwd_noalloc(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                       const float* dres, int64_t T, int D, float* dx, void* dx_bf16, float* dgamma, float* dbeta,
                       float* dres_colsum, int accumulate, void* stream) {
  SCM_REQ(T > 0 && D > 0 && D <= 256 && D % 4 == 0, "scm_ln_bwd_noalloc: need T > 0, 0 < D <= 256, D %% 4 == 0");
  SCM_REQ((int64_t)ln_bwd_blocks(T) * 3 * D <= LN_STATIC_SCRATCH_FLOATS,
          "scm_ln_bwd_noalloc: T=%lld, D=%d need more than the built-in scratch of %d floats", (long long)T, D,
          LN_STATIC_SCRATCH_FLOATS);
  // every GPU holds its own copy of the array at its own address: the address looked up below is the current
  // device's, so the stream must belong to the current device
  int cur = -1;
  hipDevice_t sdev = -1;
  SCM_HIP(hipGetDevice(&cur), "scm_ln_bwd_noalloc(current device)");
  SCM_HIP(hipStreamGetDevice((hipStream_t)stream, &sdev), "scm_ln_bwd_noalloc(stream's device)");
  SCM_REQ((int)sdev == cur, "scm_ln_bwd_noalloc: the stream belongs to device %d but device %d is current: make the "
          "stream's device current before the call", (int)sdev, cur);
  float* ws = nullptr;
  SCM_HIP(ln_bwd_static_scratch(&ws), "scm_ln_bwd_noalloc(scratch address)");
  return scm_ln_bwd(dy, x, mean, rstd, gamma, dres, T, D, dx, dx_bf16, ws, dgamma, dbeta, dres_colsum, accumulate,
                    stream);
}

int scm_gemm_ln_bwd_supported(int64_t M, int D, int K, int64_t lda, int64_t ldw) {
  return (M > 0 && M < (1ll << 31) && D > 0 && K > 0 && lda % 8 == 0 && ldw % 8 == 0 && lda >= K && ldw >= K &&
          gemm_ln_bwd_supported((long)M, D, K, (long)lda, (long)ldw)) ? 1 : 0;
}
int64_t scm_gemm_ln_bwd_ws_floats(int64_t M, int D) { return (int64_t)gemm_ln_bwd_blocks((long)M) * 3 * D; }

int scm_gemm_ln_bwd(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t M, int D, int K, const float* x,
                    const float* mean, const float* rstd, const float* gamma, const float* dres, float* dx,
                    void* dx_bf16, float* ws, float* dgamma, float* dbeta, float* dres_colsum, int accumulate,
                    void* stream) {
  SCM_REQ(A && W && x && mean && rstd && gamma && dx && ws, "scm_gemm_ln_bwd: null pointer");
  SCM_REQ(scm_gemm_ln_bwd_supported(M, D, K, lda, ldw),
          "scm_gemm_ln_bwd: unsupported sizes (M=%lld D=%d K=%d): use scm_gemm_nt + scm_ln_bwd", (long long)M, D, K);
  SCM_REQ(aligned16(A) && aligned16(W) && aligned16(x) && aligned16(gamma) && aligned16(dx) &&
              (!dres || aligned16(dres)) && (!dx_bf16 || aligned16(dx_bf16)),
          "scm_gemm_ln_bwd: unaligned pointer");
  hipStream_t s = (hipStream_t)stream;
  GemmEpi e{};
  e.out0 = dx; e.ldo0 = D; e.resid = dres; e.ldr = D;
  e.ln_x = x; e.ln_mean = mean; e.ln_rstd = rstd; e.ln_gamma = gamma; e.ln_dxb = (bf16*)dx_bf16; e.ln_partial = ws;
  SCM_HIP(gemm_ln_bwd(A, lda, W, ldw, (int)M, D, K, e, s), "scm_gemm_ln_bwd");
  if (dgamma || dbeta || dres_colsum) {
    const ColsumSegs seg{{dgamma, dbeta, dres_colsum, nullptr}, {0, D, 2 * D, 0}, {D, 2 * D, 3 * D, 0}};
    SCM_HIP(colsum_reduce(ws, gemm_ln_bwd_blocks((long)M), 3 * D, seg, 3, accumulate, s), "scm_gemm_ln_bwd(column sums)");
  }
  return 0;
}

int scm_gemm_res_ln_supported(int64_t M, int D, int K, int64_t lda, int64_t ldw) {
  return (M > 0 && M < (1ll << 31) && D > 0 && K > 0 && lda % 8 == 0 && ldw % 8 == 0 && lda >= K && ldw >= K &&
          gemm_res_ln_supported((long)M, D, K, (long)lda, (long)ldw)) ? 1 : 0;
}

int scm_gemm_res_ln(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t M, int D, int K, const float* bias,
                    const float* resid, float* y, const float* gamma, const float* beta, void* h_bf16, float* mean,
                    float* rstd, void* stream) {
  SCM_REQ(A && W && y && gamma && beta && h_bf16 && mean && rstd, "scm_gemm_res_ln: null pointer");
  SCM_REQ(scm_gemm_res_ln_supported(M, D, K, lda, ldw),
          "scm_gemm_res_ln: unsupported sizes (M=%lld D=%d K=%d): use scm_gemm_nt + scm_ln_fwd", (long long)M, D, K);
  SCM_REQ(aligned16(A) && aligned16(W) && aligned16(y) && aligned16(gamma) && aligned16(beta) && aligned16(h_bf16) &&
              (!resid || aligned16(resid)) && (!bias || aligned16(bias)),
          "scm_gemm_res_ln: unaligned pointer");
  GemmEpi e{};
  e.out0 = y; e.ldo0 = D; e.bias = bias; e.resid = resid; e.ldr = D;
  e.ln_gamma = gamma; e.ln_beta = beta; e.ln_dxb = (bf16*)h_bf16; e.ln_mean_out = mean; e.ln_rstd_out = rstd;
  SCM_HIP(gemm_res_ln(A, lda, W, ldw, (int)M, D, K, e, (hipStream_t)stream), "scm_gemm_res_ln");
  return 0;
}

int scm_gemm_nt(int epi, int a_f32, const void* A, int64_t lda, const void* W, int64_t ldw, int M, int N, int K,
                void* out0, int64_t ldo0, void* out1, int64_t ldo1, const float* bias, const float* resid, int64_t ldr,
                const void* pre, int64_t ldp, void* stream) {
  SCM_REQ(A && W && out0, "scm_gemm_nt: null pointer");
  SCM_REQ(M > 0 && N > 0 && K > 0 && K % 8 == 0 && N % 8 == 0, "scm_gemm_nt: need K%%8==0, N%%8==0 (M=%d N=%d K=%d)",
          M, N, K);
  SCM_REQ(lda % 8 == 0 && ldw % 8 == 0 && lda >= K && ldw >= K, "scm_gemm_nt: lda/ldw must be >=K and %%8==0");
  SCM_REQ(aligned16(A) && aligned16(W), "scm_gemm_nt: A/W must be 16-byte aligned");
  SCM_REQ(epi >= 0 && epi <= 5, "scm_gemm_nt: unknown epilogue %d", epi);
  SCM_REQ(ldo0 >= N && ldo0 % 8 == 0 && aligned16(out0), "scm_gemm_nt: out0 needs ldo0 >= N, ldo0%%8==0, 16-byte base");
  SCM_REQ(!out1 || (ldo1 % 8 == 0 && aligned16(out1)), "scm_gemm_nt: out1 needs ldo1%%8==0, 16-byte base");
  SCM_REQ(!resid || (ldr % 4 == 0 && aligned16(resid)), "scm_gemm_nt: resid needs ldr%%4==0, 16-byte base");
  SCM_REQ(!pre || (ldp % 8 == 0 && aligned16(pre)), "scm_gemm_nt: pre needs ldp%%8==0, 16-byte base");
  SCM_REQ(!bias || aligned16(bias), "scm_gemm_nt: bias must be 16-byte aligned");
  if (epi == SCM_EPI_GELU_DUAL) SCM_REQ(out1 && ldo1 >= N, "scm_gemm_nt: GELU_DUAL needs out1");
  if (epi == SCM_EPI_DGELU) SCM_REQ(pre && ldp >= N, "scm_gemm_nt: DGELU needs pre");
  if (resid) SCM_REQ(ldr >= N, "scm_gemm_nt: ldr < N");
  GemmEpi e{out0, (long)ldo0, out1, (long)ldo1, bias, resid, (long)ldr, (const bf16*)pre, (long)ldp};
  SCM_HIP(gemm_nt(epi, a_f32, A, lda, W, ldw, M, N, K, e, (hipStream_t)stream), "scm_gemm_nt");
  return 0;
}

// ---------------------------------------------------------------------------- FeedForward sublayer forward, chained
int scm_ffn_chain_supported(int64_t M, int D, int F) { return ffn_chain_supported(M, D, F) ? 1 : 0; }

int scm_ffn_chain_fwd(const scm_ffn_fwd_args* p, void* stream) {
  SCM_REQ(p, "scm_ffn_chain_fwd: null argument block");
  SCM_REQ(ffn_chain_supported(p->M, p->D, p->F), "scm_ffn_chain_fwd: need D = 200, F %% 8 == 0, F <= 2048 (D=%d F=%d)",
          p->D, p->F);
  SCM_REQ(p->x && p->gamma && p->beta && p->w1 && p->w2c && p->y, "scm_ffn_chain_fwd: null pointer");
  SCM_REQ(p->ldx >= p->D && p->ldx % 4 == 0 && p->ldy >= p->D && p->ldy % 4 == 0 && p->ldw1 >= p->D && p->ldw1 % 8 == 0 &&
              p->ldw2c >= (p->F + 63) / 64 * 64 && p->ldw2c % 8 == 0, "scm_ffn_chain_fwd: bad leading dimension");
  SCM_REQ(aligned16(p->x) && aligned16(p->y) && aligned16(p->w1) && aligned16(p->w2c) && aligned16(p->gamma) &&
              aligned16(p->beta) && (!p->b1 || aligned16(p->b1)) && (!p->b2 || aligned16(p->b2)),
          "scm_ffn_chain_fwd: unaligned pointer");
  SCM_REQ((!p->h && !p->mean && !p->rstd) || (p->h && p->mean && p->rstd && p->ldh >= p->D && p->ldh % 8 == 0 && aligned16(p->h)),
          "scm_ffn_chain_fwd: h, mean and rstd go together (ldh %% 8 == 0)");
  SCM_REQ(!p->pre || (p->ldpre >= p->F && p->ldpre % 8 == 0 && aligned16(p->pre)), "scm_ffn_chain_fwd: bad pre");
  SCM_REQ(!p->act || (p->ldact >= p->F && p->ldact % 8 == 0 && aligned16(p->act)), "scm_ffn_chain_fwd: bad act");
  const bool nx = p->h_next != nullptr;
  SCM_REQ(nx ? (p->gamma_next && p->beta_next && p->mean_next && p->rstd_next && p->ldh_next >= p->D &&
                p->ldh_next % 8 == 0 && aligned16(p->h_next))
             : (!p->gamma_next && !p->beta_next && !p->mean_next && !p->rstd_next),
          "scm_ffn_chain_fwd: the next LayerNorm's five pointers go together");
  FfnChainArgs a{};
  a.M = p->M; a.F = p->F; a.W1 = (const bf16*)p->w1; a.ldw1 = p->ldw1; a.W2c = (const bf16*)p->w2c; a.ldw2 = p->ldw2c;
  a.x = p->x; a.ldx = p->ldx; a.gamma = p->gamma; a.beta = p->beta; a.mean = p->mean; a.rstd = p->rstd;
  a.bias1 = p->b1; a.bias2 = p->b2; a.h = (bf16*)p->h; a.ldh = p->ldh; a.pre = (bf16*)p->pre; a.act = (bf16*)p->act;
  a.ldo0 = p->ldpre; a.ldo1 = p->ldact; a.y = p->y; a.ldy = p->ldy;
  a.gamma_n = p->gamma_next; a.beta_n = p->beta_next; a.h_n = (bf16*)p->h_next; a.ldhn = p->ldh_next;
  a.mean_n = p->mean_next; a.rstd_n = p->rstd_next;
  SCM_HIP(ffn_chain(a, (hipStream_t)stream), "scm_ffn_chain_fwd");
  return 0;
}

int scm_wgrad_multi(const scm_wgrad_item* items, int n, int64_t T, int slices, void* stream) {
  SCM_REQ(items && n >= 1 && n <= 4, "scm_wgrad_multi: need 1..4 items");
  SCM_REQ(T > 0 && T < (1ll << 31) && slices > 0, "scm_wgrad_multi: bad T or slices");
  WgradDesc d[4];
  for (int i = 0; i < n; ++i) {
    const scm_wgrad_item* it = items + i;
    SCM_REQ(it->Pm && it->Qm && it->ws, "scm_wgrad_multi: null pointer");
    SCM_REQ(it->P > 0 && it->P % 8 == 0 && it->Q > 0 && it->Q <= 200 && it->Q % 8 == 0,
            "scm_wgrad_multi: need P%%8==0, Q<=200, Q%%8==0 (P=%d Q=%d)", it->P, it->Q);
    SCM_REQ(it->ldp >= it->P && it->ldq >= it->Q && it->ldp % 8 == 0 && it->ldq % 8 == 0 && aligned16(it->Pm) &&
                aligned16(it->Qm),
            "scm_wgrad_multi: bad leading dimension or alignment");
    {
      const int64_t slice_tokens = (((T + 63) / 64 + slices - 1) / slices) * 64;
      SCM_REQ(slice_tokens * it->ldp * 2 < 0x7FFFF000ll && slice_tokens * it->ldq * 2 < 0x7FFFF000ll,
              "scm_wgrad_multi: a token slice of 2 GiB or more: use more slices");
    }
    d[i] = WgradDesc{it->Pm, (long)it->ldp, it->Qm, (long)it->ldq, it->ws, it->P, it->Q, it->with_bias ? 1 : 0};
  }
  SCM_HIP(wgrad_multi(d, n, (int)T, slices, (hipStream_t)stream), "scm_wgrad_multi");
  return 0;
}

int64_t scm_wgrad_ws_floats(int P, int slices) { return (int64_t)slices * P * W_LDSLAB; }

int scm_wgrad(const void* Pm, int64_t ldp, const void* Qm, int64_t ldq, int q_f32, int64_t T, int P, int Q,
              int slices, float* ws, float* grad, float* bias_grad, int trans, int accumulate, void* stream) {
  SCM_REQ(Pm && Qm && ws, "scm_wgrad: null pointer");
  SCM_REQ(T > 0 && T < (1ll << 31) && P > 0 && P % 8 == 0 && Q > 0 && Q <= 200 && Q % 8 == 0 && slices > 0,
          "scm_wgrad: need P%%8==0, Q<=200, Q%%8==0 (P=%d Q=%d)", P, Q);
  SCM_REQ(ldp % 8 == 0 && ldp >= P && ldq >= Q && ldq % 8 == 0 && aligned16(Pm) && aligned16(Qm),
          "scm_wgrad: bad leading dimension or alignment");
  hipStream_t s = (hipStream_t)stream;
  SCM_HIP(wgrad(Pm, ldp, Qm, ldq, q_f32, bias_grad ? 1 : 0, (int)T, P, Q, slices, ws, s), "scm_wgrad");
  if (grad) SCM_HIP(wgrad_reduce(ws, slices, P, Q, grad, bias_grad, trans, accumulate, s), "scm_wgrad(reduce)");
  return 0;
}

int scm_reduce_batch(const scm_reduce_job* jobs, int n, void* stream) {
  SCM_REQ(jobs && n > 0, "scm_reduce_batch: no jobs");
  for (int i = 0; i < n; ++i) {
    const scm_reduce_job& j = jobs[i];
    SCM_REQ(j.src && j.S > 0 && j.P > 0, "scm_reduce_batch: job %d: null source or empty shape", i);
    if (j.kind == 0) {
      SCM_REQ(j.out[0] && j.Q > 0 && j.Q <= 200, "scm_reduce_batch: job %d: slab job needs out[0] and 0 < Q <= 200", i);
    } else {
      SCM_REQ(j.kind == 1, "scm_reduce_batch: job %d: kind is 0 or 1", i);
      for (int k = 0; k < 4; ++k)
        SCM_REQ(!j.out[k] || (j.lo[k] >= 0 && j.hi[k] > j.lo[k] && j.hi[k] <= j.P),
                "scm_reduce_batch: job %d: segment %d outside the row", i, k);
    }
  }
  SCM_HIP(reduce_batch(jobs, n, (hipStream_t)stream), "scm_reduce_batch");
  return 0;
}

// ---------------------------------------------------------------------------- FAVOR+
static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

int scm_favor_make_plan(int B, int H, int N, scm_favor_plan* plan) {
  SCM_REQ(plan && B > 0 && H > 0 && N > 0, "scm_favor_make_plan: bad arguments");
  const int bh = B * H;
  const int cus = 256;
  plan->B = B; plan->H = H; plan->N = N;
  plan->group_cells = B;   // one key stabiliser over the whole call, as one reference call has
  // Both passes run one workgroup per CU at a time (LDS / register bound), so the launch is a whole number of
  // rounds of 256 workgroups: pick the number of token chunks per (b,h) so that the last round is nearly full.
  // Few, long chunks are preferred (every workgroup first loads 74 KB of images; every context chunk costs a
  // 75 KB partial record): start from one round and add chunks only while the rounds are under 90 % full.
  auto pick = [&](int max_chunks) {
    int c = clampi(cus / bh, 1, max_chunks);
    auto fill = [&](int cc) { const double r = (double)bh * cc / cus; return r / (double)(long)(r + 0.999999); };
    if (fill(c) < 0.9)
      for (int cc = c; cc <= 4 * c + 4 && cc <= max_chunks; ++cc)
        if (fill(cc) >= 0.93) { c = cc; break; }
    return c;
  };
  plan->apply_chunks = pick((N + 31) / 32);
  plan->ctx_chunks = pick((N + 63) / 64);
  plan->kmax_chunks = NFT * plan->ctx_chunks;   // one key-stabiliser partial per (chunk, wave) of the context pass
  return 0;
}

int64_t scm_favor_ws_floats(const scm_favor_plan* p) {
  if (!p) return 0;
  return (int64_t)p->B * p->H * ((int64_t)p->ctx_chunks * CTX_PART + CTX_FIN_SCRATCH);
}

static int plan_ok(const scm_favor_plan* p) {
  return p && p->B > 0 && p->H > 0 && p->N > 0 && p->kmax_chunks > 0 && p->ctx_chunks > 0 && p->apply_chunks > 0 &&
         p->group_cells > 0 && p->group_cells <= p->B &&
         (int64_t)p->B * p->H <= 65535 && (int64_t)p->B * p->H * p->N < (1ll << 31);
}

int scm_favor_fwd(const scm_favor_plan* plan, scm_view4 q, scm_view4 k, scm_view4 v, const void* omega_b,
                  scm_view4 out, scm_view4 out_centered, void* ctx_bf16, float* z, float* sums, float* rowmax, float* den, float* kmax,
                  int64_t* kmax_idx, float* ws, void* stream) {
  SCM_REQ(plan_ok(plan), "scm_favor_fwd: bad plan (sizes > 0, 1 <= group_cells <= B)");
  SCM_REQ(view_ok(q) && view_ok(k) && view_ok(v) && view_ok(out) && (!out_centered.p || view_ok(out_centered)),
          "scm_favor_fwd: bad tensor view (null/unaligned)");
  SCM_REQ(omega_b && ctx_bf16 && z && sums && kmax && kmax_idx && ws, "scm_favor_fwd: null pointer");
  SCM_REQ((rowmax && den) || (!rowmax && !den), "scm_favor_fwd: rowmax and den go together");
  SCM_REQ(aligned16(omega_b) && aligned16(ctx_bf16), "scm_favor_fwd: unaligned pointer");
  hipStream_t s = (hipStream_t)stream;
  const int B = plan->B, H = plan->H, N = plan->N;
  SCM_REQ(plan->kmax_chunks == NFT * plan->ctx_chunks, "scm_favor_fwd: plan.kmax_chunks must be 9 * ctx_chunks");
  CtxArgs c{};
  c.x = cview(k); c.y = cview(v); c.omega_b = (const bf16*)omega_b;
  c.stab = nullptr; c.nstab = 0; c.den = nullptr; c.dden = nullptr;
  c.stab_out = kmax; c.key_out = (long long*)kmax_idx;
  c.H = H; c.N = N; c.part = ws;
  SCM_HIP(favor_ctx(0, c, B, plan->ctx_chunks, s), "favor_ctx");
  float* fin = ws + (int64_t)B * H * plan->ctx_chunks * CTX_PART;
  SCM_HIP(favor_ctx_finalize(ws, plan->ctx_chunks, B * H, 0, kmax, plan->kmax_chunks, plan->group_cells * H,
                             (const bf16*)omega_b, nullptr, ctx_bf16, z,
                             sums, fin, s),
          "favor_ctx_finalize");
  ApplyArgs a{};
  a.x = cview(q); a.omega_b = (const bf16*)omega_b; a.Mb = (const bf16*)ctx_bf16; a.zf = z; a.sums = sums;
  a.out0 = oview(out); a.out1 = oview(out_centered); a.rowmax_out = rowmax; a.den_out = den; a.H = H; a.N = N;
  SCM_HIP(favor_apply(0, a, B, plan->apply_chunks, s), "favor_apply");
  return 0;
}

int scm_favor_bwd(const scm_favor_plan* plan, scm_view4 q, scm_view4 k, scm_view4 v, scm_view4 o_centered,
                  scm_view4 dout,
                  const void* omega_b, const void* ctx_bf16, const float* z, const float* sums, const float* rowmax,
                  const float* den, const float* kmax, const int64_t* kmax_idx, scm_view4 dq, scm_view4 dk,
                  scm_view4 dv, void* dctx_bf16, float* dz, float* dsums, float* dden, float* ws, void* stream) {
  SCM_REQ(plan_ok(plan), "scm_favor_bwd: bad plan (sizes > 0, 1 <= group_cells <= B)");
  SCM_REQ(view_ok(q) && view_ok(k) && view_ok(v) && view_ok(o_centered) && view_ok(dout) && view_ok(dq) && view_ok(dk) &&
              view_ok(dv), "scm_favor_bwd: bad tensor view (null/unaligned)");
  SCM_REQ(omega_b && ctx_bf16 && z && sums && rowmax && den && kmax && kmax_idx && dctx_bf16 && dz && dsums &&
              dden && ws, "scm_favor_bwd: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const int B = plan->B, H = plan->H, N = plan->N;
  // 1. query side: dq and dden (needs C, z of the forward)
  ApplyArgs a{};
  a.x = cview(q); a.y = cview(dout); a.y2 = cview(o_centered); a.omega_b = (const bf16*)omega_b; a.Mb = (const bf16*)ctx_bf16;
  a.zf = z; a.sums = sums; a.stab = rowmax; a.den_in = den; a.out0 = oview(dq); a.dden_out = dden; a.H = H; a.N = N;
  SCM_HIP(favor_apply(1, a, B, plan->apply_chunks, s), "favor_apply(dq)");
  // 2. dC, dz = sum over tokens of the query features against dO/den and dden
  CtxArgs c{};
  c.x = cview(q); c.y = cview(dout); c.omega_b = (const bf16*)omega_b; c.stab = rowmax; c.nstab = 0; c.den = den;
  c.dden = dden; c.H = H; c.N = N; c.part = ws;
  SCM_HIP(favor_ctx(1, c, B, plan->ctx_chunks, s), "favor_ctx(dC)");
  float* fin = ws + (int64_t)B * H * plan->ctx_chunks * CTX_PART;
  SCM_HIP(favor_ctx_finalize(ws, plan->ctx_chunks, B * H, 1, nullptr, 0, 1, nullptr, sums, dctx_bf16, dz, dsums, fin, s),
          "favor_ctx_finalize(dC)");
  // 3. key side: dk, dv
  ApplyArgs g{};
  g.x = cview(k); g.y = cview(v); g.omega_b = (const bf16*)omega_b; g.Mb = (const bf16*)dctx_bf16; g.zf = dz;
  g.sums = dsums; g.colpart = favor_fin_colpart(fin, B * H); g.out0 = oview(dk); g.out1 = oview(dv);
  g.stab = sums + DH + 1; g.nstab = 1; g.stab_stride = SUMS_STRIDE;   // each (b,h)'s key stabiliser, left there by the forward
  g.H = H; g.N = N;
  g.ksum_part = ws;  // the chunk partials in ws were consumed by the finalize step above
  SCM_HIP(favor_apply(2, g, B, plan->apply_chunks, s), "favor_apply(dk,dv)");
  // 4. gradient through the global key stabiliser lands on one token
  SCM_HIP(favor_kfix(kmax, (const long long*)kmax_idx, plan->kmax_chunks, ws, plan->apply_chunks, B * H,
                     plan->group_cells * H, omega_b, oview(dk), H, N, s), "favor_kfix");
  return 0;
}

// ---------------------------------------------------------------------------- free functions (materialised, f32)
int64_t scm_feature_map_ws_floats(int64_t BH, int N) { return (BH > 0 && N > 0) ? feature_map_ws_floats(BH, N) : 0; }

int scm_feature_map(const float* data, const float* projection, int64_t BH, int N, int M, int is_query,
                    int normalize_data, float eps, float* out, float* ws, void* stream) {
  SCM_REQ(data && projection && out, "scm_feature_map: null pointer");
  SCM_REQ(is_query || ws, "scm_feature_map: key features need the workspace (scm_feature_map_ws_floats)");
  SCM_REQ(BH > 0 && BH <= 65535 && N > 0 && M > 0 && M <= FREE_MAX_FEATURES && BH * N < (1ll << 31),
          "scm_feature_map: need 0 < B*H <= 65535, 0 < M <= %d (got B*H=%lld N=%d M=%d)", FREE_MAX_FEATURES,
          (long long)BH, N, M);
  SCM_REQ(aligned16(data) && aligned16(projection), "scm_feature_map: data/projection must be 16-byte aligned");
  // the scalars as the reference forms them: Python doubles rounded to f32 where they meet the f32 tensor
  const double nrm = normalize_data ? pow((double)DH, -0.25) : 1.0;
  SCM_HIP(feature_map(data, projection, BH, N, M, is_query, (float)nrm, (float)(nrm * nrm),
                      (float)pow((double)M, -0.5), eps, out, ws, (hipStream_t)stream), "scm_feature_map");
  return 0;
}

int64_t scm_linear_attention_ws_floats(int64_t BH, int N, int M) {
  return (BH > 0 && N > 0 && M > 0) ? linear_attention_ws_floats(BH, N, M) : 0;
}

int scm_linear_attention(const float* qf, const float* kf, const float* v, int64_t BH, int N, int M, float* out,
                         float* ws, void* stream) {
  SCM_REQ(qf && kf && v && out && ws, "scm_linear_attention: null pointer");
  SCM_REQ(BH > 0 && BH <= 65535 && N > 0 && M > 0 && M <= FREE_MAX_FEATURES && BH * N < (1ll << 31),
          "scm_linear_attention: need 0 < B*H <= 65535, 0 < M <= %d", FREE_MAX_FEATURES);
  SCM_HIP(linear_attention(qf, kf, v, BH, N, M, out, ws, (hipStream_t)stream), "scm_linear_attention");
  return 0;
}

// ---------------------------------------------------------------------------- S1 step tail
int scm_s1_mask(const int64_t* tokens, const float* rand_pick, const float* rand_replace, int B, int N, float mask_prob,
                float replace_prob, int max_masked, int64_t pad_id, int64_t mask_id, int64_t* masked_input,
                int64_t* labels, void* stream) {
  SCM_REQ(tokens && rand_pick && rand_replace && masked_input && labels, "scm_s1_mask: null pointer");
  SCM_REQ(B > 0 && N > 0 && N <= 20 * 1024 && (int64_t)B * N < (1ll << 31),
          "scm_s1_mask: need 0 < N <= 20480 (a cell's keys are held in the registers of one workgroup)");
  SCM_REQ(mask_prob >= 0.f && mask_prob <= 1.f && replace_prob >= 0.f && replace_prob <= 1.f && max_masked >= 0,
          "scm_s1_mask: probabilities must be in [0,1]");
  SCM_HIP(s1_mask(tokens, rand_pick, rand_replace, B, N, mask_prob, replace_prob, max_masked, pad_id, mask_id,
                  masked_input, labels, (hipStream_t)stream), "scm_s1_mask");
  return 0;
}

static int64_t ce_group_tokens(int64_t T, int64_t group_tokens) { return (group_tokens <= 0 || group_tokens > T) ? T : group_tokens; }

int64_t scm_ce_ws_floats(int64_t T, int64_t group_tokens) {
  if (T <= 0) return 0;
  const int64_t gt = ce_group_tokens(T, group_tokens);
  return 2 * (int64_t)ce_blocks(gt) * ce_groups(T, gt);
}
int64_t scm_ce_groups(int64_t T, int64_t group_tokens) { return T <= 0 ? 0 : ce_groups(T, ce_group_tokens(T, group_tokens)); }

int scm_ce_fwd(const float* logits, const int64_t* labels, int64_t T, int C, int64_t ignore_index, int64_t group_tokens,
               float* ws, float* loss_count, void* stream) {
  SCM_REQ(logits && labels && ws && loss_count, "scm_ce_fwd: null pointer");
  SCM_REQ(T > 0 && C > 0 && C <= 8, "scm_ce_fwd: need 1 <= C <= 8");
  const int64_t gt = ce_group_tokens(T, group_tokens);
  SCM_REQ(ce_groups(T, gt) <= 65535, "scm_ce_fwd: more than 65535 loss groups");
  SCM_HIP(ce_fwd(logits, labels, T, C, ignore_index, gt, ws, loss_count, (hipStream_t)stream), "scm_ce_fwd");
  return 0;
}

int scm_ce_bwd(const float* logits, const int64_t* labels, int64_t T, int C, int64_t ignore_index, int64_t group_tokens,
               const float* dloss, const float* loss_count, float* dlogits, void* stream) {
  SCM_REQ(logits && labels && dloss && loss_count && dlogits, "scm_ce_bwd: null pointer");
  SCM_REQ(T > 0 && C > 0 && C <= 8, "scm_ce_bwd: need 1 <= C <= 8");
  const int64_t gt = ce_group_tokens(T, group_tokens);
  SCM_REQ(ce_groups(T, gt) <= 65535, "scm_ce_bwd: more than 65535 loss groups");
  SCM_HIP(ce_bwd(logits, labels, T, C, ignore_index, gt, dloss, loss_count, dlogits, (hipStream_t)stream), "scm_ce_bwd");
  return 0;
}

int64_t scm_s1_accuracy_ws_ints(int B, int N) { return (B > 0 && N > 0) ? 2 * (int64_t)B * s1_accuracy_chunks(N) : 0; }

int scm_s1_accuracy(const float* logits, const int64_t* labels, int B, int N, int C, int64_t pad_id, float* ratio,
                    int* counts, int* ws, void* stream) {
  SCM_REQ(logits && labels && ratio && counts && ws, "scm_s1_accuracy: null pointer");
  SCM_REQ(B > 0 && N > 0 && C >= 3 && C <= 64, "scm_s1_accuracy: need B, N > 0 and 3 <= C <= 64");
  SCM_HIP(s1_accuracy(logits, labels, B, N, C, pad_id, ratio, counts, ws, (hipStream_t)stream), "scm_s1_accuracy");
  return 0;
}

// ---------------------------------------------------------------------------- cell-embedding tail
static bool cell_tail_sizes_ok(int B, int G, int P, int Hd, int O) {
  return B > 0 && G > 0 && P > 0 && Hd > 0 && O > 0 && P + Hd <= 1024 && P <= 1024 && Hd <= 1024 && O <= 1024 &&
         (P + Hd) % 64 == 0;
}
int64_t scm_cell_tail_ws_floats(int B, int G, int P, int Hd) {
  return cell_tail_sizes_ok(B, G, P, Hd, 1) ? (int64_t)cell_tail_ws_floats(B, G, P, Hd) : 0;
}
int scm_cell_tail_fwd(const float* genes, const float* x, int B, int G, const scm_cell_tail_args* a, float* ws,
                      float* out, void* stream) {
  SCM_REQ(genes && x && a && ws && out, "scm_cell_tail_fwd: null pointer");
  SCM_REQ(cell_tail_sizes_ok(B, G, a->P, a->Hd, a->O),
          "scm_cell_tail_fwd: need P, Hd, O <= 1024, P + Hd <= 1024 and a multiple of 64 (P=%d Hd=%d O=%d)", a->P, a->Hd, a->O);
  SCM_REQ(a->Wp && a->ln_p_w && a->ln_p_b && a->W1 && a->b1 && a->ln_h_w && a->ln_h_b && a->W2 && a->b2 && a->Wo && a->bo,
          "scm_cell_tail_fwd: null weight pointer");
  SCM_HIP(cell_tail(genes, x, B, G, *a, ws, out, (hipStream_t)stream), "scm_cell_tail_fwd");
  return 0;
}

// ---------------------------------------------------------------------------- distillation tail
static const char* distill_args_problem(const scm_distill_args* a) {
  if (!a) return "null argument block";
  const scm_cell_tail_args& w = a->tail;
  if (!cell_tail_sizes_ok(a->B, a->G, w.P, w.Hd, w.O)) return "need P, Hd, O <= 1024, P + Hd <= 1024 and a multiple of 64";
  if (w.O != 128) return "the decoders' input width (to_out's output) must be 128";
  if (a->B > distill_max_cells()) return "at most 32 cells per call";
  if (a->Na <= 0) return "need at least one ATAC peak";
  if (a->group_cells < 0 || (a->group_cells > 0 && a->B % a->group_cells != 0)) return "group_cells must divide the number of cells";
  if ((int64_t)a->B * a->G >= (1ll << 31) || (int64_t)a->B * a->Na >= (1ll << 31)) return "operand too large";
  if (!(w.Wp && w.ln_p_w && w.ln_p_b && w.W1 && w.b1 && w.ln_h_w && w.ln_h_b && w.W2 && w.b2 && w.Wo && w.bo && a->Wa &&
        a->ba && a->Wr && a->br))
    return "null weight pointer";
  if (!(a->genes && a->x && a->atac_raw && a->latent && a->pos_weight)) return "null input pointer";
  if (a->pos_weight_n != 1 && a->pos_weight_n != a->Na) return "pos_weight holds 1 or Na values";
  return nullptr;
}
int scm_distill_max_cells(void) { return distill_max_cells(); }
int64_t scm_distill_tape_floats(int B, int P, int Hd, int O) {
  return (B > 0 && P > 0 && Hd > 0 && O > 0) ? (int64_t)distill_tape_floats(B, P, Hd, O) : 0;
}
int64_t scm_distill_ws_floats(int B, int G, int P, int Hd, int O, int Na) {
  return (cell_tail_sizes_ok(B, G, P, Hd, O) && Na > 0) ? (int64_t)distill_ws_floats(B, G, P, Hd, O, Na) : 0;
}
int scm_distill_fwd(const scm_distill_args* a, float* tape, float* ws, float* losses, void* stream) {
  const char* why = distill_args_problem(a);
  SCM_REQ(!why, "scm_distill_fwd: %s", why);
  SCM_REQ(tape && ws && losses, "scm_distill_fwd: null pointer");
  SCM_HIP(distill_fwd(*a, tape, ws, losses, (hipStream_t)stream), "scm_distill_fwd");
  return 0;
}
int scm_distill_bwd(const scm_distill_args* a, const float* tape, float* ws, const float* gloss,
                    const scm_distill_grads* g, int accumulate, void* stream) {
  const char* why = distill_args_problem(a);
  SCM_REQ(!why, "scm_distill_bwd: %s", why);
  SCM_REQ(tape && ws && gloss && g, "scm_distill_bwd: null pointer");
  SCM_REQ(g->dW1 && g->db1 && g->dln_w && g->dln_b && g->dW2 && g->db2 && g->dWo && g->dbo && g->dWa && g->dba &&
          g->dWr && g->dbr, "scm_distill_bwd: null gradient pointer");
  SCM_HIP(distill_bwd(*a, tape, ws, gloss, *g, accumulate, (hipStream_t)stream), "scm_distill_bwd");
  return 0;
}

// ---------------------------------------------------------------------------- optimizer step
int scm_adam_chunk_elems(void) { return adam_chunk_elems(); }

int scm_adam_step(const scm_adam_item* items, int n_chunks, float* ws, double grad_scale, double max_norm, double lr,
                  double beta1, double beta2, double eps, int64_t step, float* norm_out, void* stream) {
  SCM_REQ(items && ws && n_chunks > 0 && n_chunks <= (1 << 20), "scm_adam_step: null pointer or bad chunk count");
  SCM_REQ(step >= 1 && beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1. && eps > 0. && grad_scale > 0.,
          "scm_adam_step: need step >= 1, 0 <= beta < 1, eps > 0, grad_scale > 0");
  // the scalars as torch.optim.Adam forms them: Python doubles, rounded to f32 where they meet the tensors
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  AdamScalars k{(float)grad_scale, (float)max_norm, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,
                (float)(lr / bc1), (float)sqrt(bc2)};
  SCM_HIP(adam_step(items, n_chunks, ws, k, norm_out, (hipStream_t)stream), "scm_adam_step");
  return 0;
}

// ---------------------------------------------------------------------------- ingest and projection redraw
int scm_bin_max_genes(void) { return BIN_MAX_GENES; }

int scm_bin_csr(const int64_t* indptr, const int32_t* indices, const float* values, const int64_t* rows,
                int64_t n_rows, int B, int N, int rule, int limit, int64_t* tokens, int* err_flag, void* stream) {
  SCM_REQ(indptr && indices && values && tokens && err_flag, "scm_bin_csr: null pointer");
  SCM_REQ(B > 0 && N > 0 && N <= BIN_MAX_GENES && n_rows > 0 && n_rows < (1ll << 31),
          "scm_bin_csr: need 0 < N <= scm_bin_max_genes() and 0 < n_rows < 2^31");
  SCM_REQ((rule == 0 || rule == 1) && limit >= 0 && limit < (1 << 24), "scm_bin_csr: rule is 0 (S1) or 1 (scMomer)");
  SCM_REQ(rows || B <= n_rows, "scm_bin_csr: without a row list B must not exceed n_rows");
  SCM_HIP(bin_csr(indptr, indices, values, rows, (int)n_rows, B, N, rule, limit, tokens, err_flag, (hipStream_t)stream),
          "scm_bin_csr");
  return 0;
}

int scm_bin_dense(const float* x, int64_t total, int rule, int limit, int64_t* tokens, void* stream) {
  SCM_REQ(x && tokens, "scm_bin_dense: null pointer");
  SCM_REQ(total > 0, "scm_bin_dense: empty input");
  SCM_REQ((rule == 0 || rule == 1) && limit >= 0 && limit < (1 << 24), "scm_bin_dense: rule is 0 (S1) or 1 (scMomer)");
  SCM_HIP(bin_dense(x, total, rule, limit, tokens, (hipStream_t)stream), "scm_bin_dense");
  return 0;
}

int scm_orthogonal_omega(const float* gauss_blocks, const float* gauss_mult, int nmat, int nb_rows, int scaling,
                         float* omega, void* stream) {
  SCM_REQ(gauss_blocks && omega && (gauss_mult || scaling == 1), "scm_orthogonal_omega: null pointer");
  SCM_REQ(nmat > 0 && nmat <= 65535 && nb_rows > 0 && nb_rows <= 64 * 1024, "scm_orthogonal_omega: bad sizes");
  SCM_REQ(scaling == 0 || scaling == 1, "scm_orthogonal_omega: scaling is 0 or 1");
  SCM_HIP(orthogonal_omega(gauss_blocks, gauss_mult, nmat, nb_rows, scaling, omega, (hipStream_t)stream),
          "scm_orthogonal_omega");
  return 0;
}

// ---------------------------------------------------------------------------- head
int scm_head_fwd(const float* x, const float* gamma, const float* beta, const float* W, const float* bias, int64_t T,
                 int D, int C, int relu, float* enc, float* out, float* mean, float* rstd, void* stream) {
  SCM_REQ(x && gamma && beta && mean && rstd && (enc || out), "scm_head_fwd: null pointer");
  SCM_REQ(T > 0 && D > 0 && D <= 256 && D % 4 == 0 && C >= 0 && C <= 8, "scm_head_fwd: need D<=256, C<=8");
  SCM_REQ(!out || (W && bias && C > 0), "scm_head_fwd: out needs W, bias and C>0");
  SCM_REQ(aligned16(x) && aligned16(gamma) && aligned16(beta) && (!W || aligned16(W)) && (!enc || aligned16(enc)),
          "scm_head_fwd: unaligned pointer");
  SCM_HIP(head_fwd(x, gamma, beta, W, bias, T, D, out ? C : 0, relu, enc, out, mean, rstd, (hipStream_t)stream),
          "scm_head_fwd");
  return 0;
}

static int head_width(int D, int C) { return 2 * D + C * D + C; }
int64_t scm_head_bwd_ws_floats(int64_t T, int D, int C) { return (int64_t)ln_bwd_blocks(T) * head_width(D, C); }

int scm_head_bwd(const float* dout, const float* denc, const float* out, const float* x, const float* mean,
                 const float* rstd, const float* gamma, const float* beta, const float* W, int64_t T, int D, int C,
                 int relu, float* dx, void* dx_bf16, float* ws, float* dgamma, float* dbeta, float* dW, float* db,
                 int accumulate, void* stream) {
  SCM_REQ(x && mean && rstd && gamma && beta && dx && ws && (dout || denc), "scm_head_bwd: null pointer");
  SCM_REQ(T > 0 && D > 0 && D <= 256 && D % 4 == 0 && C >= 0 && C <= 8, "scm_head_bwd: need D<=256, C<=8");
  SCM_REQ(!dout || (W && C > 0 && (!relu || out)), "scm_head_bwd: dout needs W (and out when relu)");
  if (!dout) C = 0;
  hipStream_t s = (hipStream_t)stream;
  SCM_HIP(head_bwd(dout, denc, out, x, mean, rstd, gamma, beta, W, T, D, C, relu, dx, dx_bf16, ws, s), "scm_head_bwd");
  const int nb = ln_bwd_blocks(T), wd = head_width(D, C);
  const ColsumSegs seg{{dgamma, dbeta, C > 0 ? dW : nullptr, C > 0 ? db : nullptr},
                       {0, D, 2 * D, 2 * D + C * D},
                       {D, 2 * D, 2 * D + C * D, wd}};
  SCM_HIP(colsum_reduce(ws, nb, wd, seg, 4, accumulate, s), "scm_head_bwd(column sums)");
  return 0;
}

}  // extern "C"
