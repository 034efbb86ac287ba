// The reference's two free functions with MATERIALISED feature maps, in exact f32 arithmetic:
//   softmax_kernel   (performer_pytorch.py:67-92)   data [B,H,N,64], projection [M,64] -> [B,H,N,M]
//   linear_attention (performer_pytorch.py:144-149) q', k' [B,H,N,M], v [B,H,N,64]     -> [B,H,N,64]
// The model path (favor.hip) never writes q' / k' to memory; these kernels exist for callers of the free functions
// and as the f32 statement of the same math on the device.  Both are bound by the [N, M] f32 feature tensors they
// read or write (about 1 KB per token and head), not by arithmetic, so they use plain f32 vector FMAs: MFMA with f32
// operands runs at the vector rate anyway, and every product then rounds exactly like the reference's f32 path.
#include "scm_launch.hpp"

namespace scm {

constexpr int FM_TOK = 32;          // tokens per workgroup
constexpr int FM_MAXM = 320;        // features: 8 groups x 40 per thread
constexpr int FM_FPT = FM_MAXM / 8;
constexpr int FM_OP = DH + 4;       // LDS pitch of a projection row (16-byte aligned rows, conflict-free broadcasts)

// MODE 0: query features (per-token max).  MODE 1: key pass A: only the workgroup's max of the raw scores is
// written (partial[block]).  MODE 2: key pass B: features against the global max read from gmax[0].
template <int MODE>
__global__ __launch_bounds__(256) void feature_map_kernel(const float* __restrict__ data,
                                                          const float* __restrict__ proj, int N, int M,
                                                          float normalizer, float norm_sq, float ratio, float eps,
                                                          float* __restrict__ out,
                                                          float* __restrict__ partial, const float* __restrict__ gmax) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* om = (float*)smem;                        // [M][FM_OP]
  float* red = om + (size_t)M * FM_OP;             // [8][32] maxima, then [32] token maxima at red + 256
  float* tile = red + 8 * FM_TOK + FM_TOK;         // [32][M + 1] output tile (MODE 0, 2)
  const int tid = threadIdx.x, i = tid & 31, g = tid >> 5;
  const int bh = blockIdx.y, n0 = blockIdx.x * FM_TOK, n = n0 + i;
  const bool valid = n < N;
  for (int c = tid; c < M * (DH / 4); c += 256) {
    const int row = c >> 4, ch = c & 15;
    *(f32x4*)(om + row * FM_OP + 4 * ch) = *(const f32x4*)(proj + (size_t)row * DH + 4 * ch);
  }
  // this thread's token row, scaled first as the reference does: (normalizer * data) @ projection^T
  float x[DH];
  float ss = 0.f;
  const float* xr = data + ((size_t)bh * N + (valid ? n : 0)) * DH;
#pragma unroll
  for (int c = 0; c < DH / 4; ++c) {
    const f32x4 v = valid ? *(const f32x4*)(xr + 4 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) { ss += v[j] * v[j]; x[4 * c + j] = normalizer * v[j]; }
  }
  const float diag = ss / 2.0f * norm_sq;
  __syncthreads();
  float s[FM_FPT];
  float mx = -INFINITY;
#pragma unroll
  for (int f = 0; f < FM_FPT; ++f) {
    const int j = g + 8 * f;
    float acc = 0.f;
    if (j < M) {
      const float* orow = om + j * FM_OP;
#pragma unroll
      for (int c = 0; c < DH / 4; ++c) {
        const f32x4 o4 = *(const f32x4*)(orow + 4 * c);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc = fmaf(x[4 * c + jj], o4[jj], acc);
      }
      if (valid) mx = fmaxf(mx, acc);
    }
    s[f] = acc;
  }
  red[g * FM_TOK + i] = mx;
  __syncthreads();
  if (tid < FM_TOK) {
    float m = red[tid];
#pragma unroll
    for (int q = 1; q < 8; ++q) m = fmaxf(m, red[q * FM_TOK + tid]);
    red[8 * FM_TOK + tid] = m;
  }
  __syncthreads();
  if (MODE == 1) {
    if (tid < 64) {
      float m = tid < FM_TOK ? red[8 * FM_TOK + tid] : -INFINITY;
      m = wave_max(m);
      if (tid == 0) partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = m;
    }
    return;
  }
  const float stab = (MODE == 0) ? red[8 * FM_TOK + i] : gmax[0];
  const int tp = M + 1;
#pragma unroll
  for (int f = 0; f < FM_FPT; ++f) {
    const int j = g + 8 * f;
    if (j < M) tile[i * tp + j] = ratio * (expf(s[f] - diag - stab) + eps);
  }
  __syncthreads();
  // coalesced write-back of the [32][M] tile
  const int rows = min(FM_TOK, N - n0);
  float* ob = out + ((size_t)bh * N + n0) * M;
  for (int c = tid; c < rows * M; c += 256) {
    const int r = c / M, j = c - r * M;
    ob[c] = tile[r * tp + j];
  }
}

__global__ __launch_bounds__(256) void max_reduce_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
  __shared__ float wm[4];
  float m = -INFINITY;
  for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, part[i]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
}

// ---- linear attention --------------------------------------------------------------------------
constexpr int LA_KT = 16;      // tokens per staged tile of the context pass
constexpr int LA_MAXM = 320;
constexpr int LA_FPW = LA_MAXM / 4;   // features per wave (contiguous range)

// part[bh][chunk] = { ctx[M][64] = sum_n k'[n,m] v[n,e] , ksum[M] = sum_n k'[n,m] } over the chunk's tokens
__global__ __launch_bounds__(256) void la_ctx_kernel(const float* __restrict__ kf, const float* __restrict__ v, int N,
                                                     int M, int chunk_len, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* kt = (float*)smem;                 // [LA_KT][M]
  float* vt = kt + LA_KT * M;               // [LA_KT][64]
  const int tid = threadIdx.x, e = tid & 63, w = tid >> 6;
  const int bh = blockIdx.y, n_lo = blockIdx.x * chunk_len, n_hi = min(N, n_lo + chunk_len);
  const int fpw = (M + 3) / 4, j0 = w * fpw, j1 = min(M, j0 + fpw);
  float acc[LA_FPW];
#pragma unroll
  for (int f = 0; f < LA_FPW; ++f) acc[f] = 0.f;
  float zs0 = 0.f, zs1 = 0.f;               // ksum of features tid and tid + 256
  for (int t0 = n_lo; t0 < n_hi; t0 += LA_KT) {
    const int rows = min(LA_KT, n_hi - t0);
    __syncthreads();
    const float* ks = kf + ((size_t)bh * N + t0) * M;
    for (int c = tid; c < LA_KT * M; c += 256) kt[c] = c < rows * M ? ks[c] : 0.f;
    const float* vs = v + ((size_t)bh * N + t0) * DH;
    for (int c = tid; c < LA_KT * DH; c += 256) vt[c] = c < rows * DH ? vs[c] : 0.f;
    __syncthreads();
    for (int r = 0; r < LA_KT; ++r) {
      const float vv = vt[r * DH + e];
      const float* kr = kt + r * M;
#pragma unroll
      for (int f = 0; f < LA_FPW; ++f)
        if (j0 + f < j1) acc[f] = fmaf(kr[j0 + f], vv, acc[f]);
    }
    if (tid < M) {
      float z = 0.f;
      for (int r = 0; r < LA_KT; ++r) z += kt[r * M + tid];
      zs0 += z;
    }
    if (tid + 256 < M) {
      float z = 0.f;
      for (int r = 0; r < LA_KT; ++r) z += kt[r * M + tid + 256];
      zs1 += z;
    }
  }
  float* rec = part + ((size_t)bh * gridDim.x + blockIdx.x) * ((size_t)M * DH + M);
#pragma unroll
  for (int f = 0; f < LA_FPW; ++f)
    if (j0 + f < j1) rec[(size_t)(j0 + f) * DH + e] = acc[f];
  if (tid < M) rec[(size_t)M * DH + tid] = zs0;
  if (tid + 256 < M) rec[(size_t)M * DH + tid + 256] = zs1;
}

// ctx[bh] = sum over chunks of part[bh][chunk] (fixed order)
__global__ void la_reduce_kernel(const float* __restrict__ part, int nchunks, int rec, float* __restrict__ ctx) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x, bh = blockIdx.y;
  if (idx >= rec) return;
  const float* p = part + (size_t)bh * nchunks * rec + idx;
  float s = 0.f;
  for (int c = 0; c < nchunks; ++c) s += p[(size_t)c * rec];
  ctx[(size_t)bh * rec + idx] = s;
}

// out[n,:] = (q'[n,:] ctx) / (q'[n,:] . ksum)
__global__ __launch_bounds__(256) void la_out_kernel(const float* __restrict__ qf, const float* __restrict__ ctx, int N,
                                                     int M, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* cs = (float*)smem;                 // [M][64] context, then [M] ksum
  float* qt = cs + (size_t)M * DH + M;      // [32][M]
  const int tid = threadIdx.x, e = tid & 63, w = tid >> 6;
  const int bh = blockIdx.y, n0 = blockIdx.x * 32;
  const int rows = min(32, N - n0);
  const float* cb = ctx + (size_t)bh * ((size_t)M * DH + M);
  for (int c = tid; c < M * DH + M; c += 256) cs[c] = cb[c];
  const float* qs = qf + ((size_t)bh * N + n0) * M;
  for (int c = tid; c < 32 * M; c += 256) qt[c] = c < rows * M ? qs[c] : 0.f;
  __syncthreads();
  const float* ksum = cs + (size_t)M * DH;
  float acc[8], den[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) { acc[r] = 0.f; den[r] = 0.f; }
  for (int j = 0; j < M; ++j) {
    const float c = cs[j * DH + e];
#pragma unroll
    for (int r = 0; r < 8; ++r) acc[r] = fmaf(qt[(8 * w + r) * M + j], c, acc[r]);
  }
  for (int j = e; j < M; j += 64) {
    const float z = ksum[j];
#pragma unroll
    for (int r = 0; r < 8; ++r) den[r] = fmaf(qt[(8 * w + r) * M + j], z, den[r]);
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const float d = wave_sum(den[r]);
    const int n = n0 + 8 * w + r;
    if (n < N) out[((size_t)bh * N + n) * DH + e] = acc[r] * (1.0f / d);
  }
}

// ------------------------------------------------------------------------------------------------
static int fm_blocks(int N) { return (N + FM_TOK - 1) / FM_TOK; }
long feature_map_ws_floats(long BH, int N) { return BH * fm_blocks(N) + 4; }

template <int MODE>
static hipError_t fm_launch(const float* data, const float* proj, long BH, int N, int M, float normalizer,
                            float norm_sq, float ratio, float eps, float* out, float* partial, const float* gmax,
                            hipStream_t s) {
  const size_t lds = ((size_t)M * FM_OP + 8 * FM_TOK + FM_TOK + (MODE == 1 ? 0 : (size_t)FM_TOK * (M + 1))) * sizeof(float);
  static std::atomic<unsigned long long> ready;   // LDS demand grows with M: the attribute is set for the largest M
  const size_t lds_max = ((size_t)FM_MAXM * FM_OP + 8 * FM_TOK + FM_TOK + (size_t)FM_TOK * (FM_MAXM + 1)) * sizeof(float);
  if (hipError_t e = ensure_dynamic_lds((const void*)feature_map_kernel<MODE>, lds_max, ready); e != hipSuccess) return e;
  hipLaunchKernelGGL((feature_map_kernel<MODE>), dim3(fm_blocks(N), (unsigned)BH), dim3(256), lds, s, data, proj, N, M,
                     normalizer, norm_sq, ratio, eps, out, partial, gmax);
  return hipGetLastError();
}

hipError_t feature_map(const float* data, const float* proj, long BH, int N, int M, int is_query, float normalizer,
                       float norm_sq, float ratio, float eps, float* out, float* ws, hipStream_t s) {
  Traced tr(K_FEATURE_MAP, 2.0 * BH * N * M * DH, s, (double)BH * N * (DH + M) * 4.0);
  if (is_query) return fm_launch<0>(data, proj, BH, N, M, normalizer, norm_sq, ratio, eps, out, nullptr, nullptr, s);
  const int nb = (int)(BH * fm_blocks(N));
  hipError_t e = fm_launch<1>(data, proj, BH, N, M, normalizer, norm_sq, ratio, eps, out, ws, nullptr, s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(max_reduce_kernel, dim3(1), dim3(256), 0, s, ws, nb, ws + nb);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  return fm_launch<2>(data, proj, BH, N, M, normalizer, norm_sq, ratio, eps, out, nullptr, ws + nb, s);
}

static int la_chunks(long BH, int N) {
  long c = 512 / (BH > 0 ? BH : 1);
  const long maxc = (N + LA_KT - 1) / LA_KT;
  if (c < 1) c = 1;
  if (c > maxc) c = maxc;
  return (int)c;
}
long linear_attention_ws_floats(long BH, int N, int M) {
  return BH * ((long)la_chunks(BH, N) + 1) * ((long)M * DH + M);
}

hipError_t linear_attention(const float* qf, const float* kf, const float* v, long BH, int N, int M, float* out,
                            float* ws, hipStream_t s) {
  Traced tr(K_LINEAR_ATTENTION, 4.0 * BH * N * M * DH + 2.0 * BH * N * M, s, (double)BH * N * (2.0 * M + 2.0 * DH) * 4.0);
  const int chunks = la_chunks(BH, N);
  const int chunk_len = ((N + chunks - 1) / chunks + LA_KT - 1) / LA_KT * LA_KT;
  const int rec = M * DH + M;
  float* ctx = ws + (size_t)BH * chunks * rec;
  const size_t lds1 = (size_t)LA_KT * (M + DH) * sizeof(float);
  hipLaunchKernelGGL(la_ctx_kernel, dim3(chunks, (unsigned)BH), dim3(256), lds1, s, kf, v, N, M, chunk_len, ws);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(la_reduce_kernel, dim3((rec + 255) / 256, (unsigned)BH), dim3(256), 0, s, ws, chunks, rec, ctx);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  const size_t lds2 = ((size_t)rec + 32 * (size_t)M) * sizeof(float);
  static std::atomic<unsigned long long> ready;
  const size_t lds2_max = ((size_t)LA_MAXM * DH + LA_MAXM + 32 * (size_t)LA_MAXM) * sizeof(float);
  e = ensure_dynamic_lds((const void*)la_out_kernel, lds2_max, ready);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(la_out_kernel, dim3((N + 31) / 32, (unsigned)BH), dim3(256), lds2, s, qf, ctx, N, M, out);
  return hipGetLastError();
}

}  // namespace scm
