// The tail of scMomer.get_cell_embedding after the encoder (scmomer.py:96-113,124-130 with the student encoder of
// extract_cell_emb.py:27-40), forward only, f32:
//   f    = genes @ Wp^T                         (rna_projection, no bias)        [B, P]
//   f    = f / |f|                              (config.normalize)
//   r    = LayerNorm_P(f)                       (rna_ln)
//   h    = x @ W1^T + b1                        (student encoder, first Linear)   [B, Hd]
//   a    = gelu_erf(LayerNorm_Hd(h)) @ W2^T + b2    (LayerNorm, GELU, dropout off in eval, second Linear)  [B, P]
//   out  = cat(a, r) @ Wo^T + bo                (to_out)                          [B, O]
// Two launches: the two G-wide products (G = genes per cell, 16,906) share one split-K kernel -- every workgroup
// owns a slice of the gene axis and all P + Hd output columns, stages its weight slice through LDS and keeps the
// sums of up to 32 cells in registers -- then one workgroup per cell adds the slice partials in a fixed order and
// runs the narrow rest.
#include "scm_launch.hpp"

namespace scm {

SCM_DEV float gelu_erf_exact(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }   // nn.GELU()

constexpr int CT_KS = 32;            // gene columns per staged slice
constexpr int CT_BG = 32;            // most cells per workgroup (accumulators per thread)
constexpr int CT_PITCH = CT_KS + 1;  // LDS row pitch in floats: threads of consecutive rows hit distinct banks
constexpr int CT_IV = 8;             // cell values a thread fetches per batch

// grid (gene chunks, cell groups of BG); block = P + Hd threads (one output column each, <= 1024).
// partial[chunk][b][col] = sum over the chunk's genes of in[b][g] * W[col][g], in = genes for col < P, x for the rest
// MAXT: the block size the register budget is sized for (512: 256 registers per thread, room for a whole slice's
// loads in flight; 1024: 128)
template <int BG, int MAXT>
__global__ __launch_bounds__(MAXT) void cell_tail_products_kernel(const float* __restrict__ genes,
                                                                  const float* __restrict__ x, int B, int G,
                                                                  const float* __restrict__ Wp, int P,
                                                                  const float* __restrict__ W1, int Hd, int chunk_len,
                                                                  float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* wl = (float*)smem;                              // [P + Hd][CT_PITCH]
  float* il = wl + (P + Hd) * CT_PITCH;                  // [2][BG][CT_KS]: genes rows, then x rows
  const int tid = threadIdx.x, cols = P + Hd;
  const int g_lo = blockIdx.x * chunk_len, g_hi = min(G, g_lo + chunk_len);
  const int b0 = blockIdx.y * BG, nb = min(BG, B - b0);
  const bool is_p = tid < P;
  constexpr int IN_TOTAL = 2 * BG * CT_KS;
  float acc[BG];
#pragma unroll
  for (int b = 0; b < BG; ++b) acc[b] = 0.f;
  for (int g0 = g_lo; g0 < g_hi; g0 += CT_KS) {
    // stage: weight slice (cols x 32, each row's 128 bytes contiguous in memory) and the cells' slices; the loads
    // of a thread are issued together, before the first LDS write waits for one (blockDim.x == cols: exactly CT_KS
    // weights per thread)
    {
      float wv[CT_KS];
#pragma unroll
      for (int k = 0; k < CT_KS; ++k) {
        const int i = tid + k * cols, r = i / CT_KS, c = i % CT_KS, g = g0 + c;
        const float* src = r < P ? Wp + (long)r * G : W1 + (long)(r - P) * G;
        wv[k] = g < g_hi ? src[g] : 0.f;
      }
      for (int base = 0; base < IN_TOTAL; base += CT_IV * cols) {
        float iv[CT_IV];
#pragma unroll
        for (int k = 0; k < CT_IV; ++k) {
          const int i = base + tid + k * cols;
          iv[k] = 0.f;
          if (i < IN_TOTAL) {
            const int which = i / (BG * CT_KS), j = i % (BG * CT_KS), b = j / CT_KS, c = j % CT_KS, g = g0 + c;
            const float* src = which == 0 ? genes : x;
            if (b < nb && g < g_hi) iv[k] = src[(long)(b0 + b) * G + g];
          }
        }
#pragma unroll
        for (int k = 0; k < CT_IV; ++k) {
          const int i = base + tid + k * cols;
          if (i < IN_TOTAL) il[i] = iv[k];
        }
      }
#pragma unroll
      for (int k = 0; k < CT_KS; ++k) {
        const int i = tid + k * cols;
        wl[(i / CT_KS) * CT_PITCH + i % CT_KS] = wv[k];
      }
    }
    __syncthreads();
    float wr[CT_KS];
#pragma unroll
    for (int c = 0; c < CT_KS; ++c) wr[c] = wl[tid * CT_PITCH + c];
    const float* in = il + (is_p ? 0 : BG * CT_KS);
#pragma unroll
    for (int b = 0; b < BG; ++b) {
      float s = acc[b];
#pragma unroll
      for (int c = 0; c < CT_KS; ++c) s = fmaf(in[b * CT_KS + c], wr[c], s);   // broadcast reads
      acc[b] = s;
    }
    __syncthreads();
  }
  float* out = partial + ((long)blockIdx.x * B + b0) * cols + tid;
#pragma unroll
  for (int b = 0; b < BG; ++b)
    if (b < nb) out[(long)b * cols] = acc[b];
}

// sum over a row of n values held one per thread (n <= blockDim.x, a multiple of 64), result to every thread
SCM_DEV float block_sum(float v, float* red, int nwaves) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
  for (int i = 0; i < nwaves; ++i) s += red[i];
  return s;
}

// one workgroup per cell, blockDim = max(P, Hd, O) rounded up to 64 (<= 1024)
__global__ __launch_bounds__(1024) void cell_tail_finish_kernel(const float* __restrict__ partial, int nchunks, int B,
                                                                CellTailArgs a, float* __restrict__ out) {
  __shared__ float red[16];
  __shared__ float cat[2048];                     // [P (student branch) | P (encoder branch)], P <= 1024
  __shared__ float act[1024];                     // gelu(LN(h)), Hd <= 1024
  const int tid = threadIdx.x, b = blockIdx.x, cols = a.P + a.Hd, nw = blockDim.x >> 6;
  // encoder branch: projection -> L2 normalisation -> LayerNorm
  float f = 0.f;
  if (tid < a.P) {
#pragma unroll 8
    for (int c = 0; c < nchunks; ++c) f += partial[((long)c * B + b) * cols + tid];
  }
  if (a.normalize) {
    const float n2 = block_sum(tid < a.P ? f * f : 0.f, red, nw);
    f = f / sqrtf(n2);
  }
  {
    const float mu = block_sum(tid < a.P ? f : 0.f, red, nw) / (float)a.P;
    const float d = tid < a.P ? f - mu : 0.f;
    const float var = block_sum(d * d, red, nw) / (float)a.P;
    if (tid < a.P) cat[a.P + tid] = d * rsqrtf(var + 1e-5f) * a.ln_p_w[tid] + a.ln_p_b[tid];
  }
  // student branch: first Linear (partials + bias) -> LayerNorm -> GELU
  float h = 0.f;
  if (tid < a.Hd) {
#pragma unroll 8
    for (int c = 0; c < nchunks; ++c) h += partial[((long)c * B + b) * cols + a.P + tid];
    h += a.b1[tid];
  }
  {
    const float mu = block_sum(tid < a.Hd ? h : 0.f, red, nw) / (float)a.Hd;
    const float d = tid < a.Hd ? h - mu : 0.f;
    const float var = block_sum(d * d, red, nw) / (float)a.Hd;
    if (tid < a.Hd) act[tid] = gelu_erf_exact(d * rsqrtf(var + 1e-5f) * a.ln_h_w[tid] + a.ln_h_b[tid]);
  }
  __syncthreads();
  if (tid < a.P) {                                // second Linear of the student encoder
    float s = a.b2[tid];
    const float* w = a.W2 + (long)tid * a.Hd;
    for (int k = 0; k < a.Hd; ++k) s = fmaf(act[k], w[k], s);
    cat[tid] = s;
  }
  __syncthreads();
  if (tid < a.O) {                                // to_out on cat(student, encoder)
    float s = a.bo[tid];
    const float* w = a.Wo + (long)tid * 2 * a.P;
    for (int k = 0; k < 2 * a.P; ++k) s = fmaf(cat[k], w[k], s);
    out[(long)b * a.O + tid] = s;
  }
}

int cell_tail_chunks(int G) {
  // about one workgroup per CU and cell group: 256 gene chunks of whole 32-column slices
  const int slices = (G + CT_KS - 1) / CT_KS;
  const int per = (slices + 255) / 256;
  return (slices + per - 1) / per;
}
long cell_tail_ws_floats(int B, int G, int P, int Hd) { return (long)cell_tail_chunks(G) * B * (P + Hd); }

// the split-K products alone: ws[chunk][b][P + Hd]; *nchunks = the number of gene chunks the caller sums over
template <int BG, int MAXT>
static hipError_t launch_products(const float* genes, const float* x, int B, int G, const CellTailArgs& a, float* ws,
                                  int nch, int chunk_len, hipStream_t s) {
  const int cols = a.P + a.Hd;
  const size_t lds = ((size_t)cols * CT_PITCH + 2 * BG * CT_KS) * sizeof(float);
  static std::atomic<unsigned long long> ready;
  if (hipError_t e = ensure_dynamic_lds((const void*)(cell_tail_products_kernel<BG, MAXT>), lds, ready); e != hipSuccess) return e;
  hipLaunchKernelGGL((cell_tail_products_kernel<BG, MAXT>), dim3(nch, (B + BG - 1) / BG), dim3(cols), lds, s, genes, x, B, G,
                     a.Wp, a.P, a.W1, a.Hd, chunk_len, ws);
  return hipGetLastError();
}

// the split-K products alone: ws[chunk][b][P + Hd]; *nchunks = the number of gene chunks the caller sums over.
// A batch of at most 4 / 8 cells runs the kernel with that many accumulators per thread.
hipError_t cell_tail_products(const float* genes, const float* x, int B, int G, const CellTailArgs& a, float* ws,
                              int* nchunks, hipStream_t s) {
  const int cols = a.P + a.Hd;
  const int slices = (G + CT_KS - 1) / CT_KS, nch = cell_tail_chunks(G);
  const int chunk_len = ((slices + nch - 1) / nch) * CT_KS;
  const int groups = B <= 4 ? 1 : B <= 8 ? 1 : (B + CT_BG - 1) / CT_BG;
  Traced tr(K_CELL_TAIL, 2.0 * B * G * cols, s, (double)G * cols * 4 * groups + 2.0 * B * G * 4);
  *nchunks = nch;
  if (cols <= 512) {
    if (B <= 4) return launch_products<4, 512>(genes, x, B, G, a, ws, nch, chunk_len, s);
    if (B <= 8) return launch_products<8, 512>(genes, x, B, G, a, ws, nch, chunk_len, s);
    return launch_products<CT_BG, 512>(genes, x, B, G, a, ws, nch, chunk_len, s);
  }
  if (B <= 4) return launch_products<4, 1024>(genes, x, B, G, a, ws, nch, chunk_len, s);
  if (B <= 8) return launch_products<8, 1024>(genes, x, B, G, a, ws, nch, chunk_len, s);
  return launch_products<CT_BG, 1024>(genes, x, B, G, a, ws, nch, chunk_len, s);
}

hipError_t cell_tail(const float* genes, const float* x, int B, int G, CellTailArgs a, float* ws, float* out,
                     hipStream_t s) {
  int nch = 0;
  if (hipError_t e = cell_tail_products(genes, x, B, G, a, ws, &nch, s); e != hipSuccess) return e;
  int threads = a.P > a.Hd ? a.P : a.Hd;
  if (a.O > threads) threads = a.O;
  threads = (threads + 63) / 64 * 64;
  hipLaunchKernelGGL(cell_tail_finish_kernel, dim3(B), dim3(threads), 0, s, ws, nch, B, a, out);
  return hipGetLastError();
}

}  // namespace scm
