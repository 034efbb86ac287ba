// Row-wise HBM-bound kernels of the Performer path: one 64-lane wave per token row of D<=256
// floats (D % 4 == 0, lane i owns columns 4i..4i+3), wave-shuffle reductions for the statistics.
//
//  embed_fwd / embed_bwd : token_emb gather + gene2vec positional add (performer_pytorch.py:604-607,444-446)
//  ln_fwd / ln_bwd       : nn.LayerNorm(dim) of PreLayerNorm (performer_pytorch.py:289-295) and its backward,
//                          fused with the residual-gradient add (reversible.py:144-145)
//  head_fwd / head_bwd   : final LayerNorm + output head (performer_pytorch.py:628-634 Linear(dim, num_tokens);
//                          rna_encoder.py:5-16 1x200 conv + ReLU) and backward
//  colsum_reduce         : sums per-workgroup partial rows into the f32 gradient
#include "scm_launch.hpp"
#include <mutex>

namespace scm {

constexpr float LN_EPS = 1e-5f;
constexpr int MAXV = 8;    // token vocabulary bound (num_tokens = 7)
constexpr int MAXC = 8;    // head classes bound

struct RowStats { float mean, rstd; };

SCM_DEV RowStats row_stats(const f32x4& x, bool active, int D) {
  float s = active ? (x[0] + x[1] + x[2] + x[3]) : 0.f;
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
  if (active) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float d = x[j] - mean; q += d * d; }
  }
  const float var = wave_sum(q) / (float)D;
  return {mean, rsqrtf(var + LN_EPS)};
}

// ---------------------------------------------------------------- embedding
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t* __restrict__ tok, const float* __restrict__ E,
                                                        const double* __restrict__ G, int B, int N, int D, int V,
                                                        float* __restrict__ out, int* __restrict__ err) {
  const int l = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const bool active = 4 * l < D;
  for (int n = wave; n < N; n += nwaves) {
    double g[4] = {0, 0, 0, 0};
    if (active && G) {
#pragma unroll
      for (int j = 0; j < 4; ++j) g[j] = G[(long)n * D + 4 * l + j];
    }
    for (int b = 0; b < B; ++b) {
      int64_t t = tok[(long)b * N + n];
      if (t < 0 || t >= V) { if (l == 0) atomicExch(err, 1); t = 0; }
      if (active) {
        f32x4 e = *(const f32x4*)(E + t * D + 4 * l);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (float)((double)e[j] + g[j]);
        *(f32x4*)(out + ((long)b * N + n) * D + 4 * l) = o;
      }
    }
  }
}

// partial[block][V][D] = per-token-id sums of dx rows handled by the block
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int64_t* __restrict__ tok, const float* __restrict__ dx,
                                                        long T, int D, int V, float* __restrict__ partial) {
  __shared__ float red[4][MAXV][256];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  const bool active = 4 * l < D;
  f32x4 acc[MAXV];
#pragma unroll
  for (int v = 0; v < MAXV; ++v) acc[v] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (long t = wave; t < T; t += nwaves) {   // no stores in the loop: unrolling keeps four rows' loads in flight
    const int id = (int)tok[t];
    f32x4 x = active ? *(const f32x4*)(dx + t * D + 4 * l) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int v = 0; v < MAXV; ++v) {
      const float m = (id == v) ? 1.f : 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[v][j] += m * x[j];
    }
  }
#pragma unroll
  for (int v = 0; v < MAXV; ++v)
#pragma unroll
    for (int j = 0; j < 4; ++j) red[w][v][4 * l + j] = acc[v][j];
  __syncthreads();
  for (int i = threadIdx.x; i < V * D; i += blockDim.x) {
    const int v = i / D, d = i % D;
    partial[(long)blockIdx.x * V * D + i] = red[0][v][d] + red[1][v][d] + red[2][v][d] + red[3][v][d];
  }
}

// ---------------------------------------------------------------- layer norm
// y bf16 = LN(x) * gamma + beta ; stats saved for the backward
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, long T, int D,
                                                     bf16* __restrict__ y, float* __restrict__ mean,
                                                     float* __restrict__ rstd) {
  const int l = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  const bool active = 4 * l < D;
  f32x4 g = f32x4{0, 0, 0, 0}, b = f32x4{0, 0, 0, 0};
  if (active) { g = *(const f32x4*)(gamma + 4 * l); b = *(const f32x4*)(beta + 4 * l); }
  for (long t = wave; t < T; t += nwaves) {
    f32x4 v = active ? *(const f32x4*)(x + t * D + 4 * l) : f32x4{0, 0, 0, 0};
    const RowStats st = row_stats(v, active, D);
    if (active) {
      bf16x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (bf16)((v[j] - st.mean) * st.rstd * g[j] + b[j]);
      *(bf16x4*)(y + t * D + 4 * l) = o;
    }
    if (l == 0) { mean[t] = st.mean; rstd[t] = st.rstd; }
  }
}

// dx = dres + LN'(dy) ; partial[block] = {sum dy*xhat [D], sum dy [D], sum dres [D]}
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const float* __restrict__ dres,
                                                     long T, int D, float* __restrict__ dx,
                                                     bf16* __restrict__ dx_b, float* __restrict__ partial) {
  __shared__ float red[4][3][256];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  const bool active = 4 * l < D;
  f32x4 g = f32x4{0, 0, 0, 0};
  if (active) g = *(const f32x4*)(gamma + 4 * l);
  f32x4 ag = f32x4{0, 0, 0, 0}, ab = ag, ar = ag;
  const float invD = 1.f / (float)D;
  for (long t = wave; t < T; t += nwaves) {
    f32x4 d = f32x4{0, 0, 0, 0}, v = d, r = d;
    if (active) {
      d = *(const f32x4*)(dy + t * D + 4 * l);
      v = *(const f32x4*)(x + t * D + 4 * l);
      if (dres) r = *(const f32x4*)(dres + t * D + 4 * l);
    }
    const float mu = mean[t], rs = rstd[t];
    f32x4 xh, gd;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      xh[j] = (v[j] - mu) * rs;
      gd[j] = d[j] * g[j];
      s1 += gd[j];
      s2 += gd[j] * xh[j];
    }
    if (!active) { s1 = 0.f; s2 = 0.f; }
    s1 = wave_sum(s1) * invD;
    s2 = wave_sum(s2) * invD;
    if (active) {
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[j] = r[j] + rs * (gd[j] - s1 - xh[j] * s2);
        ag[j] += d[j] * xh[j];
        ab[j] += d[j];
        ar[j] += r[j];
      }
      *(f32x4*)(dx + t * D + 4 * l) = o;
      if (dx_b) *(bf16x4*)(dx_b + t * D + 4 * l) = bf16x4{(bf16)o[0], (bf16)o[1], (bf16)o[2], (bf16)o[3]};
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    red[w][0][4 * l + j] = ag[j];
    red[w][1][4 * l + j] = ab[j];
    red[w][2][4 * l + j] = ar[j];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * D; i += blockDim.x) {
    const int k = i / D, dd = i % D;
    partial[(long)blockIdx.x * 3 * D + i] = red[0][k][dd] + red[1][k][dd] + red[2][k][dd] + red[3][k][dd];
  }
}

// ---------------------------------------------------------------- final norm + head
// enc (optional) f32 = LN(x); out[t][c] = enc . W[c] + b[c]  (relu: max(.,0))
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ W,
                                                       const float* __restrict__ bias, long T, int D, int C, int relu,
                                                       float* __restrict__ enc, float* __restrict__ out,
                                                       float* __restrict__ mean, float* __restrict__ rstd) {
  const int l = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  const bool active = 4 * l < D;
  f32x4 g = f32x4{0, 0, 0, 0}, b = g;
  f32x4 wv[MAXC];
#pragma unroll
  for (int c = 0; c < MAXC; ++c) wv[c] = f32x4{0, 0, 0, 0};
  if (active) {
    g = *(const f32x4*)(gamma + 4 * l);
    b = *(const f32x4*)(beta + 4 * l);
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C && W) wv[c] = *(const f32x4*)(W + (long)c * D + 4 * l);
  }
  for (long t = wave; t < T; t += nwaves) {
    f32x4 v = active ? *(const f32x4*)(x + t * D + 4 * l) : f32x4{0, 0, 0, 0};
    const RowStats st = row_stats(v, active, D);
    f32x4 y = f32x4{0, 0, 0, 0};
    if (active) {
#pragma unroll
      for (int j = 0; j < 4; ++j) y[j] = (v[j] - st.mean) * st.rstd * g[j] + b[j];
      if (enc) *(f32x4*)(enc + t * D + 4 * l) = y;
    }
    if (l == 0) { mean[t] = st.mean; rstd[t] = st.rstd; }
    if (out) {
#pragma unroll
      for (int c = 0; c < MAXC; ++c) {
        if (c < C) {
          float s = y[0] * wv[c][0] + y[1] * wv[c][1] + y[2] * wv[c][2] + y[3] * wv[c][3];
          s = wave_sum(s) + bias[c];
          if (relu) s = fmaxf(s, 0.f);
          if (l == 0) out[t * C + c] = s;
        }
      }
    }
  }
}

// dout [T,C] (+ denc [T,D] optional) -> dx [T,D]; partial[block] = {dgamma[D], dbeta[D], dW[C][D], db[C]}
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ denc,
                                                       const float* __restrict__ out, const float* __restrict__ x,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ W, long T, int D, int C, int relu,
                                                       float* __restrict__ dx, bf16* __restrict__ dx_b,
                                                       float* __restrict__ partial) {
  extern __shared__ float red[];  // [4][(2 + C) * 256 + 64]
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  const bool active = 4 * l < D;
  f32x4 g = f32x4{0, 0, 0, 0}, b = g;
  f32x4 wv[MAXC], aw[MAXC];
  float abias[MAXC];
#pragma unroll
  for (int c = 0; c < MAXC; ++c) { wv[c] = f32x4{0, 0, 0, 0}; aw[c] = wv[c]; abias[c] = 0.f; }
  if (active) {
    g = *(const f32x4*)(gamma + 4 * l);
    b = *(const f32x4*)(beta + 4 * l);
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C && W) wv[c] = *(const f32x4*)(W + (long)c * D + 4 * l);
  }
  f32x4 ag = f32x4{0, 0, 0, 0}, ab = ag;
  const float invD = 1.f / (float)D;
  // what a row needs from memory, fetched one row ahead of its use (a wave walks about sixty rows one after the other:
  // without this every row costs a full memory round trip)
  struct RowIn { f32x4 v, d; float mu, rs; float dl[MAXC]; };
  auto fetch = [&](long t) {
    RowIn in;
    in.v = in.d = f32x4{0, 0, 0, 0};
    in.mu = in.rs = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) in.dl[c] = 0.f;
    if (t < T) {
      if (active) in.v = *(const f32x4*)(x + t * D + 4 * l);
      in.mu = mean[t];
      in.rs = rstd[t];
      if (denc && active) in.d = *(const f32x4*)(denc + t * D + 4 * l);
      if (dout) {
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
          if (c < C) {
            float dl = dout[t * C + c];
            if (relu && !(out[t * C + c] > 0.f)) dl = 0.f;
            in.dl[c] = dl;
          }
        }
      }
    }
    return in;
  };
  RowIn next = fetch(wave);
  for (long t = wave; t < T; t += nwaves) {
    const RowIn in = next;
    next = fetch(t + nwaves);
    const f32x4 v = in.v;
    const float mu = in.mu, rs = in.rs;
    f32x4 xh, y, d = in.d;
#pragma unroll
    for (int j = 0; j < 4; ++j) { xh[j] = (v[j] - mu) * rs; y[j] = xh[j] * g[j] + b[j]; }
    if (dout) {
#pragma unroll
      for (int c = 0; c < MAXC; ++c) {
        if (c < C) {
          const float dl = in.dl[c];
#pragma unroll
          for (int j = 0; j < 4; ++j) { d[j] += dl * wv[c][j]; aw[c][j] += dl * y[j]; }
          abias[c] += dl;
        }
      }
    }
    f32x4 gd;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { gd[j] = d[j] * g[j]; s1 += gd[j]; s2 += gd[j] * xh[j]; }
    if (!active) { s1 = 0.f; s2 = 0.f; }
    s1 = wave_sum(s1) * invD;
    s2 = wave_sum(s2) * invD;
    if (active) {
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[j] = rs * (gd[j] - s1 - xh[j] * s2);
        ag[j] += d[j] * xh[j];
        ab[j] += d[j];
      }
      *(f32x4*)(dx + t * D + 4 * l) = o;
      if (dx_b) *(bf16x4*)(dx_b + t * D + 4 * l) = bf16x4{(bf16)o[0], (bf16)o[1], (bf16)o[2], (bf16)o[3]};
    }
  }
  const int stride = (2 + C) * 256 + 64;
  float* mine = red + w * stride;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    mine[4 * l + j] = active ? ag[j] : 0.f;
    mine[256 + 4 * l + j] = active ? ab[j] : 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) mine[(2 + c) * 256 + 4 * l + j] = active ? aw[c][j] : 0.f;
  }
  if (l == 0) {
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) mine[(2 + C) * 256 + c] = abias[c];
  }
  __syncthreads();
  // partial row layout: dgamma[D] | dbeta[D] | dW[C*D] | db[C]
  const int width = 2 * D + C * D + C;
  for (int i = threadIdx.x; i < width; i += blockDim.x) {
    int src;
    if (i < 2 * D) src = (i / D) * 256 + (i % D);
    else if (i < 2 * D + C * D) { const int k = i - 2 * D; src = (2 + k / D) * 256 + (k % D); }
    else src = (2 + C) * 256 + (i - 2 * D - C * D);
    partial[(long)blockIdx.x * width + i] = red[src] + red[stride + src] + red[2 * stride + src] + red[3 * stride + src];
  }
}

// Column sums of the per-workgroup partial rows, for up to 4 column segments with their own outputs:
// seg.out[k][i] (+)= sum_b partial[b][seg.lo[k] + i], i < seg.hi[k] - seg.lo[k].  One launch covers all segments
// (blockIdx.y = segment).  Workgroup = 32 columns x 8 row groups; fixed summation order.
__global__ __launch_bounds__(1024) void colsum_reduce_kernel(const float* __restrict__ partial, int nblocks, int width,
                                                             ColsumSegs seg, int accumulate) {
  // 32 columns x 32 row groups per workgroup: every thread sums nblocks/32 rows with 4 loads in flight, so the
  // dependent-latency chain is a handful of round trips
  __shared__ float red[32][33];
  const int k = blockIdx.y;
  const int lo = seg.lo[k], n = seg.hi[k] - lo;
  float* __restrict__ out = seg.out[k];
  const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + c;
  if (blockIdx.x * 32 >= n || out == nullptr) return;   // uniform per workgroup
  float s = 0.f;
  if (i < n) {
    const float* p = partial + lo + i;
    int b = rg;
    for (; b + 96 < nblocks; b += 128) {
      const float a0 = p[(long)b * width], a1 = p[(long)(b + 32) * width];
      const float a2 = p[(long)(b + 64) * width], a3 = p[(long)(b + 96) * width];
      s += (a0 + a1) + (a2 + a3);
    }
    for (; b < nblocks; b += 32) s += p[(long)b * width];
  }
  red[rg][c] = s;
  __syncthreads();
  if (rg == 0 && i < n) {
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll
    for (int r = 0; r < 32; r += 4) { t0 += red[r][c]; t1 += red[r + 1][c]; t2 += red[r + 2][c]; t3 += red[r + 3][c]; }
    const float t = (t0 + t1) + (t2 + t3);
    out[i] = accumulate ? out[i] + t : t;
  }
}

// Every small reduction of a backward pass in one launch (blockIdx.y = job).  Each element is summed exactly as
// wgrad_reduce_kernel / colsum_reduce_kernel sum it, so deferring a reduction does not change a bit of the result.
struct ReduceJobs { ReduceJob job[RB_MAX]; };
__global__ __launch_bounds__(1024) void reduce_batch_kernel(ReduceJobs jobs) {
  __shared__ float red[32][33];
  const ReduceJob& j = jobs.job[blockIdx.y];
  if (j.kind == 0) {
    const int P = j.P, Q = j.Q, S = j.S;
    float* __restrict__ grad = j.out[0];
    float* __restrict__ bias = j.out[1];
    const int qn = bias ? Q + 1 : Q;
    const long st = (long)P * W_LDSLAB;
    for (int idx = blockIdx.x * 1024 + threadIdx.x; idx < P * qn; idx += gridDim.x * 1024) {
      const int p = idx / qn, q = idx % qn;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      const float* sp = j.src + (long)p * W_LDSLAB + q;
      int i = 0;
      for (; i + 3 < S; i += 4) {
        s0 += sp[i * st]; s1 += sp[(i + 1) * st]; s2 += sp[(i + 2) * st]; s3 += sp[(i + 3) * st];
      }
      for (; i < S; ++i) s0 += sp[i * st];
      const float s = (s0 + s1) + (s2 + s3);
      if (q == Q) {
        bias[p] = j.accumulate ? bias[p] + s : s;
      } else {
        const long o = j.trans ? (long)q * P + p : (long)p * Q + q;
        grad[o] = j.accumulate ? grad[o] + s : s;
      }
    }
    return;
  }
  // column partials: workgroup = 32 columns x 32 row groups of one segment
  int k = 0, cb = blockIdx.x;
  for (; k < 4; ++k) {
    const int nb = j.out[k] ? (j.hi[k] - j.lo[k] + 31) / 32 : 0;
    if (cb < nb) break;
    cb -= nb;
  }
  if (k == 4) return;
  const int lo = j.lo[k], n = j.hi[k] - lo, nblocks = j.S, width = j.P;
  float* __restrict__ out = j.out[k];
  const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int i = cb * 32 + c;
  float s = 0.f;
  if (i < n) {
    const float* p = j.src + lo + i;
    int b = rg;
    for (; b + 96 < nblocks; b += 128) {
      const float a0 = p[(long)b * width], a1 = p[(long)(b + 32) * width];
      const float a2 = p[(long)(b + 64) * width], a3 = p[(long)(b + 96) * width];
      s += (a0 + a1) + (a2 + a3);
    }
    for (; b < nblocks; b += 32) s += p[(long)b * width];
  }
  red[rg][c] = s;
  __syncthreads();
  if (rg == 0 && i < n) {
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll
    for (int r = 0; r < 32; r += 4) { t0 += red[r][c]; t1 += red[r + 1][c]; t2 += red[r + 2][c]; t3 += red[r + 3][c]; }
    const float t = (t0 + t1) + (t2 + t3);
    out[i] = j.accumulate ? out[i] + t : t;
  }
}

// ---------------------------------------------------------------- host launchers
hipError_t reduce_batch(const ReduceJob* jobs, int n, hipStream_t s) {
  for (int at = 0; at < n; at += RB_MAX) {
    const int m = n - at < RB_MAX ? n - at : RB_MAX;
    ReduceJobs pack{};
    int blocks = 1;
    for (int i = 0; i < m; ++i) {
      pack.job[i] = jobs[at + i];
      const ReduceJob& j = pack.job[i];
      int need = 0;
      if (j.kind == 0) {
        need = (j.P * (j.out[1] ? j.Q + 1 : j.Q) + 1023) / 1024;
      } else {
        for (int k = 0; k < 4; ++k)
          if (j.out[k]) need += (j.hi[k] - j.lo[k] + 31) / 32;
      }
      if (need > blocks) blocks = need;
    }
    Traced tr(K_REDUCE_BATCH, 0, s);
    hipLaunchKernelGGL(reduce_batch_kernel, dim3(blocks, m), dim3(1024), 0, s, pack);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
  }
  return hipSuccess;
}

static int row_grid(long T) {
  long waves = (T + 3) / 4;  // 4 waves per block
  return (int)(waves < 2048 ? (waves < 1 ? 1 : waves) : 2048);
}

hipError_t embed_fwd(const int64_t* tok, const float* E, const double* G, int B, int N, int D, int V, float* out,
                     int* err, hipStream_t s) {
  Traced tr(K_EMBED_FWD, 0, s);
  hipLaunchKernelGGL(embed_fwd_kernel, dim3(row_grid(N)), dim3(256), 0, s, tok, E, G, B, N, D, V, out, err);
  return hipGetLastError();
}
int embed_bwd_blocks(long T) { long b = (T + 255) / 256; return (int)(b < 512 ? (b < 1 ? 1 : b) : 512); }
hipError_t embed_bwd(const int64_t* tok, const float* dx, long T, int D, int V, float* partial, hipStream_t s) {
  Traced tr(K_EMBED_BWD, 0, s);
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(embed_bwd_blocks(T)), dim3(256), 0, s, tok, dx, T, D, V, partial);
  return hipGetLastError();
}
hipError_t ln_fwd(const float* x, const float* g, const float* b, long T, int D, void* y, float* mean, float* rstd,
                  hipStream_t s) {
  Traced tr(K_LN_FWD, 0, s, (double)T * D * 6);
  hipLaunchKernelGGL(ln_fwd_kernel, dim3(row_grid(T)), dim3(256), 0, s, x, g, b, T, D, (bf16*)y, mean, rstd);
  return hipGetLastError();
}
int ln_bwd_blocks(long T) { long b = (T + 63) / 64; return (int)(b < 1024 ? (b < 1 ? 1 : b) : 1024); }
hipError_t ln_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                  const float* dres, long T, int D, float* dx, void* dx_b, float* partial, hipStream_t s) {
  Traced tr(K_LN_BWD, 0, s, (double)T * D * (8 + (dres ? 4 : 0) + 4 + (dx_b ? 2 : 0)));
  hipLaunchKernelGGL(ln_bwd_kernel, dim3(ln_bwd_blocks(T)), dim3(256), 0, s, dy, x, mean, rstd, gamma, dres, T, D, dx,
                     (bf16*)dx_b, partial);
  return hipGetLastError();
}
// ---------------------------------------------------------------- LayerNorm backward without a caller workspace
// The per-workgroup column partials of ln_bwd (at most 1024 workgroups x 3 x D floats, D <= 256) in a fixed array of
// the code object instead of a caller buffer: 3 MiB per GPU, held for the life of the process.  On the host the
// array's name is only a handle: its device address differs per GPU and is looked up once per device.
static __device__ __align__(256) float ln_bwd_scratch[LN_STATIC_SCRATCH_FLOATS];

hipError_t ln_bwd_static_scratch(float** out) {
  static float* addr[64] = {};                // per device; written once, under the mutex
  static std::mutex mu;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> lock(mu);
  if (!addr[dev]) {
    float* p = nullptr;
    e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(ln_bwd_scratch));
    if (e != hipSuccess) return e;
    if (!p) return hipErrorInvalidSymbol;
    addr[dev] = p;
  }
  *out = addr[dev];
  return hipSuccess;
}

hipError_t head_fwd(const float* x, const float* g, const float* b, const float* W, const float* bias, long T, int D,
                    int C, int relu, float* enc, float* out, float* mean, float* rstd, hipStream_t s) {
  Traced tr(K_HEAD_FWD, 0, s);
  hipLaunchKernelGGL(head_fwd_kernel, dim3(row_grid(T)), dim3(256), 0, s, x, g, b, W, bias, T, D, C, relu, enc, out,
                     mean, rstd);
  return hipGetLastError();
}
hipError_t head_bwd(const float* dout, const float* denc, const float* out, const float* x, const float* mean,
                    const float* rstd, const float* gamma, const float* beta, const float* W, long T, int D, int C,
                    int relu, float* dx, void* dx_b, float* partial, hipStream_t s) {
  size_t lds = (size_t)4 * ((2 + C) * 256 + 64) * sizeof(float);
  Traced tr(K_HEAD_BWD, 0, s);
  hipLaunchKernelGGL(head_bwd_kernel, dim3(ln_bwd_blocks(T)), dim3(256), lds, s, dout, denc, out, x, mean, rstd, gamma,
                     beta, W, T, D, C, relu, dx, (bf16*)dx_b, partial);
  return hipGetLastError();
}
hipError_t colsum_reduce(const float* partial, int nblocks, int width, const ColsumSegs& seg, int nseg, int accumulate,
                         hipStream_t s) {
  int widest = 0;
  for (int k = 0; k < nseg; ++k)
    if (seg.out[k] && seg.hi[k] - seg.lo[k] > widest) widest = seg.hi[k] - seg.lo[k];
  if (widest == 0) return hipSuccess;
  Traced tr(K_COLSUM, 0, s);
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((widest + 31) / 32, nseg), dim3(1024), 0, s, partial, nblocks, width,
                     seg, accumulate);
  return hipGetLastError();
}
hipError_t colsum_reduce(const float* partial, int nblocks, int width, int lo, int hi, float* out, int accumulate,
                         hipStream_t s) {
  ColsumSegs seg{{out, nullptr, nullptr, nullptr}, {lo, 0, 0, 0}, {hi, 0, 0, 0}};
  return colsum_reduce(partial, nblocks, width, seg, 1, accumulate, s);
}

}  // namespace scm
