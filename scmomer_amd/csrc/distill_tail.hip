// The trainable tail of the S3 distillation step (s3_pretrain_missing.py:291-307 over scMomer.forward mode='one',
// reconstruct=True, scmomer.py:41-94, with the student encoder of extract_cell_emb.py:27-40 and the linear decoders of
// utils.py:168-179), forward with the three losses and backward to every trainable weight, f32:
//   r     = LayerNorm_P(normalised(genes @ Wp^T))                (frozen: rna_projection, rna_ln)
//   h     = x @ W1^T + b1 ; xh = (h - mean) rstd ; u = xh gamma + beta ; gd = gelu_erf(u) * drop
//   a     = gd @ W2^T + b2                                       (the student's embedding, "atac_embeds")
//   out   = cat(a, r) @ Wo^T + bo
//   z_a   = out @ Wa^T + ba   [B, Na] ;  z_r = out @ Wr^T + br   [B, G]
//   L_atac = mean BCE-with-logits(z_a, atac_raw > 0; pos_weight),  L_rna = mean (z_r - x)^2,
//   L_dist = mean (a - latent)^2
// Forward: the split-K product kernel of the cell-embedding tail, one workgroup per cell for the narrow middle, one
// pass over the decoder rows for the two reconstruction losses, one workgroup that adds the loss partials.
// Backward: a second pass over the decoder rows (the logits are formed again from `out`: 18.9 MB of weights, cheaper
// than keeping [B, Na + G] logits and reading them back) that writes the decoder gradients row by row and the
// partial d out of its rows; one workgroup per cell for the narrow middle; one launch for the small weight
// gradients; one for the [Hd, G] gradient of the first student layer (outer products with the expression values).
#include "scm_launch.hpp"

namespace scm {

SCM_DEV float dt_gelu(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }
SCM_DEV float dt_gelu_grad(float v) {
  return 0.5f * (1.f + erff(v * 0.70710678118654752f)) + v * 0.3989422804014327f * __expf(-0.5f * v * v);
}

// cells per loss group: the batch is group_cells-cell micro-batches stacked, each with its own loss means (0: one group)
__host__ __device__ inline int distill_group_cells(const DistillArgs& a) { return a.group_cells > 0 ? a.group_cells : a.B; }

constexpr int DT_O = 128;        // decoder input width (to_out: Linear(256, 128), scmomer.py:31): two floats per lane
constexpr int DT_ROWS = 64;      // decoder rows per workgroup (4 waves x 16 rows)
constexpr int DT_RPW = DT_ROWS / 4;

SCM_DEV float dt_block_sum(float v, float* red, int nwaves) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
  for (int i = 0; i < nwaves; ++i) s += red[i];
  return s;
}

// tape: what the backward reads again.  Offsets in floats.
struct Tape {
  long xh, gd, cat, out, rstd, total;
  __host__ __device__ Tape(int B, int P, int Hd, int O) {
    xh = 0; gd = xh + (long)B * Hd; cat = gd + (long)B * Hd; out = cat + (long)B * 2 * P; rstd = out + (long)B * O;
    total = rstd + B;
  }
};
// scratch: [products partials | loss partials (2 per decoder workgroup and loss group) | d out partials (per decoder workgroup)
//           | per-cell vectors of the backward: d out [B,O], d a [B,P], d u [B,Hd], d h [B,Hd]]
struct Scratch {
  long prod, loss, dout, v_dout, v_da, v_du, v_dh, total;
  int nwg;
  __host__ __device__ Scratch(int B, int G, int P, int Hd, int O, int Na) {
    nwg = (Na + G + DT_ROWS - 1) / DT_ROWS;
    prod = 0; loss = prod + cell_tail_ws_floats(B, G, P, Hd); dout = loss + 2L * nwg * B;   // room for one loss group per cell
    v_dout = dout + (long)nwg * B * O; v_da = v_dout + (long)B * O; v_du = v_da + (long)B * P;
    v_dh = v_du + (long)B * Hd; total = v_dh + (long)B * Hd;
  }
};

// ------------------------------------------------------------------------------------------------ forward, middle
// one workgroup per cell, blockDim = max(P, Hd, O) rounded up to 64 (<= 1024)
__global__ __launch_bounds__(1024) void distill_mid_fwd_kernel(const float* __restrict__ partial, int nchunks,
                                                               DistillArgs a, float* __restrict__ tape) {
  __shared__ float red[16];
  __shared__ float cat[2048];
  __shared__ float act[1024];
  const CellTailArgs& w = a.tail;
  const int tid = threadIdx.x, b = blockIdx.x, B = a.B, cols = w.P + w.Hd, nw = blockDim.x >> 6;
  const Tape tp(B, w.P, w.Hd, w.O);
  float f = 0.f;
  if (tid < w.P) {
#pragma unroll 8
    for (int c = 0; c < nchunks; ++c) f += partial[((long)c * B + b) * cols + tid];
  }
  if (w.normalize) {
    const float n2 = dt_block_sum(tid < w.P ? f * f : 0.f, red, nw);
    f = f / sqrtf(n2);
  }
  {
    const float mu = dt_block_sum(tid < w.P ? f : 0.f, red, nw) / (float)w.P;
    const float d = tid < w.P ? f - mu : 0.f;
    const float var = dt_block_sum(d * d, red, nw) / (float)w.P;
    if (tid < w.P) cat[w.P + tid] = d * rsqrtf(var + 1e-5f) * w.ln_p_w[tid] + w.ln_p_b[tid];
  }
  float h = 0.f;
  if (tid < w.Hd) {
#pragma unroll 8
    for (int c = 0; c < nchunks; ++c) h += partial[((long)c * B + b) * cols + w.P + tid];
    h += w.b1[tid];
  }
  {
    const float mu = dt_block_sum(tid < w.Hd ? h : 0.f, red, nw) / (float)w.Hd;
    const float d = tid < w.Hd ? h - mu : 0.f;
    const float var = dt_block_sum(d * d, red, nw) / (float)w.Hd;
    const float rstd = rsqrtf(var + 1e-5f);
    if (tid < w.Hd) {
      const float xh = d * rstd;
      float g = dt_gelu(xh * w.ln_h_w[tid] + w.ln_h_b[tid]);
      if (a.drop) g *= a.drop[(long)b * w.Hd + tid];
      act[tid] = g;
      tape[tp.xh + (long)b * w.Hd + tid] = xh;
      tape[tp.gd + (long)b * w.Hd + tid] = g;
    }
    if (tid == 0) tape[tp.rstd + b] = rstd;
  }
  __syncthreads();
  if (tid < w.P) {
    float s = w.b2[tid];
    const float* wr = w.W2 + (long)tid * w.Hd;
    for (int k = 0; k < w.Hd; ++k) s = fmaf(act[k], wr[k], s);
    cat[tid] = s;
  }
  __syncthreads();
  for (int k = tid; k < 2 * w.P; k += blockDim.x) tape[tp.cat + (long)b * 2 * w.P + k] = cat[k];
  if (tid < w.O) {
    float s = w.bo[tid];
    const float* wr = w.Wo + (long)tid * 2 * w.P;
    for (int k = 0; k < 2 * w.P; ++k) s = fmaf(cat[k], wr[k], s);
    tape[tp.out + (long)b * w.O + tid] = s;
  }
}

// ------------------------------------------------------------------------------------------------ decoder rows
// Row j of the stacked decoders: j < Na is peak j of ATAC.output_mean.0, else gene j - Na of RNA.output_mean.0.
// A wave owns 16 consecutive rows; lane l holds columns 2l, 2l+1 of the row and of every cell's `out`; after the
// wave sum lane b keeps cell b's logit and does that cell's loss term.
// BWD = 0: loss partials (sum over the workgroup's rows and the cells of a loss group) -> loss_partial[wg][group][2].
// BWD = 1: d z from the loss terms, the row's weight/bias gradient, and the workgroup's share of d out.
template <int BMAX, int BWD>
__global__ __launch_bounds__(256) void distill_decoder_kernel(DistillArgs a, const float* __restrict__ out,
                                                              float* __restrict__ loss_partial,
                                                              const float* __restrict__ gloss, DistillGrads gr,
                                                              int accumulate, float* __restrict__ dout_partial) {
  __shared__ f32x2 ol[BMAX][64];
  __shared__ f32x2 dsum[BWD ? BMAX : 1][64];
  __shared__ float cellsum[4][32][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, B = a.B, Na = a.Na, G = a.G;
  const int gc = distill_group_cells(a);
  for (int i = tid; i < BMAX * 64; i += 256) {
    const int b = i >> 6, l = i & 63;
    f32x2 v = {0.f, 0.f};
    if (b < B) v = *(const f32x2*)(out + (long)b * DT_O + 2 * l);
    ol[b][l] = v;
    if (BWD) dsum[b][l] = f32x2{0.f, 0.f};
  }
  __syncthreads();
  const int rows = Na + G;
  const int row0 = blockIdx.x * DT_ROWS + wave * DT_RPW;
  float la = 0.f, lr = 0.f;
  float g_atac = 0.f, g_rna = 0.f;
  if (BWD) {
    g_atac = gloss[0] / ((float)gc * (float)Na);
    g_rna = gloss[1] * 2.f / ((float)gc * (float)G);
  }
  f32x2 dacc[BMAX];
#pragma unroll
  for (int b = 0; b < BMAX; ++b) dacc[b] = f32x2{0.f, 0.f};
  // what a row needs from memory, fetched one row ahead of its use
  struct RowIn { f32x2 w; float bias, target, pw; };
  auto fetch = [&](int j) {
    RowIn in;
    const bool atac = j < Na;
    const int jj = atac ? j : j - Na;
    in.w = *(const f32x2*)((atac ? a.Wa : a.Wr) + (long)jj * DT_O + 2 * lane);
    in.bias = in.target = 0.f;
    in.pw = 1.f;
    if (lane < B) {
      in.bias = atac ? a.ba[jj] : a.br[jj];
      in.target = atac ? a.atac_raw[(long)lane * Na + jj] : a.x[(long)lane * G + jj];
      if (atac) in.pw = a.pos_weight[a.pos_weight_n == 1 ? 0 : jj];
    }
    return in;
  };
  const int nrows = min(DT_RPW, rows - row0);
  RowIn next{};
  if (nrows > 0) next = fetch(row0);
  for (int r = 0; r < nrows; ++r) {
    const int j = row0 + r;
    const RowIn in = next;
    if (r + 1 < nrows) next = fetch(j + 1);
    const bool atac = j < Na;
    const int jj = atac ? j : j - Na;
    const f32x2 w = in.w;
    float zmine = 0.f;
#pragma unroll
    for (int b = 0; b < BMAX; ++b) {
      if (b < B) {
        const f32x2 o = ol[b][lane];
        const float s = wave_sum(fmaf(w.x, o.x, w.y * o.y));
        if (lane == b) zmine = s;
      }
    }
    float dz = 0.f;
    if (lane < B) {
      const float z = zmine + in.bias;
      if (atac) {
        const float t = in.target > 0.f ? 1.f : 0.f;
        const float lw = 1.f + (in.pw - 1.f) * t;
        if (BWD) {
          // d/dz [(1 - t) z + lw softplus(-z)] = (1 - t) - lw sigmoid(-z)
          const float e = __expf(-fabsf(z));
          const float sig_neg = z >= 0.f ? e / (1.f + e) : 1.f / (1.f + e);
          dz = g_atac * ((1.f - t) - lw * sig_neg);
        } else {
          la += (1.f - t) * z + lw * (log1pf(expf(-fabsf(z))) + fmaxf(-z, 0.f));
        }
      } else {
        const float d = z - in.target;
        if (BWD) dz = g_rna * d; else lr += d * d;
      }
    }
    if (BWD) {
      f32x2 dw = {0.f, 0.f};
      float db = 0.f;
#pragma unroll
      for (int b = 0; b < BMAX; ++b) {
        if (b < B) {
          const float dzb = __shfl(dz, b);
          const f32x2 o = ol[b][lane];
          dw.x = fmaf(dzb, o.x, dw.x);
          dw.y = fmaf(dzb, o.y, dw.y);
          dacc[b].x = fmaf(dzb, w.x, dacc[b].x);
          dacc[b].y = fmaf(dzb, w.y, dacc[b].y);
          db += dzb;
        }
      }
      f32x2* dst = (f32x2*)((atac ? gr.dWa : gr.dWr) + (long)jj * DT_O + 2 * lane);
      if (accumulate) { const f32x2 old = *dst; dw.x += old.x; dw.y += old.y; }
      *dst = dw;
      if (lane == 0) {
        float* dbp = (atac ? gr.dba : gr.dbr) + jj;
        *dbp = accumulate ? *dbp + db : db;
      }
    }
  }
  if (BWD) {
    // the four waves add their shares of d out one after the other (a fixed order)
    for (int wv = 0; wv < 4; ++wv) {
      if (wave == wv) {
#pragma unroll
        for (int b = 0; b < BMAX; ++b)
          if (b < B) { f32x2 v = dsum[b][lane]; v.x += dacc[b].x; v.y += dacc[b].y; dsum[b][lane] = v; }
      }
      __syncthreads();
    }
    for (int i = tid; i < B * 64; i += 256) {
      const int b = i >> 6, l = i & 63;
      *(f32x2*)(dout_partial + ((long)blockIdx.x * B + b) * DT_O + 2 * l) = dsum[b][l];
    }
  } else {
    // lane b holds cell b's sums over this wave's rows: per loss group, the cells and waves in a fixed order
    if (lane < 32) { cellsum[wave][lane][0] = la; cellsum[wave][lane][1] = lr; }
    __syncthreads();
    const int ngroups = B / gc;
    if (tid < 2 * ngroups) {
      const int g = tid >> 1, which = tid & 1;
      float s = 0.f;
      for (int b = g * gc; b < (g + 1) * gc; ++b)
        s += (cellsum[0][b][which] + cellsum[1][b][which]) + (cellsum[2][b][which] + cellsum[3][b][which]);
      loss_partial[((long)blockIdx.x * ngroups + g) * 2 + which] = s;
    }
  }
}

// one workgroup of 256 per loss group: losses[group][0..3] = L_atac, L_rna, L_dist, L_atac + L_rna + distill_weight L_dist
__global__ __launch_bounds__(256) void distill_loss_kernel(const float* __restrict__ loss_partial, int nwg,
                                                           DistillArgs a, const float* __restrict__ tape,
                                                           float* __restrict__ losses) {
  __shared__ float red[16];
  const int tid = threadIdx.x, B = a.B, P = a.tail.P, g = blockIdx.x, gc = distill_group_cells(a), ngroups = B / gc;
  const Tape tp(B, P, a.tail.Hd, a.tail.O);
  float la = 0.f, lr = 0.f, ld = 0.f;
  for (int i = tid; i < nwg; i += 256) {
    la += loss_partial[((long)i * ngroups + g) * 2];
    lr += loss_partial[((long)i * ngroups + g) * 2 + 1];
  }
  for (int i = tid; i < gc * P; i += 256) {
    const int b = g * gc + i / P, k = i % P;
    const float d = tape[tp.cat + (long)b * 2 * P + k] - a.latent[(long)b * P + k];
    ld += d * d;
  }
  la = dt_block_sum(la, red, 4);
  lr = dt_block_sum(lr, red, 4);
  ld = dt_block_sum(ld, red, 4);
  if (tid == 0) {
    const float l0 = la / ((float)gc * (float)a.Na), l1 = lr / ((float)gc * (float)a.G), l2 = ld / ((float)gc * (float)P);
    losses[4 * g] = l0;
    losses[4 * g + 1] = l1;
    losses[4 * g + 2] = l2;
    losses[4 * g + 3] = (l0 + l1) + a.distill_weight * l2;
  }
}

// ------------------------------------------------------------------------------------------------ backward, middle
// one workgroup per cell: d out (sum of the decoder workgroups' shares) -> to_out -> d a (+ the distillation term)
// -> second student Linear -> dropout, GELU, LayerNorm -> d h.  Per-cell vectors go to scratch for the weight
// gradients, which sum over the cells.
__global__ __launch_bounds__(1024) void distill_mid_bwd_kernel(DistillArgs a, const float* __restrict__ tape,
                                                               const float* __restrict__ dout_partial, int nwg,
                                                               const float* __restrict__ gloss, float* __restrict__ ws,
                                                               long off_dout, long off_da, long off_du, long off_dh) {
  __shared__ float red[16];
  __shared__ float s_dout[1024];
  __shared__ float s_da[1024];
  const CellTailArgs& w = a.tail;
  const int tid = threadIdx.x, b = blockIdx.x, B = a.B, P = w.P, Hd = w.Hd, O = w.O, nw = blockDim.x >> 6;
  const Tape tp(B, P, Hd, O);
  {
    // d out = the sum of the decoder workgroups' shares: the block's threads split them into blockDim / O ranges
    // (loads of a range in flight together), the ranges are added in order
    const int parts = blockDim.x / O, part = tid / O, o = tid % O;          // O <= blockDim.x
    const int per = (nwg + parts - 1) / parts;
    float s = 0.f;
    if (part < parts) {
      const int hi = min(nwg, (part + 1) * per);
#pragma unroll 8
      for (int i = part * per; i < hi; ++i) s += dout_partial[((long)i * B + b) * O + o];
      s_da[tid] = s;
    }
    __syncthreads();
    if (tid < O) {
      float t = 0.f;
      for (int q = 0; q < parts; ++q) t += s_da[q * O + tid];
      s_dout[tid] = t;
      ws[off_dout + (long)b * O + tid] = t;
    }
  }
  __syncthreads();
  if (tid < P) {
    float s = 0.f;
    for (int o = 0; o < O; ++o) s = fmaf(s_dout[o], w.Wo[(long)o * 2 * P + tid], s);
    const float g_dist = gloss[2] * 2.f / ((float)distill_group_cells(a) * (float)P);
    s += g_dist * (tape[tp.cat + (long)b * 2 * P + tid] - a.latent[(long)b * P + tid]);
    s_da[tid] = s;
    ws[off_da + (long)b * P + tid] = s;
  }
  __syncthreads();
  float dxh = 0.f, xh = 0.f;
  if (tid < Hd) {
    float s = 0.f;
    for (int p = 0; p < P; ++p) s = fmaf(s_da[p], w.W2[(long)p * Hd + tid], s);
    xh = tape[tp.xh + (long)b * Hd + tid];
    const float gam = w.ln_h_w[tid];
    const float u = xh * gam + w.ln_h_b[tid];
    float du = s * dt_gelu_grad(u);
    if (a.drop) du *= a.drop[(long)b * Hd + tid];
    ws[off_du + (long)b * Hd + tid] = du;
    dxh = du * gam;
  }
  const float m1 = dt_block_sum(dxh, red, nw) / (float)Hd;
  const float m2 = dt_block_sum(dxh * xh, red, nw) / (float)Hd;
  if (tid < Hd) ws[off_dh + (long)b * Hd + tid] = tape[tp.rstd + b] * (dxh - m1 - xh * m2);
}

// the small weight gradients, one thread per element: d Wo [O, 2P], d W2 [P, Hd], d bo [O], d b2 [P],
// d gamma [Hd], d beta [Hd], d b1 [Hd]; each a sum over the cells
__global__ __launch_bounds__(256) void distill_small_grads_kernel(DistillArgs a, const float* __restrict__ tape,
                                                                  const float* __restrict__ ws, long off_dout,
                                                                  long off_da, long off_du, long off_dh,
                                                                  DistillGrads gr, int accumulate) {
  const int B = a.B, P = a.tail.P, Hd = a.tail.Hd, O = a.tail.O;
  const Tape tp(B, P, Hd, O);
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  float s = 0.f;
  float* dst = nullptr;
  if (i < (long)O * 2 * P) {
    const int o = i / (2 * P), k = i % (2 * P);
    for (int b = 0; b < B; ++b) s = fmaf(ws[off_dout + (long)b * O + o], tape[tp.cat + (long)b * 2 * P + k], s);
    dst = gr.dWo + i;
  } else if ((i -= (long)O * 2 * P) < (long)P * Hd) {
    const int p = i / Hd, k = i % Hd;
    for (int b = 0; b < B; ++b) s = fmaf(ws[off_da + (long)b * P + p], tape[tp.gd + (long)b * Hd + k], s);
    dst = gr.dW2 + i;
  } else if ((i -= (long)P * Hd) < O) {
    for (int b = 0; b < B; ++b) s += ws[off_dout + (long)b * O + i];
    dst = gr.dbo + i;
  } else if ((i -= O) < P) {
    for (int b = 0; b < B; ++b) s += ws[off_da + (long)b * P + i];
    dst = gr.db2 + i;
  } else if ((i -= P) < Hd) {
    for (int b = 0; b < B; ++b) s = fmaf(ws[off_du + (long)b * Hd + i], tape[tp.xh + (long)b * Hd + i], s);
    dst = gr.dln_w + i;
  } else if ((i -= Hd) < Hd) {
    for (int b = 0; b < B; ++b) s += ws[off_du + (long)b * Hd + i];
    dst = gr.dln_b + i;
  } else if ((i -= Hd) < Hd) {
    for (int b = 0; b < B; ++b) s += ws[off_dh + (long)b * Hd + i];
    dst = gr.db1 + i;
  }
  if (dst) *dst = accumulate ? *dst + s : s;
}

// d W1[k][g] = sum_b d h[b][k] x[b][g]: thread = gene column (the cells' values in registers), 32 rows per workgroup
constexpr int DT_W1_ROWS = 32;
template <int BMAX>
__global__ __launch_bounds__(256) void distill_w1_grad_kernel(const float* __restrict__ x, const float* __restrict__ dh,
                                                              int B, int G, int Hd, float* __restrict__ dW1,
                                                              int accumulate) {
  __shared__ float dl[BMAX][DT_W1_ROWS];
  const int tid = threadIdx.x, k0 = blockIdx.y * DT_W1_ROWS;
  for (int i = tid; i < BMAX * DT_W1_ROWS; i += 256) {
    const int b = i / DT_W1_ROWS, k = i % DT_W1_ROWS;
    dl[b][k] = (b < B && k0 + k < Hd) ? dh[(long)b * Hd + k0 + k] : 0.f;
  }
  __syncthreads();
  const int g = blockIdx.x * 256 + tid;
  if (g >= G) return;
  float xv[BMAX];
#pragma unroll
  for (int b = 0; b < BMAX; ++b) xv[b] = b < B ? x[(long)b * G + g] : 0.f;
#pragma unroll 8
  for (int k = 0; k < DT_W1_ROWS; ++k) {
    if (k0 + k >= Hd) break;
    float s = 0.f;
#pragma unroll
    for (int b = 0; b < BMAX; ++b) s = fmaf(dl[b][k], xv[b], s);
    float* dst = dW1 + (long)(k0 + k) * G + g;
    *dst = accumulate ? *dst + s : s;
  }
}

// ------------------------------------------------------------------------------------------------ host
long distill_tape_floats(int B, int P, int Hd, int O) { return Tape(B, P, Hd, O).total; }
long distill_ws_floats(int B, int G, int P, int Hd, int O, int Na) { return Scratch(B, G, P, Hd, O, Na).total; }
int distill_max_cells() { return 32; }

template <int BWD>
static void launch_decoder(const DistillArgs& a, const float* out, float* loss_partial, const float* gloss,
                           const DistillGrads& gr, int accumulate, float* dout_partial, int nwg, hipStream_t s) {
#define SCM_DT_CASE(BM)                                                                                            \
  hipLaunchKernelGGL((distill_decoder_kernel<BM, BWD>), dim3(nwg), dim3(256), 0, s, a, out, loss_partial, gloss, gr, \
                     accumulate, dout_partial)
  if (a.B <= 4) SCM_DT_CASE(4);
  else if (a.B <= 8) SCM_DT_CASE(8);
  else if (a.B <= 16) SCM_DT_CASE(16);
  else SCM_DT_CASE(32);
#undef SCM_DT_CASE
}

hipError_t distill_fwd(const DistillArgs& a, float* tape, float* ws, float* losses, hipStream_t s) {
  const CellTailArgs& w = a.tail;
  const Tape tp(a.B, w.P, w.Hd, w.O);
  const Scratch sc(a.B, a.G, w.P, w.Hd, w.O, a.Na);
  int nch = 0;
  if (hipError_t e = cell_tail_products(a.genes, a.x, a.B, a.G, w, ws + sc.prod, &nch, s); e != hipSuccess) return e;
  Traced tr(K_DISTILL_TAIL, 2.0 * a.B * (a.Na + a.G) * DT_O, s, 4.0 * (a.Na + a.G) * (DT_O + 1 + a.B));
  int threads = w.P > w.Hd ? w.P : w.Hd;
  if (w.O > threads) threads = w.O;
  threads = (threads + 63) / 64 * 64;
  hipLaunchKernelGGL(distill_mid_fwd_kernel, dim3(a.B), dim3(threads), 0, s, ws + sc.prod, nch, a, tape);
  launch_decoder<0>(a, tape + tp.out, ws + sc.loss, nullptr, DistillGrads{}, 0, nullptr, sc.nwg, s);
  hipLaunchKernelGGL(distill_loss_kernel, dim3(a.B / distill_group_cells(a)), dim3(256), 0, s, ws + sc.loss, sc.nwg, a, tape,
                     losses);
  return hipGetLastError();
}

hipError_t distill_bwd(const DistillArgs& a, const float* tape, float* ws, const float* gloss, const DistillGrads& gr,
                       int accumulate, hipStream_t s) {
  const CellTailArgs& w = a.tail;
  const Tape tp(a.B, w.P, w.Hd, w.O);
  const Scratch sc(a.B, a.G, w.P, w.Hd, w.O, a.Na);
  Traced tr(K_DISTILL_TAIL, 4.0 * a.B * (a.Na + a.G) * DT_O + 2.0 * a.B * a.G * w.Hd, s,
            4.0 * (a.Na + a.G) * (2 * DT_O + 2 + a.B) + 4.0 * a.G * (w.Hd + a.B));
  launch_decoder<1>(a, tape + tp.out, nullptr, gloss, gr, accumulate, ws + sc.dout, sc.nwg, s);
  int threads = w.P > w.Hd ? w.P : w.Hd;
  if (w.O > threads) threads = w.O;
  threads = (threads + 63) / 64 * 64;
  hipLaunchKernelGGL(distill_mid_bwd_kernel, dim3(a.B), dim3(1024), 0, s, a, tape, ws + sc.dout, sc.nwg, gloss, ws,
                     sc.v_dout, sc.v_da, sc.v_du, sc.v_dh);
  const long small = (long)w.O * 2 * w.P + (long)w.P * w.Hd + w.O + w.P + 3L * w.Hd;
  hipLaunchKernelGGL(distill_small_grads_kernel, dim3((unsigned)((small + 255) / 256)), dim3(256), 0, s, a, tape, ws,
                     sc.v_dout, sc.v_da, sc.v_du, sc.v_dh, gr, accumulate);
  const dim3 grid((a.G + 255) / 256, (w.Hd + DT_W1_ROWS - 1) / DT_W1_ROWS);
#define SCM_DT_CASE(BM)                                                                                            \
  hipLaunchKernelGGL((distill_w1_grad_kernel<BM>), grid, dim3(256), 0, s, a.x, ws + sc.v_dh, a.B, a.G, w.Hd, gr.dW1, \
                     accumulate)
  if (a.B <= 4) SCM_DT_CASE(4);
  else if (a.B <= 8) SCM_DT_CASE(8);
  else if (a.B <= 16) SCM_DT_CASE(16);
  else SCM_DT_CASE(32);
#undef SCM_DT_CASE
  return hipGetLastError();
}

}  // namespace scm
