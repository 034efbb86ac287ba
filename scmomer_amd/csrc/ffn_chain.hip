// Forward of the FeedForward sublayer as ONE kernel (PreLayerNorm(Chunk(FeedForward)), performer_pytorch.py:289-331,
// with the residual add of reversible.py:145), for model dim 200 and a hidden width that is a multiple of 8 (at most 2048).
//
//   y = x + w2(gelu(w1(LN(x)) + b1)) + b2          (optionally also LN_next(y) for the next sublayer)
//
// Used where no tape is kept (extraction, evaluation): there it replaces four launches and the 2 x [T, 800] round
// trip of the hidden activations.  With a tape the separate kernels are as fast (all workgroups of this kernel
// load their rows at the same time, then compute, then store: the phases do not overlap at 3 cells per GPU).
//
// Two chained MFMA products.  A workgroup (8 waves) owns up to 256 token rows for the whole hidden width: the rows'
// A fragments (13 k-steps) stay in registers; the hidden width is walked 64 units at a time.  The first product is
// formed transposed (hidden units on the MFMA rows, tokens on the lanes) exactly as in gemm_skinny_kernel, its
// epilogue (bias + GELU) runs on the accumulators in registers, and the bf16 result -- which is also
// stored, for the weight-gradient products -- is at once the B operand of the second product, which contracts
// over accumulator rows: the hidden activations never make a round trip through memory between the two products.
// The second product accumulates, per wave, the full 200-wide output rows of its 32 tokens (7 accumulator tiles);
// a lane ends up holding half of its token's row, so the LayerNorm statistics on either side are in-lane sums plus
// one exchange between the two half-waves.
//
// Weight panels of a 64-unit chunk are copied global -> LDS asynchronously into a two-stage ring (no staging
// registers): the [64 x 208] slice of the first weight (rows permuted so that every lane holds 8 consecutive hidden
// units per group) and the [200 x 64] slice of the second weight (rows permuted the same way over pairs of
// accumulator tiles, k in "chain order" -- see chain_pos() -- prepared once per weight update).
#include "scm_launch.hpp"

namespace scm {

constexpr int FC_KS = 13;                       // k-steps of the first product (K = 200 -> 208)
constexpr int FC_W1BYTES = 4 * 64 * 128;        // first-product panel: 4 k-blocks of [64 rows][64 k] images
constexpr int FC_W2ROWS = 224;                  // second-product panel rows (7 tiles)
constexpr int FC_W2BYTES = FC_W2ROWS * 128;     // [224 rows][64 k]
constexpr int FC_STAGE = FC_W1BYTES + FC_W2BYTES;   // 60 KB
constexpr int FC_VEC = 208;                     // floats per small vector in LDS (gamma, beta, ...)
constexpr int FC_MAXF = 2048;                   // hidden width bound (bias vector in LDS)

// column (0..63) of a chunk, or of a pair of output tiles, that sits in panel row pr (0..63): tile t = pr>>5, MFMA
// row i = pr&31 = 8m + 4h + n -> column 16m + 8h + 4t + n; register r = 4m + n of tile t on lane-half h then holds
// that column, i.e. the lane has 8 consecutive columns per register group.
SCM_DEV int fc_panel_col(int pr) {
  const int t = pr >> 5, i = pr & 31;
  return 16 * (i >> 3) + 8 * ((i >> 2) & 1) + 4 * t + (i & 3);
}

// The 8 f32 values of a lane's token at columns 16 m + 8 half .. + 7 (m = 0..12; m = 12 exists on half 0 only)
// map to the second product's accumulators as: m = 4 q + g (q < 3) -> tiles 2q, 2q+1, registers 4g..4g+3;
// m = 12 -> tile 6, registers 0..7 of half 0.
SCM_DEV float fc_get(const f32x16 (&Y)[7], int m, int j) {   // value j (0..7) of column group m
  return m < 12 ? Y[2 * (m >> 2) + (j >> 2)][4 * (m & 3) + (j & 3)] : Y[6][j];
}
SCM_DEV void fc_set(f32x16 (&Y)[7], int m, int j, float v) {
  if (m < 12) Y[2 * (m >> 2) + (j >> 2)][4 * (m & 3) + (j & 3)] = v;
  else Y[6][j] = v;
}

__global__ __launch_bounds__(512) void ffn_chain_kernel(FfnChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, half = l >> 5, lr = l & 31;
  const int wu = __builtin_amdgcn_readfirstlane(w);
  constexpr int D = 200;
  const int nchunks = (a.F + 63) >> 6;   // the last chunk may be partial: units from F on are zero-filled
  const long r0 = (long)a.M * blockIdx.x / gridDim.x, r1 = (long)a.M * (blockIdx.x + 1) / gridDim.x;
  const int R = (int)((r1 - r0 + 7) >> 3);                 // rows per wave, <= 32
  const long row = r0 + (long)wu * R + lr;
  const bool valid = lr < R && row < r1;

  // A launch of several rounds of workgroups (one per CU at a time) would run in lockstep: all CUs load their rows
  // together, compute together, store together, and the memory system idles through every compute phase (measured:
  // the row phases then cost as much as the products).  The first workgroup of every CU therefore starts late by
  // (its number / 8 mod 8) eighths of a workgroup's compute time; the rounds that follow inherit the spread, and a
  // CU's row phases fall into the other CUs' compute phases.  The 100 MHz counter times the delay.
  if (gridDim.x >= 512 && blockIdx.x < 256) {
    const unsigned phase = (blockIdx.x >> 3) & 7;
    const unsigned long long until = __builtin_amdgcn_s_memrealtime() + (unsigned long long)phase * (48u * nchunks);
    // bounded: at most 512 naps of ~1 us, whatever the counter does
    for (int nap = 0; nap < 512 && __builtin_amdgcn_s_memrealtime() < until; ++nap) __builtin_amdgcn_s_sleep(32);
  }

  float* bias1_s = (float*)(smem + 2 * FC_STAGE);          // [F] (forward)
  float* vec_s = bias1_s + FC_MAXF;                        // [5][FC_VEC]: gamma, beta, bias2, gamma_next, beta_next
  {
    for (int i = tid; i < 64 * nchunks; i += 512) bias1_s[i] = (a.bias1 && i < a.F) ? a.bias1[i] : 0.f;
    for (int i = tid; i < FC_VEC; i += 512) {
      const bool in = i < D;
      vec_s[i] = in ? a.gamma[i] : 0.f;
      vec_s[FC_VEC + i] = in ? a.beta[i] : 0.f;
      vec_s[2 * FC_VEC + i] = (in && a.bias2) ? a.bias2[i] : 0.f;
      vec_s[3 * FC_VEC + i] = (in && a.gamma_n) ? a.gamma_n[i] : 0.f;
      vec_s[4 * FC_VEC + i] = (in && a.beta_n) ? a.beta_n[i] : 0.f;
    }
  }

  // ---- asynchronous weight-panel copies: per-lane source offsets (relative to the chunk's first unit)
  const u32x4 rW1 = dma_rsrc(a.W1, (uint32_t)(((long)(a.F - 1) * a.ldw1 + D) * 2));
  const u32x4 rW2 = dma_rsrc(a.W2c, (uint32_t)(((long)(D - 1) * a.ldw2 + 64 * nchunks) * 2));   // padded to whole chunks
  const uint32_t lds0 = lds_addr_of(smem);
  // The per-lane source offsets are recomputed at every issue (a few integer operations) rather than held in
  // registers for the whole kernel: the main loop has none to spare.
  auto issue = [&](int c, int buf) {
    if (c >= nchunks) return;
    const uint32_t b1 = lds0 + buf * FC_STAGE, b2 = b1 + FC_W1BYTES;
    const uint32_t s1 = (uint32_t)((long)64 * c * a.ldw1 * 2), s2 = (uint32_t)(64 * c * 2);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      // first weight: copy unit u = 4 w + jj: k-block u >> 3, image rows 8 (u & 7) .. + 7
      const int u = 4 * w + jj, kb = u >> 3, pr = 8 * (u & 7) + (l >> 3);
      const int k = 64 * kb + 8 * swz128(pr, l & 7);
      const int unit = fc_panel_col(pr);                   // hidden unit inside the chunk
      const uint32_t v = (k < D && 64 * c + unit < a.F) ? (uint32_t)((long)unit * a.ldw1 + k) * 2 : BUF_OOB;
      dma16_to_lds(rW1, b1 + 1024 * (4 * wu + jj), v, s1);
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      // second weight: copy unit u = w + 8 jj (< 28): image rows 8 u .. + 7
      const int u = w + 8 * jj, pr = 8 * u + (l >> 3);
      int o = -1;
      if (pr < 192) o = 64 * (pr >> 6) + fc_panel_col(pr & 63);
      else if (pr - 192 < 4) o = pr;                       // columns 192..195 on rows 0..3 of tile 6
      else if (pr - 192 >= 8 && pr - 192 < 12) o = pr - 4; // columns 196..199 on rows 8..11 (lane-half 0, registers 4..7)
      const uint32_t v = (o >= 0) ? (uint32_t)((long)o * a.ldw2 + 8 * swz128(pr, l & 7)) * 2 : BUF_OOB;
      if (wu + 8 * jj < 28) dma16_to_lds(rW2, b2 + 1024 * (wu + 8 * jj), v, s2);
    }
  };
  issue(0, 0);

  int roff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) roff[i] = img_off(lr, 16 * i + 8 * half);

  // ---- prologue: this lane's half row -> A fragments (bf16); second-product accumulators
  bf16x8 af[FC_KS];
  f32x16 Y[7];
#pragma unroll
  for (int t = 0; t < 7; ++t) Y[t] = zero16();
  __syncthreads();   // the small vectors are in LDS
  {
    const brsrc rX = make_rsrc(a.x, (uint32_t)(((long)(a.M - 1) * a.ldx + D) * 4));
    f32x4 xr[FC_KS][2];
#pragma unroll
    for (int m = 0; m < FC_KS; ++m) {
      const bool in = valid && (m < 12 || half == 0);
      const uint32_t off = in ? (uint32_t)(row * a.ldx + 16 * m + 8 * half) * 4 : BUF_OOB;
      xr[m][0] = as_f32x4(buf_load16(rX, off));
      xr[m][1] = as_f32x4(buf_load16(rX, off + 16));
    }
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < FC_KS; ++m)
#pragma unroll
      for (int j = 0; j < 8; ++j) s += xr[m][j >> 2][j & 3];
    const float mean = xhalf_sum(s) * (1.f / D);
    float q = 0.f;
#pragma unroll
    for (int m = 0; m < FC_KS; ++m)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = xr[m][j >> 2][j & 3] - mean;
        q += (m < 12 || half == 0) ? d * d : 0.f;   // columns 200..207 do not exist
      }
    const float rstd = rsqrtf(xhalf_sum(q) * (1.f / D) + 1e-5f);
    const brsrc rH = make_rsrc(a.h, a.h ? (uint32_t)(((long)(a.M - 1) * a.ldh + D) * 2) : 0u);
#pragma unroll
    for (int m = 0; m < FC_KS; ++m) {
      const int col = 16 * m + 8 * half;
      const f32x4 g0 = *(const f32x4*)(vec_s + col), g1 = *(const f32x4*)(vec_s + col + 4);
      const f32x4 b0 = *(const f32x4*)(vec_s + FC_VEC + col), b1 = *(const f32x4*)(vec_s + FC_VEC + col + 4);
      const f32x4 c0 = *(const f32x4*)(vec_s + 2 * FC_VEC + col), c1 = *(const f32x4*)(vec_s + 2 * FC_VEC + col + 4);
      bf16x8 hv;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xv = xr[m][j >> 2][j & 3];
        const float g = j < 4 ? g0[j & 3] : g1[j & 3], b = j < 4 ? b0[j & 3] : b1[j & 3];
        const float c2 = j < 4 ? c0[j & 3] : c1[j & 3];
        const bool in = valid && (m < 12 || half == 0);
        hv[j] = (bf16)(in ? (xv - mean) * rstd * g + b : 0.f);
        // second-product accumulators start at residual + bias
        if (m < 12 || half == 0) fc_set(Y, m, j, xv + c2);
      }
      af[m] = hv;
      if (a.h) buf_store16(rH, (valid && (m < 12 || half == 0)) ? (uint32_t)(row * a.ldh + col) * 2 : BUF_OOB, as_u32x4(hv));
    }
    if (a.mean && valid && half == 0) { a.mean[row] = mean; a.rstd[row] = rstd; }
  }

  // ---- main loop over 64-unit chunks of the hidden width
  const brsrc rO0 = make_rsrc(a.pre, a.pre ? (uint32_t)(((long)(a.M - 1) * a.ldo0 + a.F) * 2) : 0u);
  const brsrc rO1 = make_rsrc(a.act, a.act ? (uint32_t)(((long)(a.M - 1) * a.ldo1 + a.F) * 2) : 0u);
  const int nstores = (a.pre ? 4 : 0) + (a.act ? 4 : 0);   // stores per chunk and lane

  for (int c = 0; c < nchunks; ++c) {
    // This wave's copies of chunk c have landed once only the memory operations issued after them are still in
    // flight (operations complete in issue order): the previous chunk's stores.  Waiting for those as well would
    // expose a write round trip per chunk.
    if (c == 0) wait_vmcnt<0>();
    else if (nstores == 8) wait_vmcnt<8>();
    else if (nstores == 4) wait_vmcnt<4>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();          // everyone's have; everyone is done reading the other stage
    issue(c + 1, (c + 1) & 1);
    const char* W1s = smem + (c & 1) * FC_STAGE;
    const char* W2s = W1s + FC_W1BYTES;
    f32x16 acc0 = zero16(), acc1 = zero16();
#pragma unroll
    for (int ks = 0; ks < FC_KS; ++ks) {
      const char* blk = W1s + (ks >> 2) * (64 * 128);
      acc0 = mfma32(*(const bf16x8*)(blk + roff[ks & 3]), af[ks], acc0);
      acc1 = mfma32(*(const bf16x8*)(blk + 4096 + roff[ks & 3]), af[ks], acc1);
    }
    // epilogue on registers: lane = its token, group g = hidden units 64 c + 16 g + 8 half .. + 7
    const uint32_t obase = 64 * c + 8 * half;
    bf16x8 gv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float v[8] = {acc0[4 * g], acc0[4 * g + 1], acc0[4 * g + 2], acc0[4 * g + 3],
                    acc1[4 * g], acc1[4 * g + 1], acc1[4 * g + 2], acc1[4 * g + 3]};
      const bool st = valid && (int)obase + 16 * g < a.F;
      const float* bp = bias1_s + 64 * c + 16 * g + 8 * half;
      const f32x4 b0 = *(const f32x4*)bp, b1 = *(const f32x4*)(bp + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] += b0[j]; v[4 + j] += b1[j]; }
      bf16x8 ov;
#pragma unroll
      for (int j = 0; j < 8; ++j) { ov[j] = (bf16)v[j]; gv[g][j] = (bf16)gelu_erf(v[j]); }
      if (a.pre) buf_store16(rO0, st ? (uint32_t)(row * a.ldo0 + obase + 16 * g) * 2 : BUF_OOB, as_u32x4(ov));
    }
    if (a.act) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        buf_store16(rO1, (valid && (int)obase + 16 * g < a.F) ? (uint32_t)(row * a.ldo1 + obase + 16 * g) * 2 : BUF_OOB,
                    as_u32x4(gv[g]));
    }
    // second product: Y^T[200 x tokens] += W2c[200 x 64] * (bf16 hidden chunk)^T, k-step s' = 2 t + s pairs
    // registers 8s..8s+7 of first-product tile t with chain positions 16 s' + 8 half + j of the panel
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 lo8 = gv[2 * s], hi8 = gv[2 * s + 1];
        bf16x8 bfrag;
#pragma unroll
        for (int j = 0; j < 4; ++j) { bfrag[j] = lo8[4 * t + j]; bfrag[4 + j] = hi8[4 * t + j]; }
        const int sp = 2 * t + s;
#pragma unroll
        for (int ot = 0; ot < 7; ++ot) Y[ot] = mfma32(*(const bf16x8*)(W2s + 4096 * ot + roff[sp]), bfrag, Y[ot]);
      }
  }

  // ---- final epilogue
  {
    const brsrc rY = make_rsrc(a.y, (uint32_t)(((long)(a.M - 1) * a.ldy + D) * 4));
    const bool ln_next = a.h_n != nullptr;
    float mean = 0.f, rstd = 0.f;
    if (ln_next) {
      float s = 0.f;
#pragma unroll
      for (int m = 0; m < FC_KS; ++m)
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (m < 12 || half == 0) s += fc_get(Y, m, j);
      mean = xhalf_sum(s) * (1.f / D);
      float q = 0.f;
#pragma unroll
      for (int m = 0; m < FC_KS; ++m)
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (m < 12 || half == 0) { const float d = fc_get(Y, m, j) - mean; q += d * d; }
      rstd = rsqrtf(xhalf_sum(q) * (1.f / D) + 1e-5f);
    }
    const brsrc rHn = make_rsrc(a.h_n, ln_next ? (uint32_t)(((long)(a.M - 1) * a.ldhn + D) * 2) : 0u);
#pragma unroll
    for (int m = 0; m < FC_KS; ++m) {
      const int col = 16 * m + 8 * half;
      const bool in = valid && (m < 12 || half == 0);
      f32x4 y0, y1;
#pragma unroll
      for (int j = 0; j < 4; ++j) { y0[j] = fc_get(Y, m, j); y1[j] = fc_get(Y, m, 4 + j); }
      const uint32_t off = in ? (uint32_t)(row * a.ldy + col) * 4 : BUF_OOB;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y0), rY, (int)off, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y1), rY, (int)(off + 16), 0, 0);
      if (ln_next) {
        const f32x4 g0 = *(const f32x4*)(vec_s + 3 * FC_VEC + col), g1 = *(const f32x4*)(vec_s + 3 * FC_VEC + col + 4);
        const f32x4 b0 = *(const f32x4*)(vec_s + 4 * FC_VEC + col), b1 = *(const f32x4*)(vec_s + 4 * FC_VEC + col + 4);
        bf16x8 hv;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          hv[j] = (bf16)((y0[j] - mean) * rstd * g0[j] + b0[j]);
          hv[4 + j] = (bf16)((y1[j] - mean) * rstd * g1[j] + b1[j]);
        }
        buf_store16(rHn, in ? (uint32_t)(row * a.ldhn + col) * 2 : BUF_OOB, as_u32x4(hv));
      }
    }
    if (ln_next && valid && half == 0) { a.mean_n[row] = mean; a.rstd_n[row] = rstd; }
  }
}

// ------------------------------------------------------------------------------------------
int ffn_chain_grid(long M) {
  const long rounds = (M + 256l * 256 - 1) / (256l * 256);
  long g = 256 * rounds;
  const long need = (M + 31) / 32;          // small problems: at least 32 rows per workgroup
  if (g > need) g = need;
  return (int)(g < 1 ? 1 : g);
}

bool ffn_chain_supported(long M, int D, int F) {
  return D == 200 && F >= 8 && F % 8 == 0 && F <= FC_MAXF && M > 0;
}

// The kernel forms 32-bit byte offsets from row numbers and leading dimensions: a tall input runs as several launches
// over row slabs (rows are independent), each slab with every operand below 2 GiB.
hipError_t ffn_chain(FfnChainArgs a, hipStream_t s) {
  const size_t lds = 2 * FC_STAGE + (FC_MAXF + 5 * FC_VEC) * sizeof(float);
  static std::atomic<unsigned long long> ready;
  if (hipError_t e = ensure_dynamic_lds((const void*)ffn_chain_kernel, lds, ready); e != hipSuccess) return e;
  long row_bytes = a.ldx * 4;
  if (a.ldy * 4 > row_bytes) row_bytes = a.ldy * 4;
  if (a.h && a.ldh * 2 > row_bytes) row_bytes = a.ldh * 2;
  if (a.pre && a.ldo0 * 2 > row_bytes) row_bytes = a.ldo0 * 2;
  if (a.act && a.ldo1 * 2 > row_bytes) row_bytes = a.ldo1 * 2;
  if (a.h_n && a.ldhn * 2 > row_bytes) row_bytes = a.ldhn * 2;
  const long lim = (1l << 31) - 4096;
  long rows = a.M;
  if (a.M * row_bytes > lim) {
    rows = lim / row_bytes / (256 * 32) * (256 * 32);      // whole rounds of 256 workgroups x 32-row tiles
    if (rows < 32) return hipErrorInvalidValue;             // a leading dimension beyond 64 MiB per row
  }
  const long Mtot = a.M;
  for (long r0 = 0; r0 < Mtot; r0 += rows) {
    FfnChainArgs b = a;
    b.M = Mtot - r0 < rows ? Mtot - r0 : rows;
    b.x = a.x + r0 * a.ldx; b.y = a.y + r0 * a.ldy;
    if (a.mean) b.mean = a.mean + r0;
    if (a.rstd) b.rstd = a.rstd + r0;
    if (a.h) b.h = a.h + r0 * a.ldh;
    if (a.pre) b.pre = a.pre + r0 * a.ldo0;
    if (a.act) b.act = a.act + r0 * a.ldo1;
    if (a.h_n) { b.h_n = a.h_n + r0 * a.ldhn; b.mean_n = a.mean_n + r0; b.rstd_n = a.rstd_n + r0; }
    const int grid = ffn_chain_grid(b.M);
    const double M = (double)b.M, D = 200.0, F = (double)a.F;
    // algorithmic bytes: x in, y out (+ pre, act, h when saved, + h_next); weights once
    const double bytes = 2.0 * 2 * D * F +
                         M * (D * 8 + (a.pre ? F * 2 : 0) + (a.act ? F * 2 : 0) + (a.h ? D * 2 : 0) + (a.h_n ? D * 2 : 0));
    Traced tr(K_FFN_CHAIN_FWD, 4.0 * M * D * F, s, bytes);
    hipLaunchKernelGGL(ffn_chain_kernel, dim3(grid), dim3(512), lds, s, b);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
  }
  return hipSuccess;
}

}  // namespace scm
