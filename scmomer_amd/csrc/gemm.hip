// Tall-skinny bf16 MFMA GEMMs of the Performer layer (tokens x {200,640,800,1920}).
//
//  gemm_nt : Y[M,N] = X[M,K] * W[N,K]^T (+ fused epilogue).  nn.Linear forward
//            (performer_pytorch.py:323,330,374,402) and the data-gradient products of its
//            backward (W passed pre-transposed).
//  wgrad   : C[P,Q] = sum_t A[t,P] * B[t,Q] over a token slice -> f32 slab per slice; the
//            weight-gradient products.  Both operands are token-major, so MFMA fragments are
//            read with the hardware-transposing LDS read.
//  wgrad_reduce : sums the slabs into the f32 gradient (optionally transposed, optional bias
//            column), deterministic.
#include "scm_launch.hpp"
#include <stdlib.h>

namespace scm {

// ------------------------------------------------------------------------------------------
// gemm_nt
// ------------------------------------------------------------------------------------------
enum { EPI_BF16 = 0, EPI_F32_BIAS_RES = 1, EPI_GELU_DUAL = 2, EPI_DGELU = 3, EPI_F32 = 4, EPI_BF16_BIAS = 5 };
constexpr int EPI_RES_LN = 7;   // internal: bias + residual, then the next sublayer's LayerNorm forward on the rows (gemm_res_ln)
constexpr int EPI_LN_BWD = 6;   // internal: LayerNorm backward on the product's rows (gemm_ln_bwd), not an scm_gemm_nt epilogue

constexpr int G_BM = 128;
constexpr int G_BK = 64;
constexpr int G_SP = 68;   // floats per staged epilogue row (64 + 4: rows 16-byte aligned, conflict-free)

// Epilogue of one pair of 32x32 accumulator tiles (columns col0..col0+63; `two` = second tile present).
// Accumulators (column on the lane, rows in registers) are staged per wave as [32 rows][64 columns] f32,
// then read back row-wise so that every lane handles 8 consecutive columns of one row: 16-byte (bf16) /
// 32-byte (f32) global stores, 128..256 contiguous bytes per row; the bias / residual / pre-activation
// operands are read with the same coalesced shape.
template <int EPI, bool FOLDED = false>   // FOLDED: bias and residual already sit in the accumulators
SCM_DEV void epilogue_pair(float* stage, const f32x16& acc0, const f32x16& acc1, bool two, long row0, int col0,
                           int M, int N, const GemmEpi& e) {
  const int l = lane_id(), half = l >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    stage[acc_row(r, half) * G_SP + (l & 31)] = acc0[r];
    if (two) stage[acc_row(r, half) * G_SP + 32 + (l & 31)] = acc1[r];
  }
  __builtin_amdgcn_wave_barrier();
  const int ch = l & 7;
  const int col = col0 + 8 * ch;
  constexpr bool FOLDED_OR_PLAIN = EPI == EPI_F32 || (EPI == EPI_F32_BIAS_RES && FOLDED);
  const bool col_ok = FOLDED_OR_PLAIN ? true : (two || ch < 4) && col < N;
  float bias[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bias[j] = 0.f;
  if (!FOLDED && (EPI == EPI_F32_BIAS_RES || EPI == EPI_GELU_DUAL || EPI == EPI_BF16_BIAS) && e.bias && col_ok) {
    const f32x4 b0 = *(const f32x4*)(e.bias + col), b1 = *(const f32x4*)(e.bias + col + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { bias[j] = b0[j]; bias[4 + j] = b1[j]; }
  }
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int rl = 8 * pass + (l >> 3);
    const long row = row0 + rl;
    if (!col_ok || row >= M) continue;
    if (FOLDED_OR_PLAIN) {
      // f32 output with nothing left to add: the lane takes columns 4ch..4ch+3 of each 32-column half, so the eight
      // lanes of a row write 128 contiguous bytes per store instruction (16-byte pieces at a 32-byte stride made
      // twice the write requests)
      float* op = (float*)e.out0 + row * e.ldo0 + col0 + 4 * ch;
      if (col0 + 4 * ch < N) *(f32x4*)op = *(const f32x4*)(stage + rl * G_SP + 4 * ch);
      if (two && col0 + 32 + 4 * ch < N) *(f32x4*)(op + 32) = *(const f32x4*)(stage + rl * G_SP + 32 + 4 * ch);
      continue;
    }
    const f32x4 a0 = *(const f32x4*)(stage + rl * G_SP + 8 * ch);
    const f32x4 a1 = *(const f32x4*)(stage + rl * G_SP + 8 * ch + 4);
    float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += bias[j];
    if (EPI == EPI_BF16 || EPI == EPI_BF16_BIAS) {
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (bf16)v[j];
      *(bf16x8*)((bf16*)e.out0 + row * e.ldo0 + col) = o;
    } else if (EPI == EPI_F32 || EPI == EPI_F32_BIAS_RES) {
      if (!FOLDED && EPI == EPI_F32_BIAS_RES && e.resid) {
        const f32x4 r0 = *(const f32x4*)(e.resid + row * e.ldr + col);
        const f32x4 r1 = *(const f32x4*)(e.resid + row * e.ldr + col + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] += r0[j]; v[4 + j] += r1[j]; }
      }
      float* op = (float*)e.out0 + row * e.ldo0 + col;
      *(f32x4*)op = f32x4{v[0], v[1], v[2], v[3]};
      *(f32x4*)(op + 4) = f32x4{v[4], v[5], v[6], v[7]};
    } else if (EPI == EPI_GELU_DUAL) {
      bf16x8 o, g;
#pragma unroll
      for (int j = 0; j < 8; ++j) { o[j] = (bf16)v[j]; g[j] = (bf16)gelu_erf(v[j]); }
      *(bf16x8*)((bf16*)e.out0 + row * e.ldo0 + col) = o;
      *(bf16x8*)((bf16*)e.out1 + row * e.ldo1 + col) = g;
    } else if (EPI == EPI_DGELU) {
      const bf16x8 pv = *(const bf16x8*)(e.pre + row * e.ldp + col);
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (bf16)(v[j] * gelu_erf_grad((float)pv[j]));
      *(bf16x8*)((bf16*)e.out0 + row * e.ldo0 + col) = o;
    }
  }
  __builtin_amdgcn_wave_barrier();
}

template <int NT, bool A_F32>
struct GemmStage {
  // per-thread staging registers: 4 chunks of A, NT chunks of W (8 bf16 each)
  bf16x8 a[4];
  bf16x8 w[NT];
};

template <int NT, bool A_F32, int EPI>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const void* __restrict__ Ap, long lda,
                                                      const bf16* __restrict__ W, long ldw, int M, int N, int K,
                                                      GemmEpi e) {
  constexpr int BN = NT * 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;                 // [128][64] bf16 swizzled image
  char* Ws = smem + G_BM * 128;    // [BN][64]
  const int tid = threadIdx.x, w = tid >> 6;
  const int m0 = blockIdx.x * G_BM, n0 = blockIdx.y * BN;
  const int nk = (K + G_BK - 1) / G_BK;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = zero16();

  GemmStage<NT, A_F32> st;
  auto load_stage = [&](int kt) {
    const int k0 = kt * G_BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 256 * i, row = c >> 3, ch = c & 7;
      const int gm = m0 + row, gk = k0 + 8 * ch;
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (bf16)0.f;
      if (gm < M && gk < K) {
        if (A_F32) {
          const float* p = (const float*)Ap + (long)gm * lda + gk;
          f32x4 x0 = *(const f32x4*)p, x1 = *(const f32x4*)(p + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { v[j] = (bf16)x0[j]; v[4 + j] = (bf16)x1[j]; }
        } else {
          v = *(const bf16x8*)((const bf16*)Ap + (long)gm * lda + gk);
        }
      }
      st.a[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int c = tid + 256 * i, row = c >> 3, ch = c & 7;
      const int gn = n0 + row, gk = k0 + 8 * ch;
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (bf16)0.f;
      if (gn < N && gk < K) v = *(const bf16x8*)(W + (long)gn * ldw + gk);
      st.w[i] = v;
    }
  };
  auto write_stage = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 256 * i, row = c >> 3, ch = c & 7;
      *(bf16x8*)(As + img_off(row, 8 * ch)) = st.a[i];
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int c = tid + 256 * i, row = c >> 3, ch = c & 7;
      *(bf16x8*)(Ws + img_off(row, 8 * ch)) = st.w[i];
    }
  };

  // fragment byte offsets inside a 32-row block of an image, per k-step (hoisted out of the loops)
  const int l = tid & 63;
  int roff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) roff[i] = img_off(l & 31, 16 * i + 8 * (l >> 5));
  const char* Aw = As + 4096 * w;

  load_stage(0);
  const int nk_full = K / G_BK;
  for (int kt = 0; kt < nk_full; ++kt) {      // full k-tiles: straight-line, no branches between MFMAs
    write_stage();
    __syncthreads();
    if (kt + 1 < nk) load_stage(kt + 1);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 a = *(const bf16x8*)(Aw + roff[s]);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = mfma32(a, *(const bf16x8*)(Ws + 4096 * t + roff[s]), acc[t]);
    }
    __syncthreads();
  }
  if (nk_full < nk) {                          // K tail: a rolled loop over its k-steps (zero padded to 16)
    write_stage();
    __syncthreads();
    const int ks = (K - nk_full * G_BK + 15) >> 4;
    const int hoff = 16 * (l >> 5);
#pragma unroll 1
    for (int s = 0; s < ks; ++s) {
      // chunk index 2s + half, swizzled per row
      const int off = (l & 31) * 128 + (swz128(l & 31, 2 * s + (l >> 5)) << 4);
      (void)hoff;
      const bf16x8 a = *(const bf16x8*)(Aw + off);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = mfma32(a, *(const bf16x8*)(Ws + 4096 * t + off), acc[t]);
    }
    __syncthreads();
  }

  float* stage = (float*)smem + w * (32 * G_SP);
#pragma unroll
  for (int tp = 0; tp < (NT + 1) / 2; ++tp)
    epilogue_pair<EPI>(stage, acc[2 * tp], acc[(2 * tp + 1 < NT) ? 2 * tp + 1 : 2 * tp], 2 * tp + 1 < NT,
                       (long)m0 + 32 * w, n0 + 64 * tp, M, N, e);
}

// ------------------------------------------------------------------------------------------
// gemm_nt_dma: Y[M, N<=224] = A[M,K] * W[N,K]^T for bf16 A, fed by asynchronous global->LDS copies.
//   The rows are dealt out EVENLY: workgroup b owns rows [M b/G, M (b+1)/G), at most 224 of them (7 waves x 32
//   rows x 7 accumulators), with G a multiple of the CU count, so one round of workgroups keeps every CU equally
//   busy on this bandwidth-bound product.  64-wide k stages go through LDS rings (A: three stages, W: two): while
//   stage i is multiplied, the A copies of stages i+1 and i+2 are in flight (no staging registers, one barrier per
//   stage); what a CU has in flight bounds the rate at which its rows stream in from HBM.  The LDS
//   images are the swizzled 128-byte-row images of gemm_nt (each lane of a copy fetches the 16-byte chunk that
//   belongs at its linear LDS position).  Epilogue: the LDS-staged coalesced one above.
// ------------------------------------------------------------------------------------------
constexpr int GD_NT = 7;                       // accumulator tiles per wave (N <= 224)
constexpr int GD_ROWS = 224;                   // rows per workgroup (7 waves)
constexpr int GD_ABYTES = GD_ROWS * 128;       // one A stage: [224 rows][64 k] bf16, 28 KB
constexpr int GD_WBYTES = GD_NT * 32 * 128;    // one W stage: [224 columns][64 k] bf16, 28 KB
constexpr int GD_NA = 3, GD_NW = 2;            // ring depths: A streams from HBM (two stages ahead), W sits in L2
constexpr int GD_LDS = GD_NA * GD_ABYTES + GD_NW * GD_WBYTES;   // 140 KB

// LayerNorm backward as the epilogue of the N <= 224 product: dx = dres + LN'(dh) with dh = the accumulators, never
// written to memory.  Each wave stages its 32 rows, 16 at a time, as f32 rows in LDS and then walks them one row per
// trip with the lane = 4 columns shape of ln_bwd_kernel (same arithmetic, same column partials).  The x / dres rows
// come in batches of four, two batches in flight: a batch is requested before the previous one's stores are issued
// (memory operations retire in issue order).  Rows past r1 and columns past N are out of range of the buffer
// resources: they load zeros and their stores are dropped.
constexpr int LNB_PITCH = 232;                 // floats per staged row; 4 * 232 mod 64 = 32: the lane halves (rows 4 apart) use disjoint banks
constexpr int LNB_STAGE = 16 * LNB_PITCH;      // floats per wave
constexpr int LNB_LDS = 7 * LNB_STAGE * 4 + 7 * 3 * 256 * 4;   // stages + cross-wave column sums: 122.5 KB

SCM_DEV void ln_bwd_epilogue(char* smem, f32x16 (&acc)[7], long r0, long r1, int M, int N, const GemmEpi& e) {
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, half = l >> 5;
  float* stage = (float*)smem + w * LNB_STAGE;
  const bool active = 4 * l < N;
  const int lcol = active ? 4 * l : 0;                     // inactive lanes read column 0 and contribute nothing
  const uint32_t lane_off = active ? 16u * l : BUF_OOB;    // byte offset inside a row; out of range for inactive lanes
  const int base = (int)r0 + 32 * w, rend = (int)r1;
  const uint32_t bytes4 = (uint32_t)((long)M * N * 4);     // < 1 GiB (gemm_ln_bwd_supported)
  const brsrc rX = make_rsrc(e.ln_x, bytes4);
  const brsrc rR = make_rsrc(e.resid ? (const void*)e.resid : (const void*)e.ln_x, e.resid ? bytes4 : 0u);
  const brsrc rDx = make_rsrc(e.out0, bytes4);
  const brsrc rDb = make_rsrc(e.ln_dxb ? (void*)e.ln_dxb : e.out0, e.ln_dxb ? bytes4 / 2 : 0u);
  const brsrc rMu = make_rsrc(e.ln_mean, (uint32_t)M * 4), rRs = make_rsrc(e.ln_rstd, (uint32_t)M * 4);
  const brsrc rG = make_rsrc(e.ln_gamma, (uint32_t)N * 4);
  const f32x4 g = as_f32x4(buf_load16(rG, lane_off));
  const float invN = 1.f / (float)N;

  f32x4 xv[2][4], rv[2][4];
  float mu4[2], rs4[2];
  // byte offset of this lane's piece of row base + 4 b + j; rows past the workgroup's range get an offset past 1 GiB
  auto row_off = [&](int b, int j) -> uint32_t {
    const int row = base + 4 * b + j;                      // wave-uniform
    return (row < rend ? (uint32_t)row * (uint32_t)(N * 4) : 0x40000000u) + lane_off;
  };
  auto load_batch = [&](int b, int set) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t off = row_off(b, j);
      xv[set][j] = as_f32x4(buf_load16(rX, off));
      rv[set][j] = as_f32x4(buf_load16(rR, off));
    }
    const int srow = base + 4 * b + (l & 3);
    const int soff = srow < rend ? srow * 4 : (int)BUF_OOB;
    mu4[set] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rMu, soff, 0, 0));
    rs4[set] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rRs, soff, 0, 0));
    __builtin_amdgcn_sched_barrier(0);
  };
  auto stage_pass = [&](int p) {
#pragma unroll
    for (int t = 0; t < 7; ++t) {
      const int col = 32 * t + (l & 31);     // all 224 columns are staged (the pitch holds them); those past N are not read
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) stage[(acc_row(8 * p + rr, half) - 16 * p) * LNB_PITCH + col] = acc[t][8 * p + rr];
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  f32x4 ag = f32x4{0, 0, 0, 0}, ab = ag, ar = ag;
  // one batch: the four rows side by side, so that the eight wave sums overlap (gamma is zero in the inactive lanes,
  // which therefore add nothing to the sums; what they accumulate in ag/ab/ar is never read)
  auto process = [&](int b, int set) {
#pragma unroll
    for (int jp = 0; jp < 4; jp += 2) {
      f32x4 d[2], xh[2];
      float s1[2], s2[2], rsv[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int j = jp + q;
        d[q] = *(const f32x4*)(stage + (4 * (b & 3) + j) * LNB_PITCH + lcol);
        const float mu = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mu4[set]), j));
        rsv[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rs4[set]), j));
        s1[q] = 0.f; s2[q] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          xh[q][i] = (xv[set][j][i] - mu) * rsv[q];
          const float gd = d[q][i] * g[i];
          s1[q] += gd;
          s2[q] += gd * xh[q][i];
        }
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        s1[q] = wave_sum_uniform(s1[q]) * invN;
        s2[q] = wave_sum_uniform(s2[q]) * invN;
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int j = jp + q;
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          o[i] = rv[set][j][i] + rsv[q] * (d[q][i] * g[i] - s1[q] - xh[q][i] * s2[q]);
          ag[i] += d[q][i] * xh[q][i];
          ab[i] += d[q][i];
          ar[i] += rv[set][j][i];
        }
        const uint32_t off = row_off(b, j);
        buf_store16(rDx, off, __builtin_bit_cast(u32x4, o));
        const bf16x4 ob = bf16x4{(bf16)o[0], (bf16)o[1], (bf16)o[2], (bf16)o[3]};
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, ob), rDb, (int)(off >> 1), 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  load_batch(0, 0);
  stage_pass(0);
  load_batch(1, 1);
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    process(b, b & 1);
    if (b == 3) stage_pass(1);            // rows 16..31; the reads of rows 0..15 precede it in the wave's LDS order
    if (b + 2 < 8) load_batch(b + 2, b & 1);
  }
  // column partials of the workgroup: seven waves summed in wave order
  float* red = (float*)smem + 7 * LNB_STAGE;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    red[(w * 3 + 0) * 256 + 4 * l + i] = ag[i];
    red[(w * 3 + 1) * 256 + 4 * l + i] = ab[i];
    red[(w * 3 + 2) * 256 + 4 * l + i] = ar[i];
  }
  __syncthreads();
  for (int i = tid; i < 3 * N; i += 448) {
    const int k = i / N, dd = i - k * N;
    float sum = 0.f;
#pragma unroll
    for (int ww = 0; ww < 7; ++ww) sum += red[(ww * 3 + k) * 256 + dd];
    e.ln_partial[(long)blockIdx.x * 3 * N + i] = sum;
  }
}

// LayerNorm forward of the NEXT sublayer as the epilogue of the residual product: the accumulators already hold
// y = A W^T + bias + resid (they start at bias + resid), so the epilogue has no loads.  Rows are staged as in
// ln_bwd_epilogue and walked with the lane = 4 columns shape of ln_fwd_kernel: y stored as f32, the row statistics by
// two wave sums (mean, then centred squares), h = (y - mean) rstd gamma + beta stored as bf16.  Inactive lanes read the
// first column past N, which holds exact zeros (products with zero-filled weight rows); N <= 220 keeps it staged.
SCM_DEV void res_ln_epilogue(char* smem, f32x16 (&acc)[7], long r0, long r1, int M, int N, const GemmEpi& e) {
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, half = l >> 5;
  float* stage = (float*)smem + w * LNB_STAGE;
  const bool active = 4 * l < N;
  const int lcol = active ? 4 * l : N;
  const uint32_t lane_off = active ? 16u * l : BUF_OOB;
  const int base = (int)r0 + 32 * w, rend = (int)r1;
  const uint32_t bytes4 = (uint32_t)((long)M * N * 4);
  const brsrc rY = make_rsrc(e.out0, bytes4), rH = make_rsrc(e.ln_dxb, bytes4 / 2);
  const brsrc rMu = make_rsrc(e.ln_mean_out, (uint32_t)M * 4), rRs = make_rsrc(e.ln_rstd_out, (uint32_t)M * 4);
  const brsrc rG = make_rsrc(e.ln_gamma, (uint32_t)N * 4), rB = make_rsrc(e.ln_beta, (uint32_t)N * 4);
  const f32x4 g = as_f32x4(buf_load16(rG, lane_off)), bt = as_f32x4(buf_load16(rB, lane_off));
  const float fN = (float)N;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
#pragma unroll
    for (int t = 0; t < 7; ++t) {
      const int col = 32 * t + (l & 31);
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) stage[(acc_row(8 * p + rr, half) - 16 * p) * LNB_PITCH + col] = acc[t][8 * p + rr];
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int b = 0; b < 4; ++b) {          // four rows side by side: their eight wave sums overlap
      f32x4 y[4];
      float mu[4], rs[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        y[j] = *(const f32x4*)(stage + (4 * b + j) * LNB_PITCH + lcol);
        mu[j] = y[j][0] + y[j][1] + y[j][2] + y[j][3];
      }
      // sums, division and rounding in the order of ln_fwd_kernel's row statistics: h, mean and rstd come out with the
      // bits of the two-launch form
#pragma unroll
      for (int j = 0; j < 4; ++j) mu[j] = wave_sum_uniform(mu[j]) / fN;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        {
#pragma clang fp contract(off)   // squares rounded one by one and added in order (no fused multiply-add), as in ln_fwd_kernel
          const float c0 = y[j][0] - mu[j], c1 = y[j][1] - mu[j], c2 = y[j][2] - mu[j], c3 = y[j][3] - mu[j];
          const float p0 = c0 * c0, p1 = c1 * c1, p2 = c2 * c2, p3 = c3 * c3;
          rs[j] = active ? ((p0 + p1) + p2) + p3 : 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) rs[j] = rsqrtf(wave_sum_uniform(rs[j]) / fN + 1e-5f);
      float mstat = 0.f, rstat = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = base + 16 * p + 4 * b + j;       // wave-uniform
        const uint32_t off = (row < rend ? (uint32_t)row * (uint32_t)(N * 4) : 0x40000000u) + lane_off;
        buf_store16(rY, off, __builtin_bit_cast(u32x4, y[j]));
        bf16x4 h;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma clang fp contract(off)
          const float t = rs[j] * (y[j][i] - mu[j]);
          h[i] = (bf16)__builtin_fmaf(g[i], t, bt[i]);
        }
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, h), rH, (int)(off >> 1), 0, 0);
        mstat = l == j ? mu[j] : mstat;
        rstat = l == j ? rs[j] : rstat;
      }
      const int srow = base + 16 * p + 4 * b + l;
      const int soff = (l < 4 && srow < rend) ? srow * 4 : (int)BUF_OOB;
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(mstat), rMu, soff, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(rstat), rRs, soff, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <int EPI>
__global__ __launch_bounds__(448) void gemm_nt_dma_kernel(const bf16* __restrict__ A, long lda,
                                                          const bf16* __restrict__ W, long ldw, int M, int N, int K,
                                                          GemmEpi e) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
  const int wu = __builtin_amdgcn_readfirstlane(w);
  const long r0 = (long)M * blockIdx.x / gridDim.x, r1 = (long)M * (blockIdx.x + 1) / gridDim.x;
  const int nk = (K + G_BK - 1) / G_BK;
  const u32x4 rA = dma_rsrc(A, (uint32_t)(((long)(M - 1) * lda + K) * 2));
  const u32x4 rW = dma_rsrc(W, (uint32_t)(((long)(N - 1) * ldw + K) * 2));
  const uint32_t lds0 = lds_addr_of(smem);
  char* const wring = smem + GD_NA * GD_ABYTES;

  // per-lane source offsets (k-tile 0); a copy covers 8 image rows: lane -> row 8j + (l>>3), linear chunk l&7,
  // which holds source chunk (l&7) ^ swizzle(row)
  uint32_t av[4], wv[4];
  int kc[4];                // source chunk index of this lane (the same for the A and the W copies)
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int row = 8 * (4 * w + jj) + (l >> 3);
    const int c = swz128(row, l & 7);
    kc[jj] = c;
    av[jj] = (r0 + row < r1) ? (uint32_t)((r0 + row) * lda + 8 * c) * 2 : BUF_OOB;
    wv[jj] = (row < N) ? (uint32_t)((long)row * ldw + 8 * c) * 2 : BUF_OOB;
  }
  // four 1 KB copies per wave and stage, for A and for W; chunks past K arrive as zeros
  auto issue_a = [&](int kt) {
    if (kt >= nk) return;
    const uint32_t ab = lds0 + (kt % GD_NA) * GD_ABYTES, so = (uint32_t)kt * (G_BK * 2);
    const int kleft = (K - kt * G_BK) >> 3;     // valid 16-byte chunks of this k-tile (>= 8 except in the tail)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) dma16_to_lds(rA, ab + 1024 * (4 * wu + jj), kc[jj] < kleft ? av[jj] : BUF_OOB, so);
  };
  auto piece_a = [&](int kt, int jj) {
    if (kt >= nk) return;
    const uint32_t ab = lds0 + (kt % GD_NA) * GD_ABYTES, so = (uint32_t)kt * (G_BK * 2);
    const int kleft = (K - kt * G_BK) >> 3;
    dma16_to_lds(rA, ab + 1024 * (4 * wu + jj), kc[jj] < kleft ? av[jj] : BUF_OOB, so);
  };
  auto piece_w = [&](int kt, int jj) {
    if (kt >= nk) return;
    const uint32_t wb = lds0 + GD_NA * GD_ABYTES + (kt % GD_NW) * GD_WBYTES, so = (uint32_t)kt * (G_BK * 2);
    const int kleft = (K - kt * G_BK) >> 3;
    dma16_to_lds(rW, wb + 1024 * (4 * wu + jj), kc[jj] < kleft ? wv[jj] : BUF_OOB, so);
  };
  auto issue_w = [&](int kt) {
    if (kt >= nk) return;
    const uint32_t wb = lds0 + GD_NA * GD_ABYTES + (kt % GD_NW) * GD_WBYTES, so = (uint32_t)kt * (G_BK * 2);
    const int kleft = (K - kt * G_BK) >> 3;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) dma16_to_lds(rW, wb + 1024 * (4 * wu + jj), kc[jj] < kleft ? wv[jj] : BUF_OOB, so);
  };

  f32x16 acc[GD_NT];
#pragma unroll
  for (int t = 0; t < GD_NT; ++t) acc[t] = zero16();
  int roff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) roff[i] = img_off(l & 31, 16 * i + 8 * (l >> 5));

  // issue order: A0 W0 A1 W1 | A2 W2 | A3 W3 | ...  (copies complete in issue order)
  issue_a(0); issue_w(0); issue_a(1); issue_w(1);
  if (EPI == EPI_F32_BIAS_RES || EPI == EPI_RES_LN) {
    // Y = A W^T + bias + residual: the accumulators START at bias + residual (lane = column, registers = rows; a
    // load instruction covers two rows x 128 contiguous bytes), fetched while the first k stages are in flight.
    // The epilogue then has no loads: a load issued there would wait for the stores of the previous column pair.
    const brsrc rR = make_rsrc(e.resid ? (const void*)e.resid : (const void*)W,
                               e.resid ? (uint32_t)(((long)(M - 1) * e.ldr + N) * 4) : 0u);
    const brsrc rBias = make_rsrc(e.bias ? (const void*)e.bias : (const void*)W, e.bias ? (uint32_t)N * 4 : 0u);
#pragma unroll
    for (int t = 0; t < GD_NT; ++t) {
      const int col = 32 * t + (l & 31);
      const float bv = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rBias, col < N ? col * 4 : (int)BUF_OOB, 0, 0));
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long row = r0 + 32 * w + acc_row(r, l >> 5);
        const uint32_t off = (row < r1 && col < N) ? (uint32_t)(row * e.ldr + col) * 4 : BUF_OOB;
        acc[t][r] = bv + __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rR, (int)off, 0, 0));
      }
    }
    wait_vmcnt<0>();                         // (the initial values come after the copies in issue order)
  } else {
    if (nk > 1) wait_vmcnt<8>(); else wait_vmcnt<0>();   // stage 0 has landed
  }
  __builtin_amdgcn_s_barrier();

  // Fragment registers are double-buffered by k-step: the eight LDS reads of the next k-step are issued in the gaps
  // between the seven products of this one, so no product waits for the LDS.  The last k-step of a stage carries the
  // workgroup barrier: its fragments are already in registers, so both buffers of the stage are free for the copies
  // of later stages, and the reads that follow fetch the next stage's first k-step.  The copies are not issued in a
  // burst (a copy instruction holds the wave's issue for 60-180 cycles) but one per gap between products: the four W
  // copies of stage kt+2 right after the barrier, the four A copies of stage kt+2 at the ends of k-steps 0 and 1.
  // Either measure alone (deeper ring, or spreading with one stage of prefetch) gained nothing; together -12 % at
  // K = 1920.
  bf16x8 fa[2], fw[2][GD_NT];
  auto rd = [&](int set, const char* Aw, const char* Ws, int s2, int i) {   // read i (0..7) of k-step s2
    if (i == 0) fa[set] = *(const bf16x8*)(Aw + roff[s2]);
    else fw[set][i - 1] = *(const bf16x8*)(Ws + 4096 * (i - 1) + roff[s2]);
  };
  {
    const char* Aw = smem + 4096 * w;
#pragma unroll
    for (int i = 0; i < 8; ++i) rd(0, Aw, wring, 0, i);
  }
  constexpr int first[6] = {0, 2, 4, 6, 7, 8};   // reads issued after product t: [first[t], first[t+1])
  for (int kt = 0; kt < nk; ++kt) {
    const char* Aw = smem + (kt % GD_NA) * GD_ABYTES + 4096 * w;
    const char* Ws = wring + (kt % GD_NW) * GD_WBYTES;
    const char* nAw = smem + ((kt + 1) % GD_NA) * GD_ABYTES + 4096 * w;
    const char* nWs = wring + ((kt + 1) % GD_NW) * GD_WBYTES;
#pragma unroll
    for (int s2 = 0; s2 < 3; ++s2) {
#pragma unroll
      for (int t = 0; t < GD_NT; ++t) {
        acc[t] = mfma32(fa[s2 & 1], fw[s2 & 1][t], acc[t]);
        if (t < 5) {
#pragma unroll
          for (int i = first[t]; i < first[t + 1]; ++i) rd((s2 + 1) & 1, Aw, Ws, s2 + 1, i);
        }
        if (s2 < 2 && (t == 5 || t == 6)) piece_a(kt + 2, 2 * s2 + (t - 5));   // A two stages ahead: one per gap
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    acc[0] = mfma32(fa[1], fw[1][0], acc[0]);
    acc[1] = mfma32(fa[1], fw[1][1], acc[1]);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(15 | (7 << 4) | (0 << 8) | (3 << 14));   // lgkmcnt(0): this stage is in registers
    if (kt + 2 < nk) wait_vmcnt<4>(); else wait_vmcnt<0>();                 // stage kt+1 landed (A(kt+2) may fly)
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int t = 2; t < GD_NT; ++t) {
      acc[t] = mfma32(fa[1], fw[1][t], acc[t]);
#pragma unroll
      for (int i = first[t - 2]; i < first[t - 1]; ++i) rd(0, nAw, nWs, 0, i);
      if (t < 6) piece_w(kt + 2, t - 2);     // W two stages ahead, into the buffer this stage has just left
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();                          // all reads of the buffers are done: reuse them as the epilogue stage
  if constexpr (EPI == EPI_LN_BWD) {
    ln_bwd_epilogue(smem, acc, r0, r1, M, N, e);
  } else if constexpr (EPI == EPI_RES_LN) {
    res_ln_epilogue(smem, acc, r0, r1, M, N, e);
  } else {
    float* stage = (float*)smem + w * (32 * G_SP);
    const int mlim = (int)r1;               // rows from r1 on belong to the next workgroup
#pragma unroll
    for (int tp = 0; tp < (GD_NT + 1) / 2; ++tp)
      epilogue_pair<EPI, EPI == EPI_F32_BIAS_RES>(stage, acc[2 * tp], acc[(2 * tp + 1 < GD_NT) ? 2 * tp + 1 : 2 * tp],
                                                  2 * tp + 1 < GD_NT, r0 + 32 * w, 64 * tp, mlim, N, e);
  }
}

// ------------------------------------------------------------------------------------------
// gemm_skinny: Y[M,N] = A[M,K] * W[N,K]^T for K in (192, 208] (the model dim).
//  * the A fragments of a wave's 32 rows (13 k-steps) are loaded from global memory straight into
//    REGISTERS once and stay there; the workgroup walks its range of output columns 64 at a time,
//    streaming the matching W rows through one 32 KB LDS panel;
//  * the product is formed TRANSPOSED (W rows on the MFMA row index, tokens on the lane), and the W rows
//    of a chunk are permuted inside the panel so that every lane ends up holding, for ITS token, 32
//    consecutive output columns: the epilogue (bias, GELU, GELU', bf16 conversion) runs on the accumulators
//    in registers; the 16-byte pieces are then transposed inside each quad of lanes so that a quad stores 64
//    contiguous bytes of one token per instruction -- no LDS round trip for the output.
//  grid = (ceil(M/128), column splits).
// ------------------------------------------------------------------------------------------
constexpr int SK_KS = 13;                 // k-steps held in registers (K <= 208)
constexpr int SK_WBYTES = 4 * 64 * 128;   // one W chunk: 4 k-blocks of [64 rows][64 k] bf16 images
#ifndef SK_AHEAD_STEPS
#define SK_AHEAD_STEPS 2
#endif
constexpr int SK_AHEAD = SK_AHEAD_STEPS;  // k-steps by which the panel reads lead the products

// Output column (0..63 inside a chunk) whose W row sits in panel row pr: tile t = pr>>5, MFMA row i = pr&31 with
// i = 8m + 4h + n  ->  column 32h + 8m + 4t + n.  Accumulator register r = 4m + n of tile t on lane-half h then
// holds that column: a lane ends with 32 consecutive columns (64 bytes of bf16) of its token, as four 16-byte pieces.
SCM_DEV int sk_panel_col(int pr) { return 32 * ((pr >> 2) & 1) + 8 * ((pr >> 3) & 3) + 4 * (pr >> 5) + (pr & 3); }
// byte offset of 16-byte k-chunk `chunk` (0..7) of panel row `row` inside a [64][64] bf16 k-block image: the chunk
// position is XORed with (row >> 1) & 7, which makes the 16-lane groups of ds_read_b128 (rows 0-3, 12-15, 20-27 /
// 4-11, 16-19, 28-31) hit sixteen different 16-byte bank groups.
SCM_DEV int sk_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// 4x4 transpose of 16-byte pieces inside each quad of lanes: on return v[j] of quad lane i holds what v[i] of quad
// lane j held.  With lane = token and v[p] = piece p of the token's 64 bytes, store j then writes, per quad, the 64
// CONTIGUOUS bytes of token (quad base + j): one 64-byte request instead of four 16-byte ones.  The write path
// takes requests, not bytes: with 16-byte pieces the K=200 products were bound by it (QKV: 62 -> 53 us).
SCM_DEV void quad_transpose(u32x4 (&v)[4], int lane) {
  const bool b0 = lane & 1, b1 = lane & 2;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
#pragma unroll
    for (int p = 0; p < 4; p += 2) {      // exchange across lane bit 0: registers p (even) and p+1
      const uint32_t n0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v[p][d], 0xB1, 0xF, 0xF, true);       // lane ^ 1
      const uint32_t n1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v[p + 1][d], 0xB1, 0xF, 0xF, true);
      v[p][d] = b0 ? n1 : v[p][d];
      v[p + 1][d] = b0 ? v[p + 1][d] : n0;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {         // exchange across lane bit 1: registers p and p+2
      const uint32_t n0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v[p][d], 0x4E, 0xF, 0xF, true);       // lane ^ 2
      const uint32_t n2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v[p + 2][d], 0x4E, 0xF, 0xF, true);
      v[p][d] = b1 ? n2 : v[p][d];
      v[p + 2][d] = b1 ? v[p + 2][d] : n0;
    }
  }
}

// Loads that the epilogue waits for must not be issued after stores: memory operations retire in issue order, so
// such a load would wait for the stores of the previous item.  Hence the bias vector is copied to LDS once, and the
// pre-activation rows of the GELU' epilogue are fetched one item ahead, before the current item's stores.
// The W panel is double-buffered in LDS (two workgroups per CU; a third one was measured to add nothing): the
// chunk of item i+2 is copied global->LDS asynchronously into the buffer item i has just left, one barrier per item.
template <bool A_F32, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_skinny_kernel(const void* __restrict__ Ap, long lda,
                                                             const bf16* __restrict__ W, long ldw, int M, int N,
                                                             int K, GemmEpi e) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, half = l >> 5;
  constexpr uint32_t AE = A_F32 ? 4 : 2;

  // Work items = (128-row tile, 64-column chunk) pairs in row-major order, dealt out in equal contiguous
  // ranges to a grid that just fills the machine: no tail round, no CU with twice the work of another.
  const int nchunks = (N + 63) >> 6;
  const long total = (long)((M + G_BM - 1) / G_BM) * nchunks;
  const int it_lo = (int)(total * blockIdx.x / gridDim.x), it_hi = (int)(total * (blockIdx.x + 1) / gridDim.x);
  if (it_lo >= it_hi) return;

  // every global access goes through a bounds-checked buffer resource: out-of-range lanes pass BUF_OOB
  const brsrc rA = make_rsrc(Ap, (uint32_t)(((long)(M - 1) * lda + K) * AE));
  const brsrc rO0 = make_rsrc(e.out0, (uint32_t)(((long)(M - 1) * e.ldo0 + N) * 2));
  const brsrc rO1 = make_rsrc(EPI == EPI_GELU_DUAL ? e.out1 : e.out0,
                              (uint32_t)(((long)(M - 1) * (EPI == EPI_GELU_DUAL ? e.ldo1 : e.ldo0) + N) * 2));
  const brsrc rP = make_rsrc(EPI == EPI_DGELU ? (const void*)e.pre : (const void*)W,
                             EPI == EPI_DGELU ? (uint32_t)(((long)(M - 1) * e.ldp + N) * 2) : 0u);
  const bool has_bias = (EPI == EPI_GELU_DUAL || EPI == EPI_BF16_BIAS) && e.bias != nullptr;
  const brsrc rB = make_rsrc(has_bias ? (const void*)e.bias : (const void*)W, has_bias ? (uint32_t)N * 4 : 0u);

  float* bias_s = (float*)(smem + 2 * SK_WBYTES);  // [64 * nchunks] f32 (zeros where there is no bias)
  if (EPI == EPI_GELU_DUAL || EPI == EPI_BF16_BIAS)
    for (int i = tid; i < 16 * nchunks; i += 256) *(u32x4*)(bias_s + 4 * i) = buf_load16(rB, (uint32_t)i * 16);
  int roff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) roff[i] = sk_off(l & 31, 2 * i + half);

  // W staging: asynchronous global->LDS copies, 32 of 1 KB per chunk, instruction 4 j + w by wave w.  A copy fills
  // LDS linearly in lane order, so each lane fetches the 16 bytes that BELONG at its slot: slot (panel row pr,
  // position cp) of k-block j>>1 holds k-chunk cp ^ swizzle(pr) of W row n0 + sk_panel_col(pr).  Rows from N on and
  // k-chunks from K/8 on are requested out of range and arrive as zeros.
  const u32x4 rWd = dma_rsrc(W, (uint32_t)(((long)(N - 1) * ldw + K) * 2));
  const int wu = __builtin_amdgcn_readfirstlane(w);
  const uint32_t lds0 = lds_addr_of(smem);
  const int colA = sk_panel_col(8 * w + (l >> 3));                // panel rows 8w + (l>>3) (j even) and 32 + that (j odd: + 4)
  const int chw = (l & 7) ^ ((4 * w + (l >> 4)) & 7);             // k-chunk inside the k-block
  const uint32_t wbase = (uint32_t)(colA * ldw + 8 * chw) * 2;
  const bool k3ok = 192 + 8 * chw < K;
  auto issue_w = [&](int it, int buf) {
    if (it >= it_hi) return;
    const int n0 = 64 * (it % nchunks);
    const uint32_t vb = wbase + (uint32_t)(n0 * ldw) * 2;
    const bool r0ok = n0 + colA < N, r1ok = n0 + colA + 4 < N;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool ok = ((j & 1) ? r1ok : r0ok) && (j < 6 || k3ok);
      dma16_to_lds(rWd, lds0 + buf * SK_WBYTES + 1024 * (4 * j + wu),
                   ok ? vb + (uint32_t)((j & 1) * 4 * ldw) * 2 + (j >> 1) * 128 : BUF_OOB, 0);
    }
  };

  // A fragments of a row tile: global -> registers (lane: row l&31, k = 16 ks + 8 half .. +7).  Issued right
  // after the last MFMA of the previous tile, so the latency hides behind that item's epilogue.
  bf16x8 af[SK_KS];
  long row = 0;
  auto load_a = [&](int tile) {
    row = (long)tile * G_BM + 32 * w + (l & 31);
    const uint32_t base = (row < M ? (uint32_t)(row * lda) * AE : BUF_OOB) + 8 * half * AE;
#pragma unroll
    for (int ks = 0; ks < SK_KS; ++ks) {
      // k = 16 ks + 8 half < K for every ks < 12 (K > 192); the last step's upper half exists only for K = 208
      const uint32_t off = (ks < SK_KS - 1 || half == 0 || K > 200) ? base + 16 * ks * AE : BUF_OOB;
      if (A_F32) {
        const f32x4 x0 = as_f32x4(buf_load16(rA, off)), x1 = as_f32x4(buf_load16(rA, off + 16));
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = (bf16)x0[j]; v[4 + j] = (bf16)x1[j]; }
        af[ks] = v;
        if (ks & 1) __builtin_amdgcn_sched_barrier(0);   // at most 4 raw f32 loads (16 registers) in flight
      } else {
        af[ks] = as_bf16x8(buf_load16(rA, off));
      }
    }
  };
  load_a(it_lo / nchunks);
  // pre-activation rows of an item (GELU' epilogue): lane = its token, 4 groups of 8 columns
  u32x4 pv[4];
  auto load_pre = [&](int item, u32x4 (&dst)[4]) {
    const int tl = item / nchunks, cc = item - tl * nchunks;
    const long r = (long)tl * G_BM + 32 * w + (l & 31);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int col = 64 * cc + 32 * half + 8 * g;
      dst[g] = buf_load16(rP, (item < it_hi && r < M && col < N) ? ((uint32_t)(r * e.ldp) + (uint32_t)col) * 2 : BUF_OOB);
    }
  };
  if (EPI == EPI_DGELU) load_pre(it_lo, pv);
  issue_w(it_lo, 0);
  issue_w(it_lo + 1, 1);
  // Everything issued so far is waited for HERE, once.  Left to the compiler, the wait for the first row tile's A
  // fragments lands inside the loop (its entry from above and its back edge are merged conservatively) and every
  // item would then wait for the previous item's output stores to complete before its third product.
  wait_vmcnt<0>();
  __syncthreads();
  constexpr int NST = EPI == EPI_GELU_DUAL ? 8 : 4;   // output stores per item
  bf16x8 wf[SK_AHEAD + 1][2];             // ring of panel fragments, indexed by k-step mod (SK_AHEAD + 1)
  auto read_w = [&](const char* Wp, int ks) {
    const char* blk = Wp + (ks >> 2) * (64 * 128);
    wf[ks % (SK_AHEAD + 1)][0] = *(const bf16x8*)(blk + roff[ks & 3]);
    wf[ks % (SK_AHEAD + 1)][1] = *(const bf16x8*)(blk + 4096 + roff[ks & 3]);
  };
#pragma unroll
  for (int ks = 0; ks < SK_AHEAD; ++ks) read_w(smem, ks);
  for (int it = it_lo; it < it_hi; ++it) {
    const int tile = it / nchunks, c = it - tile * nchunks;
    const long orow = row;          // the epilogue below still belongs to this item's tile
    const char* Ws = smem + ((it - it_lo) & 1) * SK_WBYTES;
    f32x16 acc0 = zero16(), acc1 = zero16();
    // The panel fragments run SK_AHEAD k-steps ahead of the products in a register ring, so a product never waits
    // for the LDS; the ring's first steps were read at the end of the previous item, behind its epilogue.
#pragma unroll
    for (int ks = 0; ks < SK_KS; ++ks) {
      if (ks + SK_AHEAD < SK_KS) read_w(Ws, ks + SK_AHEAD);
      acc0 = mfma32(wf[ks % (SK_AHEAD + 1)][0], af[ks], acc0);
      acc1 = mfma32(wf[ks % (SK_AHEAD + 1)][1], af[ks], acc1);
      __builtin_amdgcn_sched_barrier(0);
    }
    // The copies of the next item's chunk were issued one item ago, ahead of that item's NST output stores; memory
    // operations complete in issue order, so "all but the last NST" means they have landed while the stores may
    // still be in flight.
    wait_vmcnt<NST>();
    __syncthreads();                       // next item's chunk visible to all; this item's buffer is free
#pragma unroll
    for (int ks = 0; ks < SK_AHEAD; ++ks) read_w(smem + ((it - it_lo + 1) & 1) * SK_WBYTES, ks);
    if (c + 1 == nchunks && it + 1 < it_hi) load_a(tile + 1);
    // ---- epilogue on registers: lane = token `orow`, group g = columns col .. col+7; straight-line
    const int colb = 64 * c + 32 * half;
    uint32_t bias_off[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bias_off[g] = (uint32_t)(colb + 8 * g) * 4;
    // after the in-quad transpose, store j carries piece (l & 3) of token (orow & ~3) + j
    const int scol = colb + 8 * (l & 3);
    const long srow = orow & ~3l;
    u32x4 npv[4];
    if (EPI == EPI_DGELU) load_pre(it + 1, npv);   // ahead of this item's stores
    issue_w(it + 2, (it - it_lo) & 1);             // last among the loads: only the stores below follow it
    u32x4 ov[4], gv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float v[8] = {acc0[4 * g], acc0[4 * g + 1], acc0[4 * g + 2], acc0[4 * g + 3],
                    acc1[4 * g], acc1[4 * g + 1], acc1[4 * g + 2], acc1[4 * g + 3]};
      if (EPI == EPI_GELU_DUAL || EPI == EPI_BF16_BIAS) {   // a null bias reads zeros
        const f32x4 b0 = *(const f32x4*)(bias_s + (bias_off[g] >> 2)), b1 = *(const f32x4*)(bias_s + (bias_off[g] >> 2) + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] += b0[j]; v[4 + j] += b1[j]; }
      }
      bf16x8 o;
      if (EPI == EPI_DGELU) {
        const bf16x8 pb = as_bf16x8(pv[g]);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)(v[j] * gelu_erf_grad((float)pb[j]));
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)v[j];
      }
      ov[g] = as_u32x4(o);
      if (EPI == EPI_GELU_DUAL) {
        bf16x8 gl;
#pragma unroll
        for (int j = 0; j < 8; ++j) gl[j] = (bf16)gelu_erf(v[j]);
        gv[g] = as_u32x4(gl);
      }
    }
    // each tensor's four stores back to back: the two 64-byte halves of a 128-byte line come from lanes l, l+32
    quad_transpose(ov, l);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      buf_store16(rO0, (srow + j < M && scol < N) ? ((uint32_t)((srow + j) * e.ldo0) + (uint32_t)scol) * 2 : BUF_OOB, ov[j]);
    if (EPI == EPI_GELU_DUAL) {
      quad_transpose(gv, l);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        buf_store16(rO1, (srow + j < M && scol < N) ? ((uint32_t)((srow + j) * e.ldo1) + (uint32_t)scol) * 2 : BUF_OOB, gv[j]);
    }
    if (EPI == EPI_DGELU) {
#pragma unroll
      for (int g = 0; g < 4; ++g) pv[g] = npv[g];
    }
  }
}

// ------------------------------------------------------------------------------------------
// wgrad: C[P,Q<=224] = sum over tokens of Pm[t,P]^T Qm[t,Q]
// ------------------------------------------------------------------------------------------
constexpr int W_BP = 160;       // P columns per workgroup (5 waves x 32)
constexpr int W_BQ = 224;       // Q columns (7 tiles), Q <= 223 real + optional ones column
constexpr int W_BT = 64;        // tokens per k-tile
constexpr int W_PP = W_BP * 2;  // 320 B row pitch  (320 mod 256 = 64: transposed reads conflict-free)
constexpr int W_QP = W_BQ * 2;  // 448 B row pitch  (448 mod 256 = 192)

SCM_DEV bf16x8 tr_frag_pitch(const char* tile, int pitch, int kbase, int col_base) {
  const int l = lane_id();
  const int g = l >> 4, i = l & 15, half = g >> 1;
  const int col = col_base + 16 * (g & 1) + 4 * (i & 3);
  const int r0 = kbase + 8 * half + (i >> 2);
  bf16x4 lo = lds_tr_read(tile + r0 * pitch + col * 2);
  bf16x4 hi = lds_tr_read(tile + (r0 + 4) * pitch + col * 2);
  bf16x8 f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}

// Pm: bf16 [T, P] (ldp), Qm: bf16 or f32 [T, Q] (ldq), Q % 8 == 0, Q <= 216.
// ONES: column Q of the Q tile is 1 for valid tokens, so C[:,Q] = column sums of Pm (bias grad).
template <bool Q_F32, bool ONES>
__global__ __launch_bounds__(320) void wgrad_kernel(const bf16* __restrict__ Pm, long ldp, const void* __restrict__ Qm,
                                                    long ldq, int T, int P, int Q, int tiles_per_slice,
                                                    float* __restrict__ slab) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ps = smem;                   // [64][160] bf16
  char* Qs = smem + W_BT * W_PP;     // [64][224] bf16
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, half = l >> 5;
  // 1-D grid of ptiles*slices blocks.  Blocks b and b+8 share an XCD (round-robin dispatch): the P tiles
  // of one token slice are given ids 8 apart so that they run on one XCD and read the slice's Q-operand
  // tile (common to all of them) through that XCD's L2.  Speed only: any placement is correct.
  const int ptiles = (P + W_BP - 1) / W_BP;
  int slice, ptile;
  {
    const int b = blockIdx.x, nslices = gridDim.x / ptiles, full = (nslices / 8) * 8 * ptiles;
    if (b < full) { slice = (b % 8) + 8 * (b / (8 * ptiles)); ptile = (b / 8) % ptiles; }
    else { slice = (nslices / 8) * 8 + (b - full) / ptiles; ptile = (b - full) % ptiles; }
  }
  const int p0 = ptile * W_BP;
  const int ntiles = (T + W_BT - 1) / W_BT;
  const int tile_lo = slice * tiles_per_slice;
  const int tile_hi = min(ntiles, tile_lo + tiles_per_slice);
  const int qchunks = Q >> 3;        // 16-byte bf16 chunks per Q row

  f32x16 acc[7];
#pragma unroll
  for (int t = 0; t < 7; ++t) acc[t] = zero16();

  // zero the padding columns of the Q tile once; the ones column is rewritten per tile
  for (int i = tid; i < W_BT * (W_BQ / 8); i += 320) {
    const int row = i / (W_BQ / 8), ch = i % (W_BQ / 8);
    if (ch >= qchunks) {
      bf16x8 z;
#pragma unroll
      for (int j = 0; j < 8; ++j) z[j] = (bf16)0.f;
      *(bf16x8*)(Qs + row * W_QP + ch * 16) = z;
    }
  }

  bf16x8 pst[4];
  bf16x8 qst[5];
  auto load_stage = [&](int tile) {
    const int t0 = tile * W_BT;
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // 64 rows x 20 chunks = 1280 = 4 x 320
      const int c = tid + 320 * i, row = c / 20, ch = c % 20;
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (bf16)0.f;
      if (t0 + row < T && p0 + 8 * ch < P) v = *(const bf16x8*)(Pm + (long)(t0 + row) * ldp + p0 + 8 * ch);
      pst[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {  // 64 rows x qchunks(<=25..27) chunks
      const int c = tid + 320 * i;
      const int row = c / qchunks, ch = c % qchunks;
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (bf16)0.f;
      if (row < W_BT && t0 + row < T) {
        if (Q_F32) {
          const float* p = (const float*)Qm + (long)(t0 + row) * ldq + 8 * ch;
          f32x4 x0 = *(const f32x4*)p, x1 = *(const f32x4*)(p + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { v[j] = (bf16)x0[j]; v[4 + j] = (bf16)x1[j]; }
        } else {
          v = *(const bf16x8*)((const bf16*)Qm + (long)(t0 + row) * ldq + 8 * ch);
        }
      }
      qst[i] = v;
    }
  };
  auto write_stage = [&](int tile) {
    const int t0 = tile * W_BT;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 320 * i, row = c / 20, ch = c % 20;
      *(bf16x8*)(Ps + row * W_PP + ch * 16) = pst[i];
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int c = tid + 320 * i;
      const int row = c / qchunks, ch = c % qchunks;
      if (row < W_BT) *(bf16x8*)(Qs + row * W_QP + ch * 16) = qst[i];
    }
    if (ONES && tid < W_BT) {
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (bf16)0.f;
      o[0] = (bf16)((t0 + tid < T) ? 1.f : 0.f);
      *(bf16x8*)(Qs + tid * W_QP + qchunks * 16) = o;
    }
  };

  if (tile_lo < tile_hi) load_stage(tile_lo);
  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    __syncthreads();  // previous tile's reads (and the initial zero fill) are done
    write_stage(tile);
    __syncthreads();
    if (tile + 1 < tile_hi) load_stage(tile + 1);
#pragma unroll
    for (int s = 0; s < W_BT / 16; ++s) {
      bf16x8 a = tr_frag_pitch(Ps, W_PP, 16 * s, 32 * w);
#pragma unroll
      for (int t = 0; t < 7; ++t) {
        bf16x8 b = tr_frag_pitch(Qs, W_QP, 16 * s, 32 * t);
        acc[t] = mfma32(a, b, acc[t]);
      }
    }
  }

  float* out = slab + (long)slice * P * W_LDSLAB;
  const int qlim = ONES ? Q + 1 : Q;
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    const int col = 32 * t + (l & 31);
    if (col >= qlim) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int p = p0 + 32 * w + acc_row(r, half);
      if (p < P) out[(long)p * W_LDSLAB + col] = acc[t][r];
    }
  }
}

// ------------------------------------------------------------------------------------------
// wgrad_dma: the same product for bf16 operands, fed by asynchronous global->LDS copies.
//   workgroup = 4 waves (one per SIMD), 128 P columns x up to 224 Q columns, 64-token stages in a ring of 3:
//   while stage i is multiplied, the copies of stages i+1 and i+2 are in flight (no staging registers, one
//   barrier per stage, placed inside the stage's last k-step).  P stage: [64][128] bf16, 256-byte rows, 64-byte units XOR-swizzled by (row & 3) so that
//   the transposing reads are conflict free; Q stage: [64][224] bf16 at a 448-byte pitch.
// ------------------------------------------------------------------------------------------
constexpr int WD_BP = 128;
constexpr int WD_PBYTES = W_BT * 256;             // 16 KB
constexpr int WD_QBYTES = W_BT * W_QP;            // 28 KB
constexpr int WD_STAGE = WD_PBYTES + WD_QBYTES;   // 44 KB
constexpr int WD_NSTAGE = 3;

// One launch may carry up to four products over the same tokens (the weight gradients of a layer): their P tiles
// are dealt out as those of one wide product, every workgroup picks its product from its tile number.  With fewer
// token slices per product the f32 slabs (written here, read back by the reduction) shrink in proportion, and the
// launch's fixed part (first stage, slab epilogue, drain: about 12 us) is paid once.
struct WgradOne { const bf16* Pm; long ldp; const bf16* Qm; long ldq; float* slab; int P, Q, ones; };
constexpr int WD_MAXN = 4;
struct WgradPair { WgradOne p[WD_MAXN]; int n; };

template <bool ONES>
__global__ __launch_bounds__(256) void wgrad_dma_kernel(WgradPair pr, int T, int tiles_per_slice) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, half = l >> 5;
  int ptiles = 0;
#pragma unroll
  for (int i = 0; i < WD_MAXN; ++i) ptiles += i < pr.n ? (pr.p[i].P + WD_BP - 1) / WD_BP : 0;
  int slice, ptile;
  {  // blocks b and b+8 share an XCD: the P tiles of one token slice run on one XCD and share the Q stage in its L2
    const int b = blockIdx.x, nslices = gridDim.x / ptiles, full = (nslices / 8) * 8 * ptiles;
    if (b < full) { slice = (b % 8) + 8 * (b / (8 * ptiles)); ptile = (b / 8) % ptiles; }
    else { slice = (nslices / 8) * 8 + (b - full) / ptiles; ptile = (b - full) % ptiles; }
  }
  int which = 0;                              // workgroup-uniform: the product this tile number falls into
#pragma unroll
  for (int i = 0; i < WD_MAXN - 1; ++i) {
    const int pt = (pr.p[i].P + WD_BP - 1) / WD_BP;
    if (which == i && i + 1 < pr.n && ptile >= pt) { ptile -= pt; which = i + 1; }
  }
  const WgradOne& me = pr.p[which];
  const bf16* __restrict__ Pm = me.Pm;
  const bf16* __restrict__ Qm = me.Qm;
  const long ldp = me.ldp, ldq = me.ldq;
  const int P = me.P, Q = me.Q;
  float* __restrict__ slab = me.slab;
  const bool ones = ONES && me.ones;
  const int p0 = ptile * WD_BP;
  const int ntiles = (T + W_BT - 1) / W_BT;
  const int tile_lo = slice * tiles_per_slice;
  const int tile_hi = min(ntiles, tile_lo + tiles_per_slice);
  const int qchunks = Q >> 3;
  // The resources start at the slice's first token and end with the tensor, or just below 2 GiB, whichever comes
  // first (a slice is shorter than that; BUF_OOB must stay out of range): offsets stay 32-bit however many tokens
  // there are.
  const long t_lo = (long)tile_lo * W_BT;
  const long p_left = ((long)(T - 1) - t_lo) * ldp + P, q_left = ((long)(T - 1) - t_lo) * ldq + Q;
  const u32x4 rP = dma_rsrc(Pm + t_lo * ldp, (uint32_t)(p_left <= 0 ? 0 : (p_left * 2 < 0x7FFFF000l ? p_left * 2 : 0x7FFFF000l)));
  const u32x4 rQ = dma_rsrc(Qm + t_lo * ldq, (uint32_t)(q_left <= 0 ? 0 : (q_left * 2 < 0x7FFFF000l ? q_left * 2 : 0x7FFFF000l)));
  const int wu = __builtin_amdgcn_readfirstlane(w);   // wave index as a scalar (LDS copy bases live in M0)
  const uint32_t lds0 = lds_addr_of(smem);

  // per-lane source offsets of this wave's copies (relative to the stage's first token)
  uint32_t pv[4], qv[7];
  bool qmask[7];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {        // P: copy j = 4 w + jj fills rows 4j .. 4j+3 (1 KB)
    const int row = 4 * (4 * w + jj) + (l >> 4);
    const int chunk = (l & 15) ^ ((row & 3) << 2);
    pv[jj] = (p0 + 8 * chunk < P) ? (uint32_t)(row * ldp + p0 + 8 * chunk) * 2 : BUF_OOB;
  }
#pragma unroll
  for (int jj = 0; jj < 7; ++jj) {        // Q: copy j = 7 w + jj fills bytes [1024 j, 1024 j + 1024) of the stage
    const int off = 1024 * (7 * w + jj) + 16 * l;
    const int row = off / W_QP, chunk = (off - row * W_QP) >> 4;
    qv[jj] = chunk < qchunks ? (uint32_t)(row * ldq + 8 * chunk) * 2 : BUF_OOB;   // padding chunks receive zeros
    qmask[jj] = !(ones && chunk == qchunks);                                       // the ones column is kept
  }
  // copy number i (0..10) of this wave for a stage: 0..3 fill the P image, 4..10 the Q image.  A copy instruction
  // holds the wave's issue for 60-180 cycles, so the eleven copies of a stage are not issued in one burst: they are
  // placed one by one in the gaps between the products (a burst after the barrier cost as much as the stage's
  // products themselves).
  auto issue_piece = [&](int tile, int buf, int i) {
    if (tile >= tile_hi) return;          // past the slice: nothing is copied (the stage is not read either)
    const uint32_t pb = lds0 + buf * WD_STAGE, qb = pb + WD_PBYTES;
    if (i < 4) dma16_to_lds(rP, pb + 1024 * (4 * wu + i), pv[i], (uint32_t)((long)(tile - tile_lo) * W_BT * ldp * 2));
    else if (!ONES || qmask[i - 4]) dma16_to_lds(rQ, qb + 1024 * (7 * wu + i - 4), qv[i - 4], (uint32_t)((long)(tile - tile_lo) * W_BT * ldq * 2));
  };
  auto issue = [&](int tile, int buf, int i0, int i1) {
#pragma unroll
    for (int i = i0; i < i1; ++i) issue_piece(tile, buf, i);
  };
  if (ONES && ones) {   // column Q of every stage = 1 (tokens past T multiply zero-filled P rows)
    for (int i = tid; i < WD_NSTAGE * W_BT; i += 256) {
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (bf16)0.f;
      o[0] = (bf16)1.f;
      *(bf16x8*)(smem + (i / W_BT) * WD_STAGE + WD_PBYTES + (i % W_BT) * W_QP + qchunks * 16) = o;
    }
  }

  f32x16 acc[7];
#pragma unroll
  for (int t = 0; t < 7; ++t) acc[t] = zero16();

  // fragment addresses: k runs over stage rows (tokens), read with the transposing LDS read
  const int g = l >> 4, i16 = l & 15;
  const int frow = 8 * (g >> 1) + (i16 >> 2);                 // + 16 s (+4 for the upper half of the fragment)
  const int acol = 32 * w + 16 * (g & 1) + 4 * (i16 & 3);     // P column of this lane's 4-element group
  const int a_off = frow * 256 + ((((acol >> 3) ^ ((frow & 3) << 2))) << 4) + (acol & 7) * 2;
  const int b_off = frow * W_QP + (16 * (g & 1) + 4 * (i16 & 3)) * 2;

  // Fragment registers are double-buffered by k-step: while the seven products of one k-step run, the sixteen
  // transposing reads of the next one are issued in the gaps between them (at most four per gap), so every read has
  // several products' time to return and no product waits on the LDS.  The step that ends a stage carries the
  // workgroup barrier: its own fragments are already in registers, so the stage's buffer is free for the copy of
  // the stage three ahead, and the reads that follow the barrier fetch the first k-step of the next stage.
  bf16x4 fa[2][2], fb[2][7][2];
  auto rd = [&](int set, const char* pb, int s, int i) {   // read number i (0..15) of k-step s of the stage at pb
    const char* qb = pb + WD_PBYTES;
    if (i < 2) fa[set][i] = lds_tr_read(pb + a_off + s * 4096 + i * 1024);
    else {
      const int t = (i - 2) >> 1, h = (i - 2) & 1;
      fb[set][t][h] = lds_tr_read(qb + b_off + s * (16 * W_QP) + t * 64 + h * 4 * W_QP);
    }
  };
  auto mm = [&](int set, int t) {
    const bf16x8 a = {fa[set][0][0], fa[set][0][1], fa[set][0][2], fa[set][0][3],
                      fa[set][1][0], fa[set][1][1], fa[set][1][2], fa[set][1][3]};
    const bf16x8 b = {fb[set][t][0][0], fb[set][t][0][1], fb[set][t][0][2], fb[set][t][0][3],
                      fb[set][t][1][0], fb[set][t][1][1], fb[set][t][1][2], fb[set][t][1][3]};
    acc[t] = mfma32(a, b, acc[t]);
  };
  // Issue order of the copies: stage 0, stage 1 and the first NEARLY pieces of stage 2 before the loop; then, per
  // stage t, the remaining pieces of stage t+2 (k-steps 0..2) and, after the barrier, the first pieces of stage t+3.  Copies complete in issue order, so "stage t+1 has landed" is a counted wait that
  // leaves the eleven copies issued after it in flight.
  constexpr int NPC = 11, NEARLY = 5;     // copies per stage; those issued in the stage's last k-step, after the barrier
#ifdef SCM_STAMPS   // diagnostic build only (tools/wgrad_stamps.py): shader-clock and 100 MHz stamps around the phases,
                    // written BEHIND the slabs, so the workspace must be 64 bytes per wave larger than the product needs
  const unsigned long long st0 = __builtin_amdgcn_s_memtime(), sr0 = __builtin_amdgcn_s_memrealtime();
#endif
  issue(tile_lo, 0, 0, NPC);
  issue(tile_lo + 1, 1, 0, NPC);
  issue(tile_lo + 2, 2, 0, NEARLY);
  if (tile_lo + 2 < tile_hi) wait_vmcnt<NPC + NEARLY>(); else if (tile_lo + 1 < tile_hi) wait_vmcnt<NPC>(); else wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < 16; ++i) rd(0, smem, 0, i);
  int buf = 0;
#ifdef SCM_STAMPS
  const unsigned long long st1 = __builtin_amdgcn_s_memtime(), sr1 = __builtin_amdgcn_s_memrealtime();
#endif
  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    const char* pb = smem + buf * WD_STAGE;
    const int nbuf = buf == 2 ? 0 : buf + 1, pbuf = buf == 0 ? 2 : buf - 1;
    const char* npb = smem + nbuf * WD_STAGE;
#pragma unroll
    for (int s = 0; s < 3; ++s) {          // k-steps 0..2: reads of step s+1 between the products of step s
      constexpr int first[6] = {0, 4, 7, 10, 13, 16};
#pragma unroll
      for (int t = 0; t < 7; ++t) {
        mm(s & 1, t);
        if (t < 5) {
#pragma unroll
          for (int i = first[t]; i < first[t + 1]; ++i) rd((s + 1) & 1, pb, s + 1, i);
        }
        // the other six pieces of stage tile+2 (its buffer was left at the barrier of stage tile-1): two at the end
        // of each k-step (of the placements tried, the one with the shortest launch)
        if (t == 5) issue_piece(tile + 2, pbuf, NEARLY + 2 * s);
        if (t == 6) issue_piece(tile + 2, pbuf, NEARLY + 2 * s + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // k-step 3 (set 1): two products, then the barrier, then the rest with the reads of the next stage's first
    // k-step (set 0) and the first pieces of stage tile+3 (into the buffer this stage has just left)
    mm(1, 0);
    mm(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(15 | (7 << 4) | (0 << 8) | (3 << 14));   // lgkmcnt(0): this stage is in registers
    if (tile + 2 < tile_hi) wait_vmcnt<NPC>(); else wait_vmcnt<0>();        // my copies of stage tile+1
    __builtin_amdgcn_s_barrier();
    {
      constexpr int first[6] = {0, 4, 7, 10, 13, 16};
#pragma unroll
      for (int t = 2; t < 7; ++t) {
        mm(1, t);
#pragma unroll
        for (int i = first[t - 2]; i < first[t - 1]; ++i) rd(0, npb, 0, i);
        issue_piece(tile + 3, buf, t - 2);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    buf = nbuf;
  }

#ifdef SCM_STAMPS
  const unsigned long long st2 = __builtin_amdgcn_s_memtime(), sr2 = __builtin_amdgcn_s_memrealtime();
#endif
  float* out = slab + (long)slice * P * W_LDSLAB;
  const int qlim = ones ? Q + 1 : Q;
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    const int col = 32 * t + (l & 31);
    if (col >= qlim) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int p = p0 + 32 * w + acc_row(r, half);
      if (p < P) out[(long)p * W_LDSLAB + col] = acc[t][r];
    }
  }
#ifdef SCM_STAMPS
  wait_vmcnt<0>();
  const unsigned long long st3 = __builtin_amdgcn_s_memtime();
  if (l == 0) {
    unsigned long long* dbg = (unsigned long long*)(slab + (long)(gridDim.x / ptiles) * P * W_LDSLAB) + ((long)blockIdx.x * 4 + w) * 8;
    dbg[0] = st1 - st0; dbg[1] = st2 - st1; dbg[2] = st3 - st2; dbg[3] = sr2 - sr1; dbg[4] = tile_hi - tile_lo; dbg[5] = sr0; dbg[6] = sr2;
  }
#endif
}

// grad (+)= sum_s slab[s][p][q];  trans: grad is [Q,P] (grad[q*P+p]) else [P,Q].
// bias (+)= sum_s slab[s][p][Q]  when bias != null.
__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, int S, int P, int Q, float* __restrict__ grad,
                                    float* __restrict__ bias, int trans, int accumulate) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int qn = bias ? Q + 1 : Q;
  if (idx >= P * qn) return;
  const int p = idx / qn, q = idx % qn;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const float* sp = slab + (long)p * W_LDSLAB + q;
  const long st = (long)P * W_LDSLAB;
  int i = 0;
  for (; i + 3 < S; i += 4) {   // four independent loads in flight; fixed summation order
    s0 += sp[i * st]; s1 += sp[(i + 1) * st]; s2 += sp[(i + 2) * st]; s3 += sp[(i + 3) * st];
  }
  for (; i < S; ++i) s0 += sp[i * st];
  const float s = (s0 + s1) + (s2 + s3);
  if (q == Q) {
    bias[p] = accumulate ? bias[p] + s : s;
  } else {
    const long o = trans ? (long)q * P + p : (long)p * Q + q;
    grad[o] = accumulate ? grad[o] + s : s;
  }
}

// dst_bf16 = src_f32 (plain) and dstT_bf16 = transpose(src) for a [R,C] weight.
__global__ void cast_weight_kernel(const float* __restrict__ src, int R, int C, bf16* __restrict__ dst,
                                   bf16* __restrict__ dstT) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= R * C) return;
  const int r = idx / C, c = idx % C;
  const bf16 v = (bf16)src[idx];
  if (dst) dst[idx] = v;
  if (dstT) dstT[(long)c * R + r] = v;
}

// "Chain order" of the k index of a weight that is the SECOND operand of a chained product (ffn_chain.hip): inside
// every block of 64, unit u = 32 s + 16 e + 8 h + 4 t + i sits at position 32 t + 16 s + 8 h + 4 e + i, the order in
// which the first product's accumulator registers (tile t, registers 8s..8s+7, lane-half h) come out as the bf16
// B fragment of k-step 2t + s.
__host__ __device__ inline int chain_pos(int u) {
  const int b = u & ~63, v = u & 63;
  return b | (((v >> 2) & 1) << 5) | (((v >> 5) & 1) << 4) | (((v >> 3) & 1) << 3) | (((v >> 4) & 1) << 2) | (v & 3);
}

// The same for a whole table of weights in one launch (grid.y = entry): an entry casts src [R,C] into
// dst[r * ldd + c] and dstT[c * ldt + r], so several sources can fill slices of one fused matrix.
// chain bit 0: dst columns in chain order; bit 1: dstT columns (the source's rows) in chain order.
__global__ void cast_weights_batch_kernel(const CastItem* __restrict__ items) {
  const CastItem it = items[blockIdx.y];
  const long n = (long)it.R * it.C;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    const int r = (int)(idx / it.C), c = (int)(idx % it.C);
    const bf16 v = (bf16)it.src[idx];
    if (it.dst) ((bf16*)it.dst)[(long)r * it.ldd + ((it.chain & 1) ? chain_pos(c) : c)] = v;
    if (it.dstT) ((bf16*)it.dstT)[(long)c * it.ldt + ((it.chain & 2) ? chain_pos(r) : r)] = v;
  }
}

}  // namespace scm

// ------------------------------------------------------------------------------------------
// host launchers (C linkage wrappers live in capi.hip)
// ------------------------------------------------------------------------------------------
namespace scm {

// algorithmic HBM bytes of one product: A and W read once, every output written once, the epilogue's extra
// operands (residual, pre-activation) read once
template <bool A_F32, int EPI>
static double gemm_bytes(int M, int N, int K, const GemmEpi& e) {
  const bool f32_out = EPI == EPI_F32 || EPI == EPI_F32_BIAS_RES;
  double b = (double)M * K * (A_F32 ? 4 : 2) + (double)N * K * 2 + (double)M * N * (f32_out ? 4 : 2);
  if (EPI == EPI_GELU_DUAL) b += (double)M * N * 2;
  if (EPI == EPI_DGELU) b += (double)M * N * 2;
  if (EPI == EPI_F32_BIAS_RES && e.resid) b += (double)M * N * 4;
  return b;
}

// products that ran on the general register-staged kernel (shapes none of the asynchronous-copy kernels takes)
static std::atomic<long long> g_general_launches{0};
long long gemm_general_launches() { return g_general_launches.load(std::memory_order_relaxed); }

template <int NT, bool A_F32, int EPI>
static hipError_t launch_gemm(const void* A, long lda, const bf16* W, long ldw, int M, int N, int K, GemmEpi e,
                              hipStream_t s) {
  constexpr int BN = NT * 32;
  g_general_launches.fetch_add(1, std::memory_order_relaxed);
  dim3 grid((M + G_BM - 1) / G_BM, (N + BN - 1) / BN);
  size_t lds = (size_t)(G_BM + BN) * 128;
  if (lds < 4 * 32 * G_SP * sizeof(float)) lds = 4 * 32 * G_SP * sizeof(float);
  Traced tr(K_GEMM_NT + EPI, 2.0 * M * N * K, s, gemm_bytes<A_F32, EPI>(M, N, K, e));
  hipLaunchKernelGGL((gemm_nt_kernel<NT, A_F32, EPI>), grid, dim3(256), lds, s, A, lda, W, ldw, M, N, K, e);
  return hipGetLastError();
}

template <bool A_F32, int EPI>
static hipError_t launch_skinny(const void* A, long lda, const bf16* W, long ldw, int M, int N, int K, GemmEpi e,
                                hipStream_t s) {
  const long total = (long)((M + G_BM - 1) / G_BM) * ((N + 63) / 64);
  // 2 workgroups per CU are resident (LDS: two W panels each): one full round of them, each with an equal item range
  static const int force = getenv("SCM_SK_GRID") ? atoi(getenv("SCM_SK_GRID")) : 0;
  long grid = force > 0 ? force : 512;
  if (grid > total) grid = total;
  Traced tr(K_GEMM_SKINNY + EPI, 2.0 * M * N * K, s, gemm_bytes<A_F32, EPI>(M, N, K, e));
  const size_t lds = 2 * SK_WBYTES + (size_t)((N + 63) / 64) * 64 * sizeof(float);
  static std::atomic<unsigned long long> ready;
  if (hipError_t e = ensure_dynamic_lds((const void*)gemm_skinny_kernel<A_F32, EPI>, 96 * 1024, ready); e != hipSuccess)
    return e;
  hipLaunchKernelGGL((gemm_skinny_kernel<A_F32, EPI>), dim3((unsigned)grid), dim3(256), lds, s, A, lda, W, ldw,
                     M, N, K, e);
  return hipGetLastError();
}

static long nt_dma_grid(long M) {
  // whole rounds of 256 workgroups with at most 224 rows each; small problems: one workgroup per 32 rows
  long grid = 256 * ((M + 256l * GD_ROWS - 1) / (256l * GD_ROWS));
  if (grid > (M + 31) / 32) grid = (M + 31) / 32;
  return grid;
}

template <bool A_F32, int EPI>
static hipError_t dispatch_nt(const void* A, long lda, const bf16* W, long ldw, int M, int N, int K, GemmEpi e,
                              hipStream_t s) {
  // K <= 208 (the model dim) with a wide output: A-in-registers kernel.
  // Otherwise N = 200 -> one 224-wide tile; N in {640, 800, 1920} -> 160-wide tiles; else 128-wide.
  const long lim = 1l << 31;   // the A-in-registers kernel addresses its operands with 32-bit byte offsets
  const bool fits = (long)M * lda * 4 < lim && (long)N * ldw * 2 < lim && (long)M * e.ldo0 * 2 < lim &&
                    (long)M * e.ldo1 * 2 < lim && (long)M * e.ldp * 2 < lim;
  if (fits && K > 16 * (SK_KS - 1) && K <= 16 * SK_KS && N >= 256 &&
      (EPI == EPI_BF16 || EPI == EPI_BF16_BIAS || EPI == EPI_GELU_DUAL || EPI == EPI_DGELU))
    return launch_skinny<A_F32, EPI>(A, lda, W, ldw, M, N, K, e, s);
  if (!A_F32 && N <= 224 && K >= 128 && (long)M * lda * 2 < lim && (long)N * ldw * 2 < lim &&
      (EPI != EPI_F32_BIAS_RES || !e.resid || (long)M * e.ldr * 4 < lim)) {
    const size_t lds = (size_t)GD_LDS;
    static std::atomic<unsigned long long> ready;
    if (hipError_t e = ensure_dynamic_lds((const void*)gemm_nt_dma_kernel<EPI>, lds, ready); e != hipSuccess) return e;
    const long grid = nt_dma_grid(M);
    Traced tr(K_GEMM_NT + EPI, 2.0 * M * N * K, s, gemm_bytes<A_F32, EPI>(M, N, K, e));
    hipLaunchKernelGGL((gemm_nt_dma_kernel<EPI>), dim3((unsigned)grid), dim3(448), lds, s, (const bf16*)A, lda, W,
                       ldw, M, N, K, e);
    return hipGetLastError();
  }
  if (N <= 224) return launch_gemm<7, A_F32, EPI>(A, lda, W, ldw, M, N, K, e, s);
  if (N % 160 == 0) return launch_gemm<5, A_F32, EPI>(A, lda, W, ldw, M, N, K, e, s);
  return launch_gemm<4, A_F32, EPI>(A, lda, W, ldw, M, N, K, e, s);
}

static hipError_t gemm_nt_one(int epi, int a_f32, const void* A, long lda, const void* W, long ldw, int M, int N, int K,
                              GemmEpi e, hipStream_t s) {
  const bf16* Wb = (const bf16*)W;
#define SCM_CASE(E)                                                               \
  case E:                                                                         \
    return a_f32 ? dispatch_nt<true, E>(A, lda, Wb, ldw, M, N, K, e, s)           \
                 : dispatch_nt<false, E>(A, lda, Wb, ldw, M, N, K, e, s);
  switch (epi) {
    SCM_CASE(EPI_BF16)
    SCM_CASE(EPI_F32_BIAS_RES)
    SCM_CASE(EPI_GELU_DUAL)
    SCM_CASE(EPI_DGELU)
    SCM_CASE(EPI_F32)
    SCM_CASE(EPI_BF16_BIAS)
  }
#undef SCM_CASE
  return hipErrorInvalidValue;
}

// The fast kernels address their operands with 32-bit byte offsets inside buffer resources, i.e. below 2 GiB.  Every
// product here is independent row by row, so a taller problem runs as several launches over row slabs, each with
// all of its operands below 2 GiB (thousands of row tiles per slab: the launches stay many rounds deep).
static long slab_rows_for(long M, long row_bytes) {
  const long lim = (1l << 31) - 4096;
  if (M * row_bytes <= lim) return M;
  const long n = (M * row_bytes + lim - 1) / lim;
  long rows = ((M + n - 1) / n + 255) / 256 * 256;
  while (rows * row_bytes > lim) rows -= 256;
  return rows < 256 ? 256 : rows;
}

hipError_t gemm_nt(int epi, int a_f32, const void* A, long lda, const void* W, long ldw, int M, int N, int K,
                   GemmEpi e, hipStream_t s) {
  const long o0 = (epi == EPI_F32 || epi == EPI_F32_BIAS_RES) ? 4 : 2;   // bytes per element of out0
  long row_bytes = lda * (a_f32 ? 4 : 2);
  if (e.ldo0 * o0 > row_bytes) row_bytes = e.ldo0 * o0;
  if (e.out1 && e.ldo1 * 2 > row_bytes) row_bytes = e.ldo1 * 2;
  if (e.pre && e.ldp * 2 > row_bytes) row_bytes = e.ldp * 2;
  if (e.resid && e.ldr * 4 > row_bytes) row_bytes = e.ldr * 4;
  const long rows = slab_rows_for(M, row_bytes);
  for (long r0 = 0; r0 < M; r0 += rows) {
    const int m = (int)(M - r0 < rows ? M - r0 : rows);
    GemmEpi g = e;
    g.out0 = (char*)e.out0 + r0 * e.ldo0 * o0;
    if (e.out1) g.out1 = (char*)e.out1 + r0 * e.ldo1 * 2;
    if (e.resid) g.resid = e.resid + r0 * e.ldr;
    if (e.pre) g.pre = e.pre + r0 * e.ldp;
    const hipError_t err = gemm_nt_one(epi, a_f32, (const char*)A + r0 * lda * (a_f32 ? 4 : 2), lda, W, ldw, m, N, K, g, s);
    if (err != hipSuccess) return err;
  }
  return hipSuccess;
}

// Row slabs of the two products with LayerNorm epilogues: a fixed slab height, so that the number of workgroups
// (one column-partial record each) depends on M alone.
constexpr long LN_SLAB_ROWS = 8l * 256 * GD_ROWS;   // 458,752 rows: 8 rounds of 256 workgroups
constexpr long LN_SLAB_MAX_LD = ((1l << 31) - 4096) / (2 * LN_SLAB_ROWS);   // elements per A row that keep a slab below 2 GiB

bool gemm_ln_bwd_supported(long M, int N, int K, long lda, long ldw) {
  const long lim = 1l << 31;
  return N <= 224 && N % 4 == 0 && K >= 128 && K % 8 == 0 && M > 0 && lda <= LN_SLAB_MAX_LD && (long)N * ldw * 2 < lim;
}
int gemm_ln_bwd_blocks(long M) {
  return (int)((M / LN_SLAB_ROWS) * nt_dma_grid(LN_SLAB_ROWS) + (M % LN_SLAB_ROWS ? nt_dma_grid(M % LN_SLAB_ROWS) : 0));
}

hipError_t gemm_ln_bwd(const void* A, long lda, const void* W, long ldw, int M, int N, int K, GemmEpi e, hipStream_t s) {
  if (!gemm_ln_bwd_supported(M, N, K, lda, ldw)) return hipErrorInvalidValue;
  static_assert(LNB_LDS <= GD_LDS, "the epilogue's LDS lies inside the stage rings");
  static std::atomic<unsigned long long> ready;
  if (hipError_t err = ensure_dynamic_lds((const void*)gemm_nt_dma_kernel<EPI_LN_BWD>, (size_t)GD_LDS, ready); err != hipSuccess)
    return err;
  for (long r0 = 0; r0 < M; r0 += LN_SLAB_ROWS) {
    const int m = (int)(M - r0 < LN_SLAB_ROWS ? M - r0 : LN_SLAB_ROWS);
    GemmEpi g = e;
    g.out0 = (float*)e.out0 + r0 * N;
    if (e.resid) g.resid = e.resid + r0 * N;
    g.ln_x = e.ln_x + r0 * N; g.ln_mean = e.ln_mean + r0; g.ln_rstd = e.ln_rstd + r0;
    if (e.ln_dxb) g.ln_dxb = e.ln_dxb + r0 * N;
    g.ln_partial = e.ln_partial + (long)gemm_ln_bwd_blocks(r0) * 3 * N;
    const double bytes = (double)m * K * 2 + (double)N * K * 2 + (double)m * N * (4 + (e.resid ? 4 : 0) + 4 + (e.ln_dxb ? 2 : 0));
    Traced tr(K_GEMM_LN_BWD, 2.0 * m * N * K, s, bytes);
    hipLaunchKernelGGL((gemm_nt_dma_kernel<EPI_LN_BWD>), dim3((unsigned)nt_dma_grid(m)), dim3(448), (size_t)GD_LDS, s,
                       (const bf16*)A + r0 * lda, lda, (const bf16*)W, ldw, m, N, K, g);
    if (hipError_t err = hipGetLastError(); err != hipSuccess) return err;
  }
  return hipSuccess;
}

bool gemm_res_ln_supported(long M, int N, int K, long lda, long ldw) {
  return N <= 220 && gemm_ln_bwd_supported(M, N, K, lda, ldw);
}

hipError_t gemm_res_ln(const void* A, long lda, const void* W, long ldw, int M, int N, int K, GemmEpi e, hipStream_t s) {
  if (!gemm_res_ln_supported(M, N, K, lda, ldw) || (e.resid && e.ldr != N)) return hipErrorInvalidValue;
  static std::atomic<unsigned long long> ready;
  if (hipError_t err = ensure_dynamic_lds((const void*)gemm_nt_dma_kernel<EPI_RES_LN>, (size_t)GD_LDS, ready); err != hipSuccess)
    return err;
  for (long r0 = 0; r0 < M; r0 += LN_SLAB_ROWS) {
    const int m = (int)(M - r0 < LN_SLAB_ROWS ? M - r0 : LN_SLAB_ROWS);
    GemmEpi g = e;
    g.out0 = (float*)e.out0 + r0 * N;
    if (e.resid) g.resid = e.resid + r0 * N;
    g.ln_dxb = e.ln_dxb + r0 * N; g.ln_mean_out = e.ln_mean_out + r0; g.ln_rstd_out = e.ln_rstd_out + r0;
    const double bytes = (double)m * K * 2 + (double)N * K * 2 + (double)m * N * (4 + (e.resid ? 4 : 0) + 2);
    Traced tr(K_GEMM_RES_LN, 2.0 * m * N * K, s, bytes);
    hipLaunchKernelGGL((gemm_nt_dma_kernel<EPI_RES_LN>), dim3((unsigned)nt_dma_grid(m)), dim3(448), (size_t)GD_LDS, s,
                       (const bf16*)A + r0 * lda, lda, (const bf16*)W, ldw, m, N, K, g);
    if (hipError_t err = hipGetLastError(); err != hipSuccess) return err;
  }
  return hipSuccess;
}

static hipError_t launch_wgrad_dma(const WgradPair& pr, int T, int slices, hipStream_t s) {
  const int ntiles = (T + W_BT - 1) / W_BT;
  const int tps = (ntiles + slices - 1) / slices;
  bool any_ones = false;
  for (int i = 0; i < pr.n; ++i) any_ones = any_ones || pr.p[i].ones;
  int ptiles = 0;
  double flops = 0, bytes = 0;
  for (int i = 0; i < pr.n; ++i) {
    ptiles += (pr.p[i].P + WD_BP - 1) / WD_BP;
    flops += 2.0 * T * pr.p[i].P * pr.p[i].Q;
    bytes += (double)T * (pr.p[i].P * 2.0 + pr.p[i].Q * 2.0) + (double)pr.p[i].P * pr.p[i].Q * 4;
  }
  Traced tr(any_ones ? K_WGRAD_BIAS : K_WGRAD, flops, s, bytes);
  const size_t lds = (size_t)WD_NSTAGE * WD_STAGE;
  static std::atomic<unsigned long long> ready[2];
  const void* kfn = any_ones ? (const void*)wgrad_dma_kernel<true> : (const void*)wgrad_dma_kernel<false>;
  if (hipError_t e = ensure_dynamic_lds(kfn, lds, ready[any_ones ? 1 : 0]); e != hipSuccess) return e;
  dim3 grid(ptiles * slices);
  if (any_ones)
    hipLaunchKernelGGL((wgrad_dma_kernel<true>), grid, dim3(256), lds, s, pr, T, tps);
  else
    hipLaunchKernelGGL((wgrad_dma_kernel<false>), grid, dim3(256), lds, s, pr, T, tps);
  return hipGetLastError();
}

hipError_t wgrad_multi(const WgradDesc* d, int n, int T, int slices, hipStream_t s) {
  if (n < 1 || n > WD_MAXN) return hipErrorInvalidValue;
  const long lim = 0x7FFFF000l;   // per token slice (the kernel re-bases its resources at the slice's first token)
  const long slice_tokens = (long)(((T + W_BT - 1) / W_BT + slices - 1) / slices) * W_BT;
  WgradPair pr{};
  for (int i = 0; i < n; ++i) {
    if (slice_tokens * d[i].ldp * 2 >= lim || slice_tokens * d[i].ldq * 2 >= lim) return hipErrorInvalidValue;
    pr.p[i] = WgradOne{(const bf16*)d[i].Pm, d[i].ldp, (const bf16*)d[i].Qm, d[i].ldq, d[i].slab, d[i].P, d[i].Q, d[i].ones};
  }
  pr.n = n;
  return launch_wgrad_dma(pr, T, slices, s);
}

hipError_t wgrad(const void* Pm, long ldp, const void* Qm, long ldq, int q_f32, int ones, int T, int P, int Q,
                 int slices, float* slab, hipStream_t s) {
  const int ntiles = (T + W_BT - 1) / W_BT;
  const int tps = (ntiles + slices - 1) / slices;
  const bf16* Pb = (const bf16*)Pm;
  const long lim = 0x7FFFF000l;   // per token slice (the kernel re-bases its resources at the slice's first token)
  if (!q_f32 && (long)tps * W_BT * ldp * 2 < lim && (long)tps * W_BT * ldq * 2 < lim) {
    // bf16 operands: asynchronous global->LDS pipeline
    WgradPair pr{};
    pr.p[0] = WgradOne{Pb, ldp, (const bf16*)Qm, ldq, slab, P, Q, ones};
    pr.n = 1;
    return launch_wgrad_dma(pr, T, slices, s);
  }
  Traced tr(ones ? K_WGRAD_BIAS : K_WGRAD, 2.0 * T * P * Q, s, (double)T * (P * 2.0 + Q * (q_f32 ? 4.0 : 2.0)) + (double)P * Q * 4);
  dim3 grid(((P + W_BP - 1) / W_BP) * slices);
  size_t lds = (size_t)W_BT * (W_PP + W_QP);
  if (q_f32 && ones)
    hipLaunchKernelGGL((wgrad_kernel<true, true>), grid, dim3(320), lds, s, Pb, ldp, Qm, ldq, T, P, Q, tps, slab);
  else if (q_f32)
    hipLaunchKernelGGL((wgrad_kernel<true, false>), grid, dim3(320), lds, s, Pb, ldp, Qm, ldq, T, P, Q, tps, slab);
  else if (ones)
    hipLaunchKernelGGL((wgrad_kernel<false, true>), grid, dim3(320), lds, s, Pb, ldp, Qm, ldq, T, P, Q, tps, slab);
  else
    hipLaunchKernelGGL((wgrad_kernel<false, false>), grid, dim3(320), lds, s, Pb, ldp, Qm, ldq, T, P, Q, tps, slab);
  return hipGetLastError();
}

hipError_t wgrad_reduce(const float* slab, int S, int P, int Q, float* grad, float* bias, int trans, int accumulate,
                        hipStream_t s) {
  const int n = P * (bias ? Q + 1 : Q);
  Traced tr(K_WGRAD_REDUCE, 0, s);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, s, slab, S, P, Q, grad, bias, trans,
                     accumulate);
  return hipGetLastError();
}

hipError_t cast_weights_batch(const CastItem* items, int n, hipStream_t s) {
  Traced tr(K_CAST, 0, s);
  hipLaunchKernelGGL(cast_weights_batch_kernel, dim3(64, n), dim3(256), 0, s, items);
  return hipGetLastError();
}

hipError_t cast_weight(const float* src, int R, int C, void* dst, void* dstT, hipStream_t s) {
  const int n = R * C;
  Traced tr(K_CAST, 0, s);
  hipLaunchKernelGGL(cast_weight_kernel, dim3((n + 255) / 256), dim3(256), 0, s, src, R, C, (bf16*)dst, (bf16*)dstT);
  return hipGetLastError();
}

}  // namespace scm
