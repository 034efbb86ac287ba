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

namespace scm {

// ------------------------------------------------------------------------------------------
// gemm_nt
// ------------------------------------------------------------------------------------------
enum { EPI_BF16 = 0, EPI_F32_BIAS_RES = 1, EPI_GELU_DUAL = 2, EPI_DGELU = 3, EPI_F32 = 4, EPI_BF16_BIAS = 5 };

constexpr int G_BM = 128;
constexpr int G_BK = 64;
constexpr int G_SP = 68;   // floats per staged epilogue row (64 + 4: rows 16-byte aligned, conflict-free)

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
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, half = l >> 5;
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

  load_stage(0);
  for (int kt = 0; kt < nk; ++kt) {
    write_stage();
    __syncthreads();
    if (kt + 1 < nk) load_stage(kt + 1);
    const int krem = K - kt * G_BK;
    const int ks = krem >= G_BK ? 4 : (krem + 15) >> 4;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s < ks) {
        bf16x8 a = img_row_frag(As, 32 * w, 16 * s);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          bf16x8 b = img_row_frag(Ws, 32 * t, 16 * s);
          acc[t] = mfma32(a, b, acc[t]);
        }
      }
    }
    __syncthreads();
  }

  // Epilogue through LDS: accumulator tiles (column on the lane, rows in registers) are staged per wave as
  // [32 rows][64 columns] f32, then read back row-wise so that every lane handles 8 consecutive columns of
  // one row: 16-byte (bf16) / 32-byte (f32) global stores, 128..256 contiguous bytes per row, and the
  // bias / residual / pre-activation operands are read with the same coalesced shape.
  float* stage = (float*)smem + w * (32 * G_SP);
#pragma unroll
  for (int tp = 0; tp < (NT + 1) / 2; ++tp) {
    const int t0 = 2 * tp, t1 = 2 * tp + 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      stage[acc_row(r, half) * G_SP + (l & 31)] = acc[t0][r];
      if (t1 < NT) stage[acc_row(r, half) * G_SP + 32 + (l & 31)] = acc[t1][r];
    }
    __builtin_amdgcn_wave_barrier();
    const int ch = l & 7;
    const int col = n0 + 64 * tp + 8 * ch;
    const bool col_ok = (t1 < NT || ch < 4) && col < N;
    float bias[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bias[j] = 0.f;
    if ((EPI == EPI_F32_BIAS_RES || EPI == EPI_GELU_DUAL || EPI == EPI_BF16_BIAS) && e.bias && col_ok) {
      const f32x4 b0 = *(const f32x4*)(e.bias + col), b1 = *(const f32x4*)(e.bias + col + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { bias[j] = b0[j]; bias[4 + j] = b1[j]; }
    }
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int rl = 8 * pass + (l >> 3);
      const long row = m0 + 32 * w + rl;
      if (!col_ok || row >= M) continue;
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
        if (EPI == EPI_F32_BIAS_RES && e.resid) {
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

}  // namespace scm

// ------------------------------------------------------------------------------------------
// host launchers (C linkage wrappers live in capi.hip)
// ------------------------------------------------------------------------------------------
namespace scm {

template <int NT, bool A_F32, int EPI>
static hipError_t launch_gemm(const void* A, long lda, const bf16* W, long ldw, int M, int N, int K, GemmEpi e,
                              hipStream_t s) {
  constexpr int BN = NT * 32;
  dim3 grid((M + G_BM - 1) / G_BM, (N + BN - 1) / BN);
  size_t lds = (size_t)(G_BM + BN) * 128;
  if (lds < 4 * 32 * G_SP * sizeof(float)) lds = 4 * 32 * G_SP * sizeof(float);
  Traced tr(K_GEMM_NT, 2.0 * M * N * K, s);
  hipLaunchKernelGGL((gemm_nt_kernel<NT, A_F32, EPI>), grid, dim3(256), lds, s, A, lda, W, ldw, M, N, K, e);
  return hipGetLastError();
}

template <bool A_F32, int EPI>
static hipError_t dispatch_nt(const void* A, long lda, const bf16* W, long ldw, int M, int N, int K, GemmEpi e,
                              hipStream_t s) {
  // N = 200 -> one 224-wide tile; N in {640, 800, 1920} -> 160-wide tiles; otherwise 128-wide.
  if (N <= 224) return launch_gemm<7, A_F32, EPI>(A, lda, W, ldw, M, N, K, e, s);
  if (N % 160 == 0) return launch_gemm<5, A_F32, EPI>(A, lda, W, ldw, M, N, K, e, s);
  return launch_gemm<4, A_F32, EPI>(A, lda, W, ldw, M, N, K, e, s);
}

hipError_t gemm_nt(int epi, int a_f32, const void* A, long lda, const void* W, long ldw, int M, int N, int K,
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

hipError_t wgrad(const void* Pm, long ldp, const void* Qm, long ldq, int q_f32, int ones, int T, int P, int Q,
                 int slices, float* slab, hipStream_t s) {
  const int ntiles = (T + W_BT - 1) / W_BT;
  const int tps = (ntiles + slices - 1) / slices;
  dim3 grid(((P + W_BP - 1) / W_BP) * slices);
  size_t lds = (size_t)W_BT * (W_PP + W_QP);
  const bf16* Pb = (const bf16*)Pm;
  Traced tr(K_WGRAD, 2.0 * T * P * Q, s);
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

hipError_t cast_weight(const float* src, int R, int C, void* dst, void* dstT, hipStream_t s) {
  const int n = R * C;
  Traced tr(K_CAST, 0, s);
  hipLaunchKernelGGL(cast_weight_kernel, dim3((n + 255) / 256), dim3(256), 0, s, src, R, C, (bf16*)dst, (bf16*)dstT);
  return hipGetLastError();
}

}  // namespace scm
