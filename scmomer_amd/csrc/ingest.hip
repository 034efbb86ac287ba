// The two host-side chores of the reference that stall a millisecond-scale training step when left on the CPU:
//
//  bin_csr / bin_dense : SCDataset.__getitem__ (s1_pretrain_rna.py:133-137: sparse row -> dense, values above
//              class_limit-2 set to it, truncation to int64) and the scMomer rule (scmomer.py:56,108:
//              round-half-to-even, clamp to [0,5]).  The expression matrix stays resident in HBM as CSR; a batch
//              of cells is gathered, densified and binned by one kernel.  Integer work: bit-exact.
//  orthogonal_omega    : gaussian_orthogonal_random_matrix / orthogonal_matrix_chunk
//              (performer_pytorch.py:110-139): Householder QR of 64x64 Gaussian blocks (LAPACK's sign and
//              scaling conventions, so Q equals the reference's up to rounding), rows of Q^T scaled by the norms
//              of a second set of Gaussian rows.
#include "scm_launch.hpp"

namespace scm {

// ------------------------------------------------------------------------------------------
// binning
// ------------------------------------------------------------------------------------------
SCM_DEV long long bin_value(float v, int rule, int limit) {
  if (rule == 0) {                       // S1: x[x > limit] = limit; .long() truncates toward zero
    // a NaN is no valid expression value: it becomes the most negative integer (what the reference's host-side
    // .long() yields), and a negative value truncates to a negative token; both are then reported by the embedding
    // kernel's token check instead of being looked up
    if (v != v) return (long long)0x8000000000000000ull;
    v = v > (float)limit ? (float)limit : v;
    return (long long)v;
  }
  float r = rintf(v);                    // scMomer: round half to even, then clamp to [0, limit]
  r = fminf(fmaxf(r, 0.f), (float)limit);
  return (long long)r;
}

// One 1024-thread workgroup per cell.  The dense f32 row lives in LDS (N <= BIN_MAX_GENES), entries are added into
// it (duplicate column indices sum, as scipy's toarray() does), then the row is binned and written as int64.
__global__ __launch_bounds__(1024) void bin_csr_kernel(const int64_t* __restrict__ indptr,
                                                       const int32_t* __restrict__ indices,
                                                       const float* __restrict__ values,
                                                       const int64_t* __restrict__ rows, int n_rows, int N, int rule,
                                                       int limit, int64_t* __restrict__ tokens, int* __restrict__ err) {
  extern __shared__ float dense[];
  const int tid = threadIdx.x;
  for (int i = tid; i < N; i += 1024) dense[i] = 0.f;
  __syncthreads();
  const long long r = rows ? rows[blockIdx.x] : (long long)blockIdx.x;
  if (r >= 0 && r < n_rows) {
    const long long lo = indptr[r], hi = indptr[r + 1];
    for (long long e = lo + tid; e < hi; e += 1024) {
      const int c = indices[e];
      if ((unsigned)c < (unsigned)N) atomicAdd(&dense[c], values[e]);
      else atomicOr(err, 2);
    }
  } else if (tid == 0) {
    atomicOr(err, 1);
  }
  __syncthreads();
  int64_t* out = tokens + (long)blockIdx.x * N;
  for (int i = tid; i < N; i += 1024) out[i] = bin_value(dense[i], rule, limit);
}

__global__ __launch_bounds__(256) void bin_dense_kernel(const float* __restrict__ x, long total, int rule, int limit,
                                                        int64_t* __restrict__ tokens) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256)
    tokens[i] = bin_value(x[i], rule, limit);
}

// ------------------------------------------------------------------------------------------
// projection redraw: one wave per 64x64 block.  grid = (blocks per matrix, matrices)
// ------------------------------------------------------------------------------------------
constexpr int QP = DH + 1;   // LDS pitch (floats): column walks are conflict-free

__global__ __launch_bounds__(64) void orthogonal_omega_kernel(const float* __restrict__ gauss_blocks,
                                                              const float* __restrict__ gauss_mult, int nb_rows,
                                                              int scaling, float* __restrict__ omega) {
  __shared__ float A[DH * QP];   // column k below the diagonal ends up holding the Householder vector v_k
  __shared__ float Q[DH * QP];
  __shared__ float tau[DH];
  const int l = threadIdx.x;
  const int blk = blockIdx.x, nblk = gridDim.x, mat = blockIdx.y;
  const float* g = gauss_blocks + ((long)mat * nblk + blk) * DH * DH;
  for (int r = 0; r < DH; ++r) A[r * QP + l] = g[r * DH + l];
  if (l == DH - 1) tau[l] = 0.f;   // the last column of a square matrix has nothing below it: H = I
  __syncthreads();

  // A = Q R, Q = H_0 H_1 ... H_62, H_k = I - tau_k v_k v_k^T with v_k[k] = 1
  for (int k = 0; k < DH - 1; ++k) {
    const float x = (l > k) ? A[l * QP + k] : 0.f;       // lane = row
    const float xn2 = wave_sum(x * x);
    const float alpha = A[k * QP + k];
    float t = 0.f;
    if (xn2 != 0.f) {                                    // uniform across the wave
      const float beta = -copysignf(sqrtf(alpha * alpha + xn2), alpha);
      t = (beta - alpha) / beta;
      const float scale = 1.f / (alpha - beta);
      if (l > k) A[l * QP + k] = x * scale;
    }
    if (l == 0) tau[k] = t;
    __syncthreads();
    if (t != 0.f && l > k) {                             // lane = trailing column: A[:,l] -= tau v (v . A[:,l])
      float w = A[k * QP + l];
      for (int r = k + 1; r < DH; ++r) w += A[r * QP + k] * A[r * QP + l];
      w *= t;
      A[k * QP + l] -= w;
      for (int r = k + 1; r < DH; ++r) A[r * QP + l] -= A[r * QP + k] * w;
    }
    __syncthreads();
  }

  // Q = H_0 ... H_62 I, applied right to left; lane = column of Q
  for (int r = 0; r < DH; ++r) Q[r * QP + l] = (r == l) ? 1.f : 0.f;
  for (int k = DH - 2; k >= 0; --k) {
    const float t = tau[k];
    if (t == 0.f) continue;
    float w = Q[k * QP + l];
    for (int r = k + 1; r < DH; ++r) w += A[r * QP + k] * Q[r * QP + l];
    w *= t;
    Q[k * QP + l] -= w;
    for (int r = k + 1; r < DH; ++r) Q[r * QP + l] -= A[r * QP + k] * w;
  }
  __syncthreads();

  // rows of Q^T, scaled: omega[row][c] = m_row * Q[c][j]
  for (int j = 0; j < DH; ++j) {
    const int row = blk * DH + j;
    if (row >= nb_rows) break;
    float m;
    if (scaling == 0) {
      const float gm = gauss_mult[((long)mat * nb_rows + row) * DH + l];
      m = sqrtf(wave_sum(gm * gm));
    } else {
      m = sqrtf((float)DH);
    }
    omega[((long)mat * nb_rows + row) * DH + l] = m * Q[l * QP + j];
  }
}

// ---------------------------------------------------------------- host launchers
hipError_t bin_csr(const int64_t* indptr, const int32_t* indices, const float* values, const int64_t* rows, int n_rows,
                   int B, int N, int rule, int limit, int64_t* tokens, int* err, hipStream_t s) {
  static std::atomic<unsigned long long> ready;
  if (hipError_t e = ensure_dynamic_lds((const void*)bin_csr_kernel, BIN_MAX_GENES * sizeof(float), ready); e != hipSuccess)
    return e;
  Traced tr(K_BIN, 0, s, (double)B * N * 8);
  hipLaunchKernelGGL(bin_csr_kernel, dim3(B), dim3(1024), (size_t)N * sizeof(float), s, indptr, indices, values, rows,
                     n_rows, N, rule, limit, tokens, err);
  return hipGetLastError();
}

hipError_t bin_dense(const float* x, long total, int rule, int limit, int64_t* tokens, hipStream_t s) {
  Traced tr(K_BIN, 0, s, (double)total * 12);
  long blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(bin_dense_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, total, rule, limit, tokens);
  return hipGetLastError();
}

hipError_t orthogonal_omega(const float* gauss_blocks, const float* gauss_mult, int nmat, int nb_rows, int scaling,
                            float* omega, hipStream_t s) {
  Traced tr(K_OMEGA_QR, 0, s);
  const int nblk = (nb_rows + DH - 1) / DH;
  hipLaunchKernelGGL(orthogonal_omega_kernel, dim3(nblk, nmat), dim3(64), 0, s, gauss_blocks, gauss_mult, nb_rows,
                     scaling, omega);
  return hipGetLastError();
}

}  // namespace scm
