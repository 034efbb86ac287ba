// The optimizer step of the S1 loop (s1_pretrain_rna.py:303-305: clip_grad_norm_(parameters, 100), Adam.step(),
// zero_grad()) over ALL parameter tensors in three launches driven by one device table, instead of the hundred small
// launches of the multi-tensor library calls.  The arithmetic is torch.optim.Adam's (betas, eps, no weight decay, no
// amsgrad) operation by operation:
//   g       = grad * grad_scale                      (grad_scale = 1 / world size: the average over the ranks)
//   norm    = sqrt(sum over tensors of (sqrt(sum g^2))^2),  coef = min(1, max_norm / (norm + 1e-6)),  g *= coef
//   m       = m + (g - m) * (1 - beta1)              (Tensor.lerp_)
//   v       = v * beta2 + (1 - beta2) * g * g
//   p       = p - (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
//   grad    = 0
// The table lists chunks of at most ADAM_CHUNK elements, each inside one tensor; `tensor` numbers the tensor a chunk
// belongs to (chunks of one tensor are consecutive), so the per-tensor norms can be formed as torch forms them.
#include "scm_launch.hpp"

namespace scm {

constexpr int ADAM_CHUNK = 4096;

// partial[c] = sum of (grad * grad_scale)^2 over chunk c
__global__ __launch_bounds__(256) void adam_sqnorm_kernel(const AdamItem* __restrict__ items, float grad_scale,
                                                          float* __restrict__ partial) {
  __shared__ float ws[4];
  const AdamItem it = items[blockIdx.x];
  float s = 0.f;
  for (int i = threadIdx.x; i < it.n; i += 256) { const float g = it.g[i] * grad_scale; s += g * g; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

// One workgroup forms the total norm from the chunk partials in a fixed order (chunk sums per tensor, square root per
// tensor, squares of those summed, square root) and leaves the clip factor in scal[0] (the norm in scal[1]).
constexpr int ADAM_LONG = 32;       // a tensor of more chunks than this is summed by the whole workgroup
constexpr int ADAM_MAX_LONG = 32;   // such tensors the workgroup takes on; further ones go the single-thread way
__global__ __launch_bounds__(256) void adam_norm_kernel(const AdamItem* __restrict__ items, int nchunks,
                                                        const float* __restrict__ partial, float max_norm,
                                                        float* __restrict__ scal, float* __restrict__ norm_out) {
  __shared__ float ws[4];
  __shared__ int n_long;
  __shared__ int long_lo[ADAM_MAX_LONG], long_hi[ADAM_MAX_LONG];
  if (threadIdx.x == 0) n_long = 0;
  __syncthreads();
  // thread t walks the chunks t, t + 256, ...; the thread that meets a tensor's FIRST chunk finds its last one (the
  // chunks of a tensor are consecutive) and sums a short tensor itself
  float acc = 0.f;
  for (int c = threadIdx.x; c < nchunks; c += 256) {
    const int tn_ = items[c].tensor;
    if (c > 0 && items[c - 1].tensor == tn_) continue;   // not the first chunk of its tensor
    int lo = c, hi = nchunks;                            // first chunk j > c with another tensor number
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (items[mid].tensor == tn_) lo = mid; else hi = mid;
    }
    if (hi - c > ADAM_LONG) {
      const int slot = atomicAdd(&n_long, 1);
      if (slot < ADAM_MAX_LONG) { long_lo[slot] = c; long_hi[slot] = hi; continue; }
    }
    float ts = 0.f;
    for (int j = c; j < hi; ++j) ts += partial[j];
    const float tn = sqrtf(ts);
    acc += tn * tn;
  }
  __syncthreads();
  const int nl = min(n_long, ADAM_MAX_LONG);
  if (threadIdx.x == 0)                                  // the long tensors in table order, whoever found them
    for (int i = 1; i < nl; ++i)
      for (int j = i; j > 0 && long_lo[j - 1] > long_lo[j]; --j) {
        const int a = long_lo[j], b = long_hi[j];
        long_lo[j] = long_lo[j - 1]; long_hi[j] = long_hi[j - 1];
        long_lo[j - 1] = a; long_hi[j - 1] = b;
      }
  __syncthreads();
  for (int i = 0; i < nl; ++i) {
    float ts = 0.f;
    for (int j = long_lo[i] + threadIdx.x; j < long_hi[i]; j += 256) ts += partial[j];
    ts = wave_sum(ts);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = ts;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float tn = sqrtf((ws[0] + ws[1]) + (ws[2] + ws[3]));
      acc += tn * tn;
    }
  }
  acc = wave_sum(acc);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float norm = sqrtf((ws[0] + ws[1]) + (ws[2] + ws[3]));
    scal[0] = max_norm > 0.f ? fminf(1.f, max_norm / (norm + 1e-6f)) : 1.f;
    scal[1] = norm;
    if (norm_out) *norm_out = norm;
  }
}

__global__ __launch_bounds__(256) void adam_update_kernel(const AdamItem* __restrict__ items,
                                                          const float* __restrict__ scal, AdamScalars k) {
  const float gs = k.grad_scale * scal[0];
  const AdamItem it = items[blockIdx.x];
  for (int i = threadIdx.x; i < it.n; i += 256) {
    const float g = it.g[i] * gs;
    float m = it.m[i], v = it.v[i];
    m = m + (g - m) * k.one_minus_beta1;
    v = v * k.beta2 + k.one_minus_beta2 * g * g;
    const float denom = sqrtf(v) / k.bc2_sqrt + k.eps;
    it.p[i] = it.p[i] - k.step_size * (m / denom);
    it.m[i] = m;
    it.v[i] = v;
    it.g[i] = 0.f;
  }
}

// partial: nchunks + 2 floats (chunk partials, then clip factor and norm)
hipError_t adam_step(const AdamItem* items, int nchunks, float* partial, AdamScalars k, float* norm_out, hipStream_t s) {
  Traced tr(K_ADAM, 0, s);
  hipLaunchKernelGGL(adam_sqnorm_kernel, dim3(nchunks), dim3(256), 0, s, items, k.grad_scale, partial);
  hipLaunchKernelGGL(adam_norm_kernel, dim3(1), dim3(256), 0, s, items, nchunks, partial, k.max_norm, partial + nchunks,
                     norm_out);
  hipLaunchKernelGGL(adam_update_kernel, dim3(nchunks), dim3(256), 0, s, items, partial + nchunks, k);
  return hipGetLastError();
}

int adam_chunk_elems() { return ADAM_CHUNK; }

}  // namespace scm
