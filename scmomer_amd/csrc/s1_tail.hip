// The two ends of the S1 masked-autoencoding step that sit around the encoder:
//
//  s1_mask   : data_mask / get_mask_subset_with_prob / mask_with_tokens / prob_mask_like
//              (s1_pretrain_rna.py:88-121): per cell, the ceil(mask_prob * #maskable) maskable positions
//              with the largest uniform draws are picked; tokens 0 and pad are never maskable;
//              picked positions whose second draw is below replace_prob become the mask token; labels hold
//              the original token at picked positions and pad elsewhere.  Index work: bit-exact.
//  ce_fwd/bwd: CrossEntropyLoss(ignore_index, reduction='mean') on the [T, C<=8] logits
//              (s1_pretrain_rna.py:268,297) and its gradient.
#include "scm_launch.hpp"

namespace scm {

// ------------------------------------------------------------------------------------------
// s1_mask: one 1024-thread workgroup per cell.  The k-th largest draw among the maskable positions is found by
// a 4 x 8-bit radix select on the float bits (draws are in [0,1): the unsigned order of the bits is the order
// of the values); draws above it are picked, and of the draws equal to it the lowest positions, as many as are
// still needed.
// ------------------------------------------------------------------------------------------
constexpr int S1_PER_THREAD = 20;   // positions per thread held in registers: N <= 20 * 1024

__global__ __launch_bounds__(1024) void s1_mask_kernel(const int64_t* __restrict__ tok,
                                                       const float* __restrict__ rand_pick,
                                                       const float* __restrict__ rand_replace, int N,
                                                       float mask_prob, float replace_prob, int max_masked,
                                                       long long pad_id, long long mask_id,
                                                       int64_t* __restrict__ masked, int64_t* __restrict__ labels) {
  __shared__ int hist[256];
  __shared__ int sh_count, sh_prefix, sh_need;
  __shared__ int tie_cnt[S1_PER_THREAD * 16];
  const int tid = threadIdx.x;
  const long base = (long)blockIdx.x * N;
  const int64_t* t = tok + base;
  const float* rp = rand_pick + base;
  const float* rr = rand_replace + base;

  // The cell is read ONCE: every thread keeps the keys of its positions (tid, tid + 1024, ...) in registers;
  // key 0xFFFFFFFF marks "not maskable" (draws are in [0,1): their bit patterns stay below 0x3F800000).
  unsigned keys[S1_PER_THREAD];
  int c = 0;
#pragma unroll
  for (int j = 0; j < S1_PER_THREAD; ++j) {
    const int i = tid + 1024 * j;
    unsigned k = 0xFFFFFFFFu;
    if (i < N) {
      const long long tv = t[i];
      if (tv != 0 && tv != pad_id) { k = __float_as_uint(rp[i]); c += 1; }
    }
    keys[j] = k;
  }

  // 1. number of maskable positions
  if (tid == 0) sh_count = 0;
  __syncthreads();
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o);
  if ((tid & 63) == 0) atomicAdd(&sh_count, c);
  __syncthreads();
  const int count = sh_count;
  // ceil(count * prob) in float32, as (num_tokens * prob).ceil() evaluates it; at most ceil(prob * N) are kept
  int k = (int)ceilf((float)count * mask_prob);
  k = min(k, max_masked);
  k = min(k, count);

  // 2. radix select: after the passes, `prefix` holds the bits of the k-th largest key
  unsigned prefix = 0;
  int need = k;   // how many are still wanted among the keys that match the prefix so far
  if (k > 0) {
    for (int pass = 3; pass >= 0; --pass) {
      const int shift = 8 * pass;
      const unsigned himask = (pass == 3) ? 0u : (0xFFFFFFFFu << (shift + 8));
      for (int i = tid; i < 256; i += 1024) hist[i] = 0;
      __syncthreads();
#pragma unroll
      for (int j = 0; j < S1_PER_THREAD; ++j) {
        const unsigned key = keys[j];
        if (key != 0xFFFFFFFFu && (key & himask) == (prefix & himask)) atomicAdd(&hist[(key >> shift) & 255], 1);
      }
      __syncthreads();
      if (tid < 256) {  // the digit d with  #(keys of a larger digit) < need <= #(keys of digit >= d)
        int above_d = 0;
        for (int j = tid + 1; j < 256; ++j) above_d += hist[j];
        const int mine = hist[tid];
        if (above_d < need && need <= above_d + mine) {
          sh_prefix = (int)(prefix | ((unsigned)tid << shift));
          sh_need = need - above_d;
        }
      }
      __syncthreads();
      prefix = (unsigned)sh_prefix;
      need = sh_need;
      __syncthreads();
    }
  }
  // now: keys > prefix are all picked; `need` of the keys == prefix are picked, lowest positions first

  // 3. write-out.  Ties are ranked in position order (position = tid + 1024 j): the number of ties per (chunk j,
  // wave) goes to LDS, one thread turns the table into exclusive prefix sums in (j, wave) order, and every thread
  // adds the ties of lower lanes of its own wave (ballot).
  const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
  for (int j = 0; j < S1_PER_THREAD; ++j) {
    const bool tie = keys[j] != 0xFFFFFFFFu && k > 0 && keys[j] == prefix;
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(tie);
    if (lane == 0) tie_cnt[j * 16 + wv] = __builtin_popcountll(bal);
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int e = 0; e < S1_PER_THREAD * 16; ++e) { const int v = tie_cnt[e]; tie_cnt[e] = run; run += v; }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < S1_PER_THREAD; ++j) {
    const int i = tid + 1024 * j;
    const unsigned key = keys[j];
    const bool maskable = key != 0xFFFFFFFFu && k > 0;
    const bool above = maskable && key > prefix;
    const bool tie = maskable && key == prefix;
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(tie);
    const int before = __builtin_popcountll(bal & ((1ull << lane) - 1ull));
    const bool picked = above || (tie && (tie_cnt[j * 16 + wv] + before) < need);
    if (i < N) {
      const long long tv = t[i];
      const bool rep = rr[i] < replace_prob;
      masked[base + i] = (picked && rep) ? mask_id : tv;
      labels[base + i] = picked ? tv : pad_id;
    }
  }
}

// ------------------------------------------------------------------------------------------
// cross entropy over C <= 8 classes, mean over the tokens whose label is not `ignore`
// ------------------------------------------------------------------------------------------
constexpr int CE_MAXC = 8;

// Tokens are split into loss groups of group_tokens consecutive tokens (the last one may be shorter): every group
// has its own mean, as when each group is one call of the loss (one micro-batch of s1_pretrain_rna.py:297).
// grid (blocks, groups): partial[group][block] = { sum of -log softmax(logits)[label], number of counted tokens }
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits,
                                                     const int64_t* __restrict__ labels, long T, int C,
                                                     long long ignore, long group_tokens, float* __restrict__ partial) {
  __shared__ float rs[4], rc[4];
  float s = 0.f, cnt = 0.f;
  const long t_lo = (long)blockIdx.y * group_tokens, t_hi = min(T, t_lo + group_tokens);
  for (long t = t_lo + (long)blockIdx.x * 256 + threadIdx.x; t < t_hi; t += (long)gridDim.x * 256) {
    const long long lab = labels[t];
    if (lab == ignore) continue;
    float v[CE_MAXC];
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < CE_MAXC; ++c) {
      v[c] = c < C ? logits[t * C + c] : -INFINITY;
      m = fmaxf(m, v[c]);
    }
    float se = 0.f, pick = 0.f;
#pragma unroll
    for (int c = 0; c < CE_MAXC; ++c) {
      se += c < C ? expf(v[c] - m) : 0.f;
      pick = (c == (int)lab) ? v[c] : pick;
    }
    s += (m + logf(se)) - pick;
    cnt += 1.f;
  }
  s = wave_sum(s);
  cnt = wave_sum(cnt);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { rs[w] = s; rc[w] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* p = partial + 2 * ((long)blockIdx.y * gridDim.x + blockIdx.x);
    p[0] = (rs[0] + rs[1]) + (rs[2] + rs[3]);
    p[1] = (rc[0] + rc[1]) + (rc[2] + rc[3]);
  }
}

// one workgroup per group: out[g][0] = loss = sum / count (nan when nothing is counted, as torch), out[g][1] = count
__global__ __launch_bounds__(64) void ce_finalize_kernel(const float* __restrict__ partial, int nblocks,
                                                         float* __restrict__ out) {
  float s = 0.f, c = 0.f;
  partial += 2 * (long)blockIdx.x * nblocks;
  for (int i = threadIdx.x; i < nblocks; i += 64) { s += partial[2 * i]; c += partial[2 * i + 1]; }
  s = wave_sum(s);
  c = wave_sum(c);
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = s / c; out[2 * blockIdx.x + 1] = c; }
}

// dlogits[t][c] = (softmax(logits[t])[c] - [c == label]) * dloss[g] / count[g] for counted tokens of group g, else 0
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits,
                                                     const int64_t* __restrict__ labels, long T, int C,
                                                     long long ignore, long group_tokens,
                                                     const float* __restrict__ dloss,
                                                     const float* __restrict__ loss_count,
                                                     float* __restrict__ dlogits) {
  const float scale = dloss[blockIdx.y] / loss_count[2 * blockIdx.y + 1];
  const long t_lo = (long)blockIdx.y * group_tokens, t_hi = min(T, t_lo + group_tokens);
  for (long t = t_lo + (long)blockIdx.x * 256 + threadIdx.x; t < t_hi; t += (long)gridDim.x * 256) {
    const long long lab = labels[t];
    float v[CE_MAXC];
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < CE_MAXC; ++c) {
      v[c] = c < C ? logits[t * C + c] : -INFINITY;
      m = fmaxf(m, v[c]);
    }
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < CE_MAXC; ++c) { v[c] = c < C ? expf(v[c] - m) : 0.f; se += v[c]; }
    const float inv = (lab == ignore) ? 0.f : scale / se;
#pragma unroll
    for (int c = 0; c < CE_MAXC; ++c)
      if (c < C) dlogits[t * C + c] = (lab == ignore) ? 0.f : (v[c] * inv - ((c == (int)lab) ? scale : 0.f));
  }
}

// ------------------------------------------------------------------------------------------
// accuracy of the S1 loop (s1_pretrain_rna.py:308-312): prediction = 1 + argmax over the classes 1..C-2 (the
// first maximum wins, as torch.argmax), counted where label != pad.  grid (chunks, cells): every workgroup leaves
// its (correct, counted) pair in ws[cell][chunk]; a second small launch adds a cell's pairs up in chunk order and
// writes counts[cell] and ratio[cell] = correct / counted (nan for a cell with nothing masked).  No atomics and
// nothing to clear between calls: the result does not depend on what the buffers held before.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void s1_accuracy_kernel(const float* __restrict__ logits,
                                                          const int64_t* __restrict__ labels, int N, int C,
                                                          long long pad_id, int* __restrict__ ws) {
  __shared__ int rc[4], rn[4];
  const int cell = blockIdx.y;
  const long base = (long)cell * N;
  int correct = 0, counted = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) {
    const long long lab = labels[base + i];
    if (lab == pad_id) continue;
    const float* v = logits + (base + i) * C;
    int best = 1;
    float bv = v[1];
    for (int c = 2; c < C - 1; ++c) {
      const float x = v[c];
      if (x > bv) { bv = x; best = c; }
    }
    counted += 1;
    correct += (best == (int)lab) ? 1 : 0;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { correct += __shfl_xor(correct, o); counted += __shfl_xor(counted, o); }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { rc[w] = correct; rn[w] = counted; }
  __syncthreads();
  if (threadIdx.x == 0) {
    int* p = ws + 2 * ((long)cell * gridDim.x + blockIdx.x);
    p[0] = (rc[0] + rc[1]) + (rc[2] + rc[3]);
    p[1] = (rn[0] + rn[1]) + (rn[2] + rn[3]);
  }
}

__global__ __launch_bounds__(64) void s1_accuracy_finalize_kernel(const int* __restrict__ ws, int chunks, int B,
                                                                  float* __restrict__ ratio, int* __restrict__ counts) {
  const int cell = blockIdx.x * 64 + threadIdx.x;
  if (cell >= B) return;
  int c = 0, n = 0;
  for (int k = 0; k < chunks; ++k) { c += ws[2 * ((long)cell * chunks + k)]; n += ws[2 * ((long)cell * chunks + k) + 1]; }
  counts[2 * cell] = c;
  counts[2 * cell + 1] = n;
  ratio[cell] = (float)c / (float)n;
}

// ---------------------------------------------------------------- host launchers
int s1_accuracy_chunks(int N) { return (N + 255) / 256 < 64 ? (N + 255) / 256 : 64; }
hipError_t s1_accuracy(const float* logits, const int64_t* labels, int B, int N, int C, long long pad_id, float* ratio,
                       int* counts, int* ws, hipStream_t s) {
  Traced tr(K_CE, 0, s);
  const int chunks = s1_accuracy_chunks(N);
  hipLaunchKernelGGL(s1_accuracy_kernel, dim3(chunks, B), dim3(256), 0, s, logits, labels, N, C, pad_id, ws);
  hipLaunchKernelGGL(s1_accuracy_finalize_kernel, dim3((B + 63) / 64), dim3(64), 0, s, ws, chunks, B, ratio, counts);
  return hipGetLastError();
}
hipError_t s1_mask(const int64_t* tok, const float* rand_pick, const float* rand_replace, int B, int N, float mask_prob,
                   float replace_prob, int max_masked, long long pad_id, long long mask_id, int64_t* masked,
                   int64_t* labels, hipStream_t s) {
  Traced tr(K_S1_MASK, 0, s);
  hipLaunchKernelGGL(s1_mask_kernel, dim3(B), dim3(1024), 0, s, tok, rand_pick, rand_replace, N, mask_prob,
                     replace_prob, max_masked, pad_id, mask_id, masked, labels);
  return hipGetLastError();
}
int ce_blocks(long T) { long b = (T + 255) / 256; return (int)(b < 256 ? (b < 1 ? 1 : b) : 256); }
int ce_groups(long T, long group_tokens) { return (int)((T + group_tokens - 1) / group_tokens); }
hipError_t ce_fwd(const float* logits, const int64_t* labels, long T, int C, long long ignore, long group_tokens,
                  float* partial, float* out, hipStream_t s) {
  Traced tr(K_CE, 0, s);
  const int nb = ce_blocks(group_tokens), ng = ce_groups(T, group_tokens);
  hipLaunchKernelGGL(ce_fwd_kernel, dim3(nb, ng), dim3(256), 0, s, logits, labels, T, C, ignore, group_tokens, partial);
  hipLaunchKernelGGL(ce_finalize_kernel, dim3(ng), dim3(64), 0, s, partial, nb, out);
  return hipGetLastError();
}
hipError_t ce_bwd(const float* logits, const int64_t* labels, long T, int C, long long ignore, long group_tokens,
                  const float* dloss, const float* loss_count, float* dlogits, hipStream_t s) {
  Traced tr(K_CE, 0, s);
  hipLaunchKernelGGL(ce_bwd_kernel, dim3(ce_blocks(group_tokens), ce_groups(T, group_tokens)), dim3(256), 0, s, logits,
                     labels, T, C, ignore, group_tokens, dloss, loss_count, dlogits);
  return hipGetLastError();
}

}  // namespace scm
