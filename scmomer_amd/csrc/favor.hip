// FAVOR+ attention (softmax_kernel + linear_attention, performer_pytorch.py:67-92,144-149)
// forward and backward for head dim 64 and 266 random features, bf16 MFMA with f32 accumulate.
//
// Notation per (cell b, head h), c = 64^-1/4, eps = 1e-4; the 266^-1/2 ratio of the reference
// cancels between numerator and denominator and never appears:
//   Eq[n,m] = exp(c*q_n.w_m - c^2/2 |q_n|^2 - c*max_m(q_n.w_m))          queries: per-token stabiliser
//   Ek[n,m] = exp(c*k_n.w_m - c^2/2 |k_n|^2 - c*max_{b,h,n,m}(k_n.w_m))  keys: ONE stabiliser per call
//   C[m,:] = sum_n (Ek+eps) v_n     z[m] = sum_n (Ek+eps)
//   o_n = ((Eq+eps) C) / ((Eq+eps) z)
// The eps terms are carried analytically (column sums) so the MFMA operands are pure
// exponentials, whose bf16 rounding is relative at any magnitude.  The context is used CENTRED:
// with obar = colsum(C)/sum(z), C' = C - z obar^T and o_n = obar + ((Eq+eps) C')_n / den_n.
// When attention is near uniform every row of C/z is close to obar; centring keeps the bf16
// rounding relative to the (small) deviations, in the forward and in the backward, where
// dE'q = dnum C'^T - (dO . (o - obar)) z / den is otherwise a difference of nearly equal terms.
//
// Kernels
//   favor_ctx<MODE>     "feature-stationary": wave w owns features 32w..32w+31 (its slice of the
//                       projection matrix stays in registers) and accumulates its 32x64 slice of
//                       sum_n E[n,:]^T y_n over a token chunk.  MODE 0: (k, v) -> C, z.
//                       MODE 1: (q, dO/den) -> dC, dz for the backward.
//   favor_ctx_finalize  sums chunk partials, adds the eps terms, emits bf16 image + f32 vectors
//   favor_apply<MODE>   "token-stationary": a wave owns 32 tokens (token = MFMA column, features
//                       in accumulator rows) so every later product contracts over accumulator
//                       rows with no lane movement.  MODE 0: o, rowmax, den.  MODE 1: dq, dden.
//                       MODE 2: dk, dv.
#include "scm_launch.hpp"
#include <type_traits>

namespace scm {

constexpr int STAGE_PITCH = 68;                     // floats per staged output row (64 + 4 spare)
// favor_apply LDS: Omega and context images, z, sums, 8 output stages
constexpr int APPLY_LDS_MAIN = 2 * NFP * 128 + (NFP + SUMS_STRIDE) * 4 + 8 * 32 * STAGE_PITCH * 4;
constexpr int APPLY_LDS = APPLY_LDS_MAIN;
constexpr int APPLY_LDS_DQ = 2 * NFP * 128 + (NFP + SUMS_STRIDE) * 4 + 12 * 16 * STAGE_PITCH * 4;   // 12 waves, half-tile stages

SCM_DEV void load_image(char* img, const bf16* __restrict__ src /*[288][64]*/, int tid, int nthreads) {
  for (int c = tid; c < NFP * 8; c += nthreads) {
    const int row = c >> 3, ch = c & 7;
    *(bf16x8*)(img + img_off(row, 8 * ch)) = *(const bf16x8*)(src + row * 64 + 8 * ch);
  }
}

SCM_DEV float reduce_partial_max(const float* __restrict__ part, int n) {
  float m = -INFINITY;
  for (int i = lane_id(); i < n; i += 64) m = fmaxf(m, part[i]);
  return wave_max(m);
}

SCM_DEV bf16x8 zero_frag() {
  bf16x8 z;
#pragma unroll
  for (int j = 0; j < 8; ++j) z[j] = (bf16)0.f;
  return z;
}

// fragment of x^T as MFMA B operand (k = d, column = token): 16 bytes of the lane's token row
SCM_DEV void load_token_frags(bf16x8 (&f)[4], const bf16* __restrict__ row, bool valid) {
  const int half = lane_id() >> 5;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) f[ks] = valid ? *(const bf16x8*)(row + 16 * ks + 8 * half) : zero_frag();
}

SCM_DEV float frag_sumsq(const bf16x8 (&f)[4]) {
  float s = 0.f;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float v = (float)f[ks][j]; s += v * v; }
  return xhalf_sum(s);
}

// Byte offsets of this lane's fragment reads inside one 32-row block (4096 bytes) of a swizzled image.
// The swizzle depends on row & 15 only, so block t of an image is at +4096*t with the same offsets:
// they are computed once per kernel instead of once per read.
struct LaneOffs {
  int row[4];        // row fragment (k along image columns) of k-step ks: rows l&31, columns 16ks + 8*half
  int tr[2][2][2];   // transposed fragment in accumulator k order: [k-step sk][32-column block dt][read]
};
SCM_DEV LaneOffs lane_offs() {
  LaneOffs o;
  const int l = lane_id(), half = l >> 5, g = l >> 4, i = l & 15, q = i >> 2;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) o.row[ks] = img_off(l & 31, 16 * ks + 8 * half);
#pragma unroll
  for (int sk = 0; sk < 2; ++sk)
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const int col = 32 * dt + 16 * (g & 1) + 4 * (i & 3);
      o.tr[sk][dt][0] = img_off(16 * sk + 4 * half + q, col);
      o.tr[sk][dt][1] = img_off(16 * sk + 8 + 4 * half + q, col);
    }
  return o;
}
SCM_DEV bf16x8 tr_frag_at(const char* blk, const int (&o)[2]) {
  const bf16x4 lo4 = lds_tr_read(blk + o[0]);
  const bf16x4 hi4 = lds_tr_read(blk + o[1]);
  bf16x8 f;
  f[0] = lo4[0]; f[1] = lo4[1]; f[2] = lo4[2]; f[3] = lo4[3];
  f[4] = hi4[0]; f[5] = hi4[1]; f[6] = hi4[2]; f[7] = hi4[3];
  return f;
}

// X[m,token] tile t = Omega_t (32x64) * x^T (64x32)
SCM_DEV f32x16 score_tile(const char* omg, int t, const bf16x8 (&xf)[4]) {
  f32x16 x = zero16();
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) x = mfma32(img_row_frag(omg, 32 * t, 16 * ks), xf[ks], x);
  return x;
}

// The last feature tile holds NF - 32 (NFT - 1) = 10 real features: accumulator rows 0..9, i.e. register groups
// g < LAST_GROUPS (rows 8g .. 8g+7) and k-steps sk < LAST_KSTEPS (rows 16 sk .. 16 sk + 15) of the products that
// contract over features.  Everything beyond is padding, known at compile time: no exponentials, no products.
constexpr int LAST_VALID = NF - 32 * (NFT - 1);
constexpr int LAST_GROUPS = (LAST_VALID + 7) / 8;
constexpr int LAST_KSTEPS = (LAST_VALID + 15) / 16;

// feature index of accumulator element r in tile t; valid iff < 266
SCM_DEV bool feat_valid(int t, int r, int half) { return t < NFT - 1 || acc_row(r, half) < NF - 32 * (NFT - 1); }

// ------------------------------------------------------------------------------------------
// prep: f32 projection matrix [266][64] -> zero padded bf16 [288][64]
// ------------------------------------------------------------------------------------------
__global__ void favor_prep_omega_kernel(const float* __restrict__ omega, bf16* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= NFP * DH) return;
  out[i] = (bf16)((i < NF * DH) ? omega[i] : 0.f);
}

// ------------------------------------------------------------------------------------------
// key stabiliser: max over (b,h,n,m) of k_n.w_m (raw, before the c scale)
// ------------------------------------------------------------------------------------------
// (value, key) argmax combine: larger value wins, equal values keep the smaller key (first index)
SCM_DEV void argmax_combine(float& v, long long& key, float ov, long long okey) {
  if (ov > v || (ov == v && okey < key)) { v = ov; key = okey; }
}
SCM_DEV long long wave_min_key(long long key) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const int lo = __shfl_xor((int)(key & 0xffffffffll), o);
    const int hi = __shfl_xor((int)(key >> 32), o);
    const long long ok = ((long long)hi << 32) | (unsigned int)lo;
    key = ok < key ? ok : key;
  }
  return key;
}
SCM_DEV void wave_argmax(float& v, long long& key) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const float ov = __shfl_xor(v, o);
    const int lo = __shfl_xor((int)(key & 0xffffffffll), o);
    const int hi = __shfl_xor((int)(key >> 32), o);
    argmax_combine(v, key, ov, ((long long)hi << 32) | (unsigned int)lo);
  }
}

// Gradient through the key stabiliser (torch.max over the whole tensor, performer_pytorch.py:90):
// dk[b*,h*,n*,:] += -c * (sum over everything of E * dE') * Omega[j*,:] at the argmax element.
// One workgroup per stabiliser group (group_bh consecutive (b,h) pairs: the cells that share one max); stab_per_bh
// and sum_per_bh are the numbers of stabiliser partials and of E * dE' partials per (b,h).
__global__ __launch_bounds__(256) void favor_kfix_kernel(const float* __restrict__ part_val,
                                                         const long long* __restrict__ part_key, int stab_per_bh,
                                                         const float* __restrict__ ksum_part, int sum_per_bh,
                                                         int BH, int group_bh,
                                                         const bf16* __restrict__ omega_b, TensorBHND_out dk, int H,
                                                         int N) {
  __shared__ float wtot[4], wmx[4];
  __shared__ long long wkey[4];
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
  const int bh0 = blockIdx.x * group_bh, nbh = min(group_bh, BH - bh0);
  part_val += (long)bh0 * stab_per_bh;
  part_key += (long)bh0 * stab_per_bh;
  ksum_part += (long)bh0 * sum_per_bh;
  const int nstab = nbh * stab_per_bh, nsum = nbh * sum_per_bh;
  float tot = 0.f;
  for (int i = tid; i < nsum; i += 256) tot += ksum_part[i];
  tot = wave_sum(tot);
  float mx = -INFINITY;
  long long key = 0x7fffffffffffffffll;
  for (int i = tid; i < nstab; i += 256) argmax_combine(mx, key, part_val[i], part_key[i]);
  wave_argmax(mx, key);
  if (l == 0) { wtot[w] = tot; wmx[w] = mx; wkey[w] = key; }
  __syncthreads();
  if (w != 0) return;
  tot = (wtot[0] + wtot[1]) + (wtot[2] + wtot[3]);
  mx = wmx[0]; key = wkey[0];
  for (int i = 1; i < 4; ++i) argmax_combine(mx, key, wmx[i], wkey[i]);
  if (!(mx > -INFINITY)) return;
  const int feat = (int)(key & 511);
  const long long tl = key >> 9;
  const int bh = (int)(tl / N), n = (int)(tl % N), b = bh / H, h = bh % H;
  bf16* p = dk.p + b * dk.sb + h * dk.sh + (long)n * dk.sn + l;
  *p = (bf16)((float)*p - CNORM * tot * (float)omega_b[feat * DH + l]);
}

// ------------------------------------------------------------------------------------------
// favor_ctx: partial[bh][chunk] = { sum_n E[n,m] y[n,d] , sum_n E[n,m] w_n , sum_n y[n,d] , sum_n w_n }
// ------------------------------------------------------------------------------------------
constexpr int CTX_TILE = 64;
constexpr int CTX_BUF = 2 * CTX_TILE * 128 + 2 * CTX_TILE * 4;  // x image, y image, off[], wgt[]

// 12 waves: nine compute waves (one per tile of 32 features) and three loader waves that bring the next token tile
// in, normalise it and write its LDS images while the compute waves work -- three waves on every SIMD, and the
// compute waves carry no loading work.
constexpr int CTX_LOADERS = 192;                        // threads of the loader waves
constexpr int CTX_LD_PER = (CTX_TILE * 8 + CTX_LOADERS - 1) / CTX_LOADERS;   // (row, 16-byte chunk) pairs per loader thread: 3
template <int MODE>
__global__ __launch_bounds__(768) void favor_ctx_kernel(CtxArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, l = tid & 63, wv = tid >> 6, half = l >> 5;
  const bool compute = __builtin_amdgcn_readfirstlane(wv) < NFT;   // wave-uniform
  const int w = compute ? wv : NFT - 1;                              // feature tile (loader waves: any valid one)
  const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
  const int n_lo = blockIdx.x * a.chunk_len, n_hi = min(a.N, n_lo + a.chunk_len);
  const bool loader = !compute;
  const int lt = tid - 64 * NFT;                                     // loader thread number

  // MODE 0 (keys): the reference stabilises with ONE max over the whole [B,H,N,M] score tensor
  // (performer_pytorch.py:89-90).  No separate pass computes it.  Every feature is exponentiated against its own
  // constant b_m = |w_m|^2 / 2:  exp(c k.w_m - c^2 |k|^2 / 2 - b_m) = exp(-|c k - w_m|^2 / 2) <= 1 for every key,
  // so nothing can overflow; the waves only TRACK the running max of the raw scores (and where it is attained),
  // one record per (b,h, chunk, wave), and the chunk reduction applies exp(b_m - c max) per feature once the
  // global max is known.
  float lane_max = -INFINITY;   // of the shifted score s - b_m/c, over this lane's feature column
  int lane_pos = 0;             // 32-token sub-tile (relative to the chunk) where it was first attained

  // this wave's slice of the projection matrix as MFMA B operand (k = d, column = feature)
  bf16x8 omf[4];
  {
    const bf16* orow = a.omega_b + (32 * w + (l & 31)) * DH;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) omf[ks] = *(const bf16x8*)(orow + 16 * ks + 8 * half);
  }
  const bool lane_feat_valid = (32 * w + (l & 31)) < NF;
  const bool pad_wave = __builtin_amdgcn_readfirstlane(w) == NFT - 1;   // wave-uniform
  const LaneOffs lo = lane_offs();
  // b_m / c for this lane's feature: the score accumulator starts there instead of at zero (MODE 0)
  float sinit = 0.f;
  if (MODE == 0) {
    float ss = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float v = (float)omf[ks][j]; ss += v * v; }
    sinit = -0.5f * xhalf_sum(ss) / CNORM;
  }

  f32x16 cacc[2] = {zero16(), zero16()};
  float zacc = 0.f;
  // Feature 266 (the first padding feature) is a "ones" feature: its E is 1 for every token, so its context row
  // accumulates sum_n y[n,:] and its z slot sum_n w_n -- the plain sums the eps terms need -- inside the same
  // MFMAs (padding tokens have y = 0 and w = 0).
  const float padv = (32 * w + (l & 31) == NF) ? 1.f : 0.f;

  const bf16* xb = a.x.p + b * a.x.sb + h * a.x.sh;
  const bf16* yb = a.y.p + b * a.y.sb + h * a.y.sh;
  const long srow = (long)bh * a.N;

  // loader thread lt handles the pairs (row, chunk) number lt, lt + 192, lt + 384 of the 64 x 8 of a tile: the eight
  // lanes of a row stay together (192 is a multiple of 8), which the row's sum of squares relies on
  bf16x8 xs_r[CTX_LD_PER], ys_r[CTX_LD_PER];
  float st_r[CTX_LD_PER], den_r[CTX_LD_PER], w_r[CTX_LD_PER];
  bool val_r[CTX_LD_PER];
  const int lch = lt & 7;
  auto load_tile = [&](int n0) {
    if (!loader) return;
#pragma unroll
    for (int k = 0; k < CTX_LD_PER; ++k) {
      const int lrow = (lt >> 3) + (CTX_LOADERS / 8) * k;
      const int n = n0 + lrow;
      val_r[k] = lrow < CTX_TILE && n < n_hi;
      xs_r[k] = zero_frag(); ys_r[k] = zero_frag(); st_r[k] = 0.f; den_r[k] = 1.f; w_r[k] = 0.f;
      if (val_r[k]) {
        xs_r[k] = *(const bf16x8*)(xb + (long)n * a.x.sn + 8 * lch);
        ys_r[k] = *(const bf16x8*)(yb + (long)n * a.y.sn + 8 * lch);
        if (MODE == 1) { st_r[k] = a.stab[srow + n]; den_r[k] = a.den[srow + n]; w_r[k] = a.dden[srow + n]; }
      }
    }
  };
  auto write_tile = [&](char* buf) {
    if (!loader) return;
#pragma unroll
    for (int k = 0; k < CTX_LD_PER; ++k) {
      const int lrow = (lt >> 3) + (CTX_LOADERS / 8) * k;
      float ss = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float v = (float)xs_r[k][j]; ss += v * v; }
      ss += dpp_move<0xB1>(ss); ss += dpp_move<0x4E>(ss); ss += dpp_move<0x141>(ss);   // the 8 lanes of a row
      bf16x8 yv = ys_r[k];
      if (MODE == 1) {
        const float inv = 1.f / den_r[k];
#pragma unroll
        for (int j = 0; j < 8; ++j) yv[j] = (bf16)((float)ys_r[k][j] * inv);
      }
      if (lrow < CTX_TILE) {
        *(bf16x8*)(buf + img_off(lrow, 8 * lch)) = xs_r[k];
        *(bf16x8*)(buf + CTX_TILE * 128 + img_off(lrow, 8 * lch)) = yv;
        if (lch == 0) {
          const float stab = (MODE == 0) ? 0.f : st_r[k];   // MODE 0: the stabiliser is applied by the consuming wave
          float* off = (float*)(buf + 2 * CTX_TILE * 128);
          off[lrow] = val_r[k] ? (HALF_C2 * ss + CNORM * stab) * LOG2E : INFINITY;
          off[CTX_TILE + lrow] = (MODE == 1) ? w_r[k] : (val_r[k] ? 1.f : 0.f);
        }
      }
    }
  };

  const int ntiles = (n_hi - n_lo + CTX_TILE - 1) / CTX_TILE;
  if (ntiles > 0) { load_tile(n_lo); write_tile(smem); }
  __syncthreads();
  for (int it = 0; it < ntiles; ++it) {
    char* buf = smem + (it & 1) * CTX_BUF;
    char* nbuf = smem + ((it + 1) & 1) * CTX_BUF;
    if (it + 1 < ntiles) load_tile(n_lo + (it + 1) * CTX_TILE);
    const char* xs = buf;
    const char* ys = buf + CTX_TILE * 128;
    const float* off = (const float*)(buf + 2 * CTX_TILE * 128);
    const float* wg = off + CTX_TILE;
#pragma unroll
    for (int sub = 0; sub < (compute ? 2 : 0); ++sub) {
      // S[token, feature] = x_tile (32x64) * Omega_w^T (64x32)
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = sinit;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) s = mfma32(*(const bf16x8*)(xs + 4096 * sub + lo.row[ks]), omf[ks], s);
      if (MODE == 0) {
        float lm = -INFINITY;   // max of the (shifted) scores over this sub-tile's valid tokens
        if (n_lo + (it + 1) * CTX_TILE <= n_hi) {   // full tile (wave-uniform): every token counts
#pragma unroll
          for (int r = 0; r < 16; ++r) lm = fmaxf(lm, s[r]);
        } else {                                     // last tile of the chunk: off = inf marks padding rows
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 o4 = *(const f32x4*)(off + 32 * sub + 8 * g + 4 * half);
#pragma unroll
            for (int i = 0; i < 4; ++i) lm = fmaxf(lm, o4[i] < INFINITY ? s[4 * g + i] : -INFINITY);
          }
        }
        // per-lane running max of this feature's column and the first sub-tile that attained it: no cross-lane
        // work in the loop (the wave-level argmax is resolved once, after the loop)
        const bool up = lm > lane_max;
        lane_max = up ? lm : lane_max;
        lane_pos = up ? 2 * it + sub : lane_pos;
      }
      f32x16 e;
      auto features = [&](auto pad_c) {   // only the last wave owns padding features: the select is compiled out elsewhere
        constexpr bool PAD = decltype(pad_c)::value;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 o4 = *(const f32x4*)(off + 32 * sub + 8 * g + 4 * half);
          const f32x4 w4 = *(const f32x4*)(wg + 32 * sub + 8 * g + 4 * half);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float v = exp2_fast(s[4 * g + i] * (CNORM * LOG2E) - o4[i]);
            if (PAD) v = lane_feat_valid ? v : padv;
            e[4 * g + i] = v;
            zacc += v * w4[i];
          }
        }
      };
      if (MODE == 1 || pad_wave) features(std::true_type{}); else features(std::false_type{});
      // C_w[feature, d] += E^T (32 features x 32 tokens) * y (32 tokens x 64)
#pragma unroll
      for (int sk = 0; sk < 2; ++sk) {
        const bf16x8 ef = acc_to_frag(e, sk);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
          cacc[dt] = mfma32(ef, tr_frag_at(ys + 4096 * sub, lo.tr[sk][dt]), cacc[dt]);
      }
    }
    if (it + 1 < ntiles) write_tile(nbuf);
    __syncthreads();
  }

  if (!compute) return;     // the loader waves are done (no barrier follows)
  // The arg-max resolution below reads global memory: it comes BEFORE the partial record is stored (a load issued
  // after the stores would wait for them to retire).
  if (MODE == 0) {
    // wave-level argmax: the max of the raw scores, and the first (token, feature) attaining it.  The sub-tile
    // that holds it is recomputed from global memory with the same MFMA sequence (bit-identical scores).
    const float lraw = lane_feat_valid ? lane_max - sinit : -INFINITY;
    const float wm = wave_max(lraw);
    int pos = (lraw == wm) ? lane_pos : (1 << 30);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) pos = min(pos, __shfl_xor(pos, o));
    long long key = 0x7fffffffffffffffll;
    if (wm > -INFINITY) {
      const int nbase = n_lo + 32 * pos;
      const int n = nbase + (l & 31);
      f32x16 sc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[r] = sinit;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 xf = (n < n_hi) ? *(const bf16x8*)(xb + (long)n * a.x.sn + 16 * ks + 8 * half) : zero_frag();
        sc = mfma32(xf, omf[ks], sc);
      }
      int best = 1 << 30;
#pragma unroll
      for (int r = 15; r >= 0; --r) {
        const bool hit = (sc[r] - sinit == wm) && (nbase + acc_row(r, half) < n_hi);
        best = hit ? acc_row(r, half) : best;
      }
      if (lane_feat_valid && best < (1 << 30))
        key = ((long long)bh * a.N + nbase + best) * 512 + 32 * w + (l & 31);
      key = wave_min_key(key);
    }
    if (l == 0) {
      const long slot = ((long)bh * gridDim.x + blockIdx.x) * NFT + w;
      a.stab_out[slot] = wm;
      a.key_out[slot] = key;
    }
  }
  // partial record
  float* rec = a.part + ((long)bh * gridDim.x + blockIdx.x) * CTX_PART;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) rec[(32 * w + acc_row(r, half)) * DH + 32 * dt + (l & 31)] = cacc[dt][r];
  const float zt = xhalf_sum(zacc);
  if (l < 32) rec[NFP * DH + 32 * w + l] = zt;
}

// stab (forward keys only, else null): the per-(b,h, chunk, wave) running maxima of the raw key scores, stab_per_bh
// of them per (b,h).  The partials were taken against the per-feature constant b_m = |w_m|^2 / 2; here they are
// brought to the reference's stabiliser, the max over the whole call -- or, when several calls of the reference
// are batched into one launch, over the group_bh consecutive (b,h) pairs that formed one call: factor
// exp(b_m - c max(stab of the group)) per feature.
// NCH > 0: the number of chunks as a compile-time constant (the usual 8): every load of the workgroup is then
// issued before the first sum needs one, instead of one round trip per chunk; the sums run in the same order.
// BWD (the backward's dC, dz; no stabiliser): the finalize step is done here as well -- it needs nothing from other
// tiles: Mb = bf16(dC) and zf[m] = dz[m] - dC[m,:] . obar_fwd are written directly (M32 / zz are not), and the
// consumer adds up the nine column partials itself.
template <int NCH, bool BWD>
__global__ __launch_bounds__(256) void favor_ctx_reduce_kernel(const float* __restrict__ part, int nchunks_rt,
                                                               const float* __restrict__ stab, int stab_per_bh,
                                                               int group_bh,
                                                               const bf16* __restrict__ omega_b,
                                                               float* __restrict__ gmax_out /*sums [bh][80]*/,
                                                               float* __restrict__ M32, float* __restrict__ zz,
                                                               float* __restrict__ colpart /*[bh][9][80]*/,
                                                               const float* __restrict__ sums_fwd, bf16* __restrict__ Mb,
                                                               float* __restrict__ zf) {
  __shared__ float ysum[DH];
  __shared__ float wsum;
  __shared__ float colp[4][DH];
  __shared__ float zl[32];
  const int tid = threadIdx.x, bh = blockIdx.x, tile = blockIdx.y;
  const int nchunks = NCH > 0 ? NCH : nchunks_rt;
  const float* base = part + (long)bh * nchunks * CTX_PART;
  __shared__ float wmaxs[4];
  __shared__ float fscale[32];
  __shared__ float zraw[32];
  // Everything this workgroup reads is independent of what it computes, so all loads are issued up front (the
  // chunk sums of its 32 x 64 context entries first) and the dependent steps follow after ONE round trip.
  const int d = tid & 63, q = tid >> 6;
  const float ob = BWD ? sums_fwd[(long)bh * SUMS_STRIDE + d] : 0.f;   // forward obar[d]
  float acc8[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc8[i] = 0.f;
  if (NCH > 0) {
    float v[NCH > 0 ? NCH : 1][8];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const float* rec = base + (long)c * CTX_PART + (32 * tile + q) * DH + d;
#pragma unroll
      for (int i = 0; i < 8; ++i) v[c][i] = rec[4 * i * DH];
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc8[i] += v[c][i];
  } else {
    for (int c = 0; c < nchunks; ++c) {   // 8 independent loads in flight per chunk
      const float* rec = base + (long)c * CTX_PART + (32 * tile + q) * DH + d;
#pragma unroll
      for (int i = 0; i < 8; ++i) acc8[i] += rec[4 * i * DH];
    }
  }
  float m = -INFINITY;
  if (stab) {   // the partial maxima of every (b,h) of this stabiliser group
    const int bh0 = bh / group_bh * group_bh;
    const int nstab = min(group_bh, (int)gridDim.x - bh0) * stab_per_bh;
    const float* sg = stab + (long)bh0 * stab_per_bh;
#pragma unroll 8
    for (int i = tid; i < nstab; i += 256) m = fmaxf(m, sg[i]);
  }
  if (tid < DH) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < nchunks; ++c) s += base[(long)c * CTX_PART + NF * DH + tid];   // row of the ones feature
    ysum[tid] = s;
  } else if (tid == 64) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < nchunks; ++c) s += base[(long)c * CTX_PART + NFP * DH + NF];    // z slot of the ones feature
    wsum = s;
  } else if (tid >= 128 && tid < 160) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < nchunks; ++c) s += base[(long)c * CTX_PART + NFP * DH + 32 * tile + (tid - 128)];
    zraw[tid - 128] = s;
  }
  // |w_m|^2 of this tile's 32 features: 8 threads per feature, 16 bytes each
  float ss = 0.f;
  if (stab) {
    const bf16x8 ov = *(const bf16x8*)(omega_b + (long)(32 * tile + (tid >> 3)) * DH + 8 * (tid & 7));
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float v = (float)ov[j]; ss += v * v; }
    ss += dpp_move<0xB1>(ss); ss += dpp_move<0x4E>(ss); ss += dpp_move<0x141>(ss);   // the 8 lanes of a row
  }
  float gmax = 0.f;
  if (stab) {   // max over all partial stabilisers, by the whole workgroup
    m = wave_max(m);
    if ((tid & 63) == 0) wmaxs[tid >> 6] = m;
    __syncthreads();
    gmax = fmaxf(fmaxf(wmaxs[0], wmaxs[1]), fmaxf(wmaxs[2], wmaxs[3]));
    if (tile == 0 && tid == 0) gmax_out[(long)bh * SUMS_STRIDE + DH + 1] = gmax;   // for the backward passes
  }
  // per-feature factor exp(b_m - c max) of this tile's 32 features (1 when there is no stabiliser: backward)
  if ((tid & 7) == 0) fscale[tid >> 3] = stab ? exp2_fast((0.5f * ss - CNORM * gmax) * LOG2E) : 1.f;
  __syncthreads();
  if (tid < 32) {
    const int mm = 32 * tile + tid;
    const float v = (mm < NF) ? fscale[tid] * zraw[tid] + KEPS * wsum : 0.f;
    zl[tid] = v;
    if (!BWD) zz[(long)bh * NFP + mm] = v;
  }
  if (BWD) __syncthreads();   // zl is read below
  float csum = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = 32 * tile + q + 4 * i;
    const float v = (m < NF) ? fscale[q + 4 * i] * acc8[i] + KEPS * ysum[d] : 0.f;
    if (BWD) {
      Mb[((long)bh * NFP + m) * DH + d] = (bf16)v;
      const float dot = wave_sum(v * ob);  // the 64 lanes of this wave hold row m
      if (d == 0) zf[(long)bh * NFP + m] = (m < NF) ? zl[q + 4 * i] - dot : 0.f;
    } else {
      M32[((long)bh * NFP + m) * DH + d] = v;
    }
    csum += v;
  }
  colp[q][d] = csum;
  __syncthreads();
  float* cp = colpart + ((long)bh * NFT + tile) * SUMS_STRIDE;
  if (tid < DH) cp[tid] = colp[0][tid] + colp[1][tid] + colp[2][tid] + colp[3][tid];
  if (tid == 64) {
    float s = 0.f;
    for (int i = 0; i < 32; ++i) s += zl[i];
    cp[DH] = s;
  }
}

// Finalize, stage 2 (grid: bh x 9):
//  mode 0 (forward context): z = zz, obar[d] = colsum(M)[d] / sum(z), writes the CENTRED context
//      Mb[m,d] = bf16(M[m,d] - z[m] * obar[d]), zf = z, sums[0..63] = obar, sums[64] = sum(z).
//  mode 1 (backward): Mb = bf16(M) = dC, sums[0..63] = colsum(dC), and, the partial z-sums having been
//      taken against the centred denominator gradient, zf[m] = zz[m] - dC[m,:] . obar_fwd.
__global__ __launch_bounds__(256) void favor_ctx_finalize_kernel(const float* __restrict__ M32,
                                                                 const float* __restrict__ zz,
                                                                 const float* __restrict__ colpart, int mode,
                                                                 const float* __restrict__ sums_fwd,
                                                                 bf16* __restrict__ Mb, float* __restrict__ zf,
                                                                 float* __restrict__ sums) {
  __shared__ float obar[DH];
  __shared__ float cs[DH + 1];
  const int tid = threadIdx.x, bh = blockIdx.x, tile = blockIdx.y;
  if (tid <= DH) {
    float s = 0.f;
    for (int t = 0; t < NFT; ++t) s += colpart[((long)bh * NFT + t) * SUMS_STRIDE + tid];
    cs[tid] = s;
  }
  __syncthreads();
  if (tid < DH) obar[tid] = (mode == 0) ? cs[tid] / cs[DH] : sums_fwd[(long)bh * SUMS_STRIDE + tid];
  __syncthreads();
  const int d = tid & 63, q = tid >> 6;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = 32 * tile + q + 4 * i;
    const float v = M32[((long)bh * NFP + m) * DH + d];
    const float z = zz[(long)bh * NFP + m];
    if (mode == 0) {
      Mb[((long)bh * NFP + m) * DH + d] = (bf16)((m < NF) ? v - z * obar[d] : 0.f);
      if (d == 0) zf[(long)bh * NFP + m] = z;
    } else {
      Mb[((long)bh * NFP + m) * DH + d] = (bf16)v;
      const float dot = wave_sum(v * obar[d]);  // the 64 lanes of this wave hold row m
      if (d == 0) zf[(long)bh * NFP + m] = (m < NF) ? z - dot : 0.f;
    }
  }
  if (tile == 0) {
    if (tid < DH) sums[(long)bh * SUMS_STRIDE + tid] = (mode == 0) ? obar[tid] : cs[tid];
    if (tid == 0) sums[(long)bh * SUMS_STRIDE + DH] = (mode == 0) ? cs[DH] : 0.f;
  }
}

// ------------------------------------------------------------------------------------------
// favor_apply
// ------------------------------------------------------------------------------------------
// stage[token][d..d+3] for the 4-register groups of a [d rows][token cols] accumulator pair
SCM_DEV void stage_rows(float* stage, const f32x16 (&y)[2], float scale, const float* add4 /*LDS [64] or null*/,
                        float addscale) {
  const int l = lane_id(), tok = l & 31, half = l >> 5;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d = 32 * dt + 8 * g + 4 * half;
      f32x4 v;
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = y[dt][4 * g + i] * scale;
      if (add4) {
        const f32x4 c4 = *(const f32x4*)(add4 + d);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (y[dt][4 * g + i] + addscale * c4[i]) * scale;
      }
      *(f32x4*)(stage + tok * STAGE_PITCH + d) = v;
    }
}

// The same for one half of the tile (tokens 16 pass .. 16 pass + 15) into a 16-row stage: the 12-wave form of the dq
// pass has room for half a tile per wave only.
SCM_DEV void stage_rows_half(float* stage, const f32x16 (&y)[2], float scale, int pass) {
  const int l = lane_id(), tok = l & 31, half = l >> 5;
  if ((tok >> 4) != pass) return;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d = 32 * dt + 8 * g + 4 * half;
      f32x4 v;
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = y[dt][4 * g + i] * scale;
      *(f32x4*)(stage + (tok & 15) * STAGE_PITCH + d) = v;
    }
}

// coalesced write-back of a staged [32][64] f32 tile as bf16; optional rank-1 correction
// out[row][d] = stage[row][d] + stage[row][64] * corr_scale * xrow[d]
// and, with omg != null, + stage[row][65] * Omega[(int)stage[row][66]][d] (Omega from its LDS image)
// The x rows of a tile in the write-back layout (row 8i + (l>>3), columns 8(l&7)..+7), for the correction term of
// store_rows.  Fetched BEFORE the tile's first store: a load issued after a store waits for that store to retire.
SCM_DEV void load_x_rows(bf16x8 (&xr)[4], brsrc rxx, long xsn, int n0, int n_hi) {
  const int l = lane_id();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + 8 * i + (l >> 3);
    xr[i] = as_bf16x8(buf_load16(rxx, n < n_hi ? (uint32_t)((long)n * xsn + 8 * (l & 7)) * 2 : BUF_OOB));
  }
}

// I0, NI: the groups of 8 rows written (all four, or the two of a half-tile stage whose row 0 is tile row 8 I0)
template <int I0 = 0, int NI = 4>
SCM_DEV void store_rows(const float* stage, brsrc ro, long osn, const bf16x8* xr /*4 rows or null*/, float corr_scale,
                        int n0, int n_hi, const char* omg = nullptr, const float* colbias = nullptr) {
  const int l = lane_id();
#pragma unroll
  for (int i = I0; i < I0 + NI; ++i) {
    const int trow = 8 * i + (l >> 3), row = trow - 8 * I0, ch = l & 7;
    const int n = n0 + trow;
    const bool ok = n < n_hi;     // rows past the chunk: stores are dropped (no branches)
    const f32x4 a = *(const f32x4*)(stage + row * STAGE_PITCH + 8 * ch);
    const f32x4 b = *(const f32x4*)(stage + row * STAGE_PITCH + 8 * ch + 4);
    float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    if (colbias) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += colbias[8 * ch + j];
    }
    if (xr) {
      const float cs = stage[row * STAGE_PITCH + 64] * corr_scale;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += cs * (float)xr[i][j];
    }
    if (omg) {
      const float os = stage[row * STAGE_PITCH + 65];
      const int jr = ((const int*)stage)[row * STAGE_PITCH + 66];
      const bf16x8 ov = *(const bf16x8*)(omg + img_off(jr, 8 * ch));
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += os * (float)ov[j];
    }
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16)v[j];
    buf_store16(ro, ok ? (uint32_t)((long)n * osn + 8 * ch) * 2 : BUF_OOB, as_u32x4(o));
  }
}

// The dq pass (MODE 1) needs 166 registers, which leaves room for three waves per SIMD: it runs 12 waves per
// workgroup, each with a half-tile output stage (written back in two passes); the other passes run 8.
template <int MODE>
__global__ __launch_bounds__(MODE == 1 ? 768 : 512) void favor_apply_kernel(ApplyArgs a) {
  constexpr int NW = MODE == 1 ? 12 : 8;              // waves per workgroup
  constexpr int NTH = 64 * NW, TRIP = 32 * NW;        // threads; tokens per trip of the workgroup
  constexpr int STAGE_ROWS = MODE == 1 ? 16 : 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* omg = smem;                                   // Omega image
  char* mim = smem + NFP * 128;                       // C or dC image
  float* zv = (float*)(smem + 2 * NFP * 128);         // [288]
  float* sm = zv + NFP;                               // [80]
  float* stage_all = sm + SUMS_STRIDE;                // NW x [STAGE_ROWS][68]
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, half = l >> 5;
  const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
  const int n_lo = blockIdx.x * a.chunk_len, n_hi = min(a.N, n_lo + a.chunk_len);
  float* stage = stage_all + w * STAGE_ROWS * STAGE_PITCH;

  load_image(omg, a.omega_b, tid, NTH);
  load_image(mim, a.Mb + (long)bh * NFP * DH, tid, NTH);
  for (int i = tid; i < NFP; i += NTH) zv[i] = a.zf[(long)bh * NFP + i];
  if (tid < SUMS_STRIDE) {
    if (MODE == 2 && a.colpart) {   // colsum(dC) = the nine tile partials of the backward context reduction, in order
      float cs = 0.f;
#pragma unroll
      for (int t = 0; t < NFT; ++t) cs += a.colpart[((long)bh * NFT + t) * SUMS_STRIDE + tid];
      sm[tid] = tid < DH ? cs : 0.f;
    } else {
      sm[tid] = a.sums[(long)bh * SUMS_STRIDE + tid];
    }
  }
  float gstab = 0.f;
  if (MODE == 2) gstab = reduce_partial_max(a.stab + (long)bh * a.stab_stride, a.nstab);
  __syncthreads();
  const float zsum = sm[DH];

  const bf16* xb = a.x.p + b * a.x.sb + h * a.x.sh;
  const long srow = (long)bh * a.N;
  float ksum = 0.f;  // MODE 2: sum over this wave's tokens of sum_m dA
  const LaneOffs lo = lane_offs();
  // Every streamed tensor of this (b,h) goes through a bounds-checked buffer resource (rows past the chunk pass
  // BUF_OOB: loads return 0, stores are dropped), so the loop body has no exec-mask branches around memory
  // operations.  The fragments of the wave's NEXT tile are fetched right before the current tile's output is
  // stored: memory operations complete in issue order, so loads issued after the stores would wait for them.
  const brsrc rx = make_rsrc(xb, (uint32_t)(((long)(a.N - 1) * a.x.sn + DH) * 2));
  const bf16* yb_ = (MODE != 0) ? a.y.p + b * a.y.sb + h * a.y.sh : xb;
  const bf16* y2b_ = (MODE == 1) ? a.y2.p + b * a.y2.sb + h * a.y2.sh : xb;
  const brsrc ry = make_rsrc(yb_, (uint32_t)(((long)(a.N - 1) * a.y.sn + DH) * 2));
  const brsrc ry2 = make_rsrc(y2b_, (uint32_t)(((long)(a.N - 1) * a.y2.sn + DH) * 2));
  const brsrc ro0 = make_rsrc(a.out0.p + b * a.out0.sb + h * a.out0.sh, (uint32_t)(((long)(a.N - 1) * a.out0.sn + DH) * 2));
  const bool has_out1 = MODE != 1 && a.out1.p != nullptr;   // MODE 0: the centred output is the backward's tape only
  const bf16* o1b_ = has_out1 ? a.out1.p + b * a.out1.sb + h * a.out1.sh : xb;
  const brsrc ro1 = make_rsrc(o1b_, has_out1 ? (uint32_t)(((long)(a.N - 1) * a.out1.sn + DH) * 2) : 0u);
  const brsrc rden = make_rsrc(MODE == 1 ? a.den_in + srow : nullptr, MODE == 1 ? (uint32_t)a.N * 4 : 0u);
  const brsrc rstab = make_rsrc(MODE == 1 ? a.stab + srow : nullptr, MODE == 1 ? (uint32_t)a.N * 4 : 0u);
  auto fetch = [&](int n0, bf16x8 (&fx)[4], bf16x8 (&fy)[4], bf16x8 (&fy2)[4], float& den, float& rmx) {
    const int n = n0 + (l & 31);
    const bool v = n < n_hi;
    const uint32_t lane_off = (uint32_t)(8 * half) * 2;
    const uint32_t bx = v ? (uint32_t)((long)n * a.x.sn) * 2 + lane_off : BUF_OOB;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fx[ks] = as_bf16x8(buf_load16(rx, bx + 32 * ks));
    if (MODE != 0) {
      const uint32_t by = v ? (uint32_t)((long)n * a.y.sn) * 2 + lane_off : BUF_OOB;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) fy[ks] = as_bf16x8(buf_load16(ry, by + 32 * ks));
    }
    if (MODE == 1) {
      const uint32_t by2 = v ? (uint32_t)((long)n * a.y2.sn) * 2 + lane_off : BUF_OOB;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) fy2[ks] = as_bf16x8(buf_load16(ry2, by2 + 32 * ks));
      const uint32_t so = v ? (uint32_t)n * 4 : BUF_OOB;
      const float d = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rden, (int)so, 0, 0));
      den = v ? d : 1.f;
      rmx = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rstab, (int)so, 0, 0));
    }
  };

  bf16x8 xf[4], yf[4], y2f[4];
  float den_c = 1.f, rmx_c = 0.f;
  fetch(n_lo + 32 * w, xf, yf, y2f, den_c, rmx_c);
  // The first tile's fragments are waited for HERE.  Otherwise the wait sits at the top of the loop, where the
  // compiler must cover both ways in: coming from above the fragment loads are the youngest operations (wait for
  // everything), on the back edge they are followed by the previous tile's stores -- and the merged wait made every
  // tile wait for the stores of the one before.
  wait_vmcnt<0>();
  for (int n0 = n_lo + 32 * w; n0 < n_hi; n0 += TRIP) {
    const int n = n0 + (l & 31);
    const bool valid = n < n_hi;
    bf16x8 nxf[4], nyf[4], ny2f[4];   // the next tile's fragments, fetched before this tile's stores
    float nden = 1.f, nrmx = 0.f;
    const float norm2 = frag_sumsq(xf);

    if (MODE == 0) {
      f32x16 X[NFT];
      float mx = -INFINITY;
#pragma unroll
      for (int t = 0; t < NFT; ++t) {
        X[t] = zero16();   // Omega_t (32x64) * q^T (64x32); block t of the image at +4096 t from the lane's offsets
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) X[t] = mfma32(*(const bf16x8*)(omg + 4096 * t + lo.row[ks]), xf[ks], X[t]);
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (feat_valid(t, r, half)) mx = fmaxf(mx, X[t][r]);
      }
      mx = xhalf_max(mx);
      const float off = (HALF_C2 * norm2 + CNORM * mx) * LOG2E;
      f32x16 Y[2] = {zero16(), zero16()};
      float dsum = 0.f;
#pragma unroll
      for (int t = 0; t < NFT; ++t) {
        f32x16 e = zero16();
        const int ng = (t == NFT - 1) ? LAST_GROUPS : 4, nsk = (t == NFT - 1) ? LAST_KSTEPS : 2;   // compile-time (unrolled)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (g >= ng) continue;
          const f32x4 z4 = *(const f32x4*)(zv + 32 * t + 8 * g + 4 * half);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = 4 * g + i;
            float v = exp2_fast(X[t][r] * (CNORM * LOG2E) - off);
            v = feat_valid(t, r, half) ? v : 0.f;
            e[r] = v;
            dsum += v * z4[i];
          }
        }
#pragma unroll
        for (int sk = 0; sk < 2; ++sk) {
          if (sk >= nsk) continue;
          const bf16x8 ef = acc_to_frag(e, sk);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt)
            Y[dt] = mfma32(tr_frag_at(mim + 4096 * t, lo.tr[sk][dt]), ef, Y[dt]);
        }
      }
      const float den = xhalf_sum(dsum) + KEPS * zsum;
      const float inv = 1.f / den;
      // the context image is centred (its feature-sum is zero, so the eps term vanishes):
      // Y / den = o - obar
      fetch(n0 + TRIP, nxf, nyf, ny2f, nden, nrmx);
      stage_rows(stage, Y, inv, nullptr, 0.f);
      __builtin_amdgcn_wave_barrier();
      if (has_out1) store_rows(stage, ro1, a.out1.sn, nullptr, 0.f, n0, n_hi);
      store_rows(stage, ro0, a.out0.sn, nullptr, 0.f, n0, n_hi, nullptr, sm);
      __builtin_amdgcn_wave_barrier();
      if (a.rowmax_out && valid && l < 32) { a.rowmax_out[srow + n] = mx; a.den_out[srow + n] = den; }
    }

    if (MODE == 1) {
      bf16x8 (&dof)[4] = yf;            // dO fragments (scaled by 1/den below), centred-output fragments
      const bf16x8 (&of)[4] = y2f;
      float delta = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) delta += (float)dof[ks][j] * (float)of[ks][j];
      delta = xhalf_sum(delta);
      const float den = den_c;
      const float rmx = rmx_c;
      const float inv = 1.f / den;
      const float dden = -delta * inv;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) dof[ks][j] = (bf16)((float)dof[ks][j] * inv);
      const float off = (HALF_C2 * norm2 + CNORM * rmx) * LOG2E;
      f32x16 Y[2] = {zero16(), zero16()};
      float rs = 0.f;
      int jstar = 1 << 20;
      auto tile = [&](int t, auto last_c) {
        constexpr bool LAST = decltype(last_c)::value;
        const char* ob_ = omg + t * 4096;
        const char* mb_ = mim + t * 4096;
        f32x16 X = zero16(), G = zero16();
        // X[m,token] = Omega_t q^T ; G[m,token] = C'_t dnum^T (centred context)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          X = mfma32(*(const bf16x8*)(ob_ + lo.row[ks]), xf[ks], X);
          G = mfma32(*(const bf16x8*)(mb_ + lo.row[ks]), dof[ks], G);
        }
        f32x16 dA = zero16();
        int jr = 64;   // row (without the 4*half part) of the first arg-max element of this tile, if any
#pragma unroll
        for (int g = (LAST ? LAST_GROUPS : 4) - 1; g >= 0; --g) {
          const f32x4 z4 = *(const f32x4*)(zv + 32 * t + 8 * g + 4 * half);
#pragma unroll
          for (int i = 3; i >= 0; --i) {
            const int r = 4 * g + i;
            float e = exp2_fast(X[r] * (CNORM * LOG2E) - off);
            const bool fv = !LAST || acc_row(r, half) < NF - 32 * (NFT - 1);
            if (LAST) e = fv ? e : 0.f;
            const float v = e * (G[r] + z4[i] * dden);
            rs += v;
            dA[r] = v;
            // where the row max sits (the stabiliser's gradient goes there, see below).  Rows are visited
            // downwards, so the last assignment to jr is the smallest matching row.
            const bool ismax = fv && X[r] == rmx;
            jr = ismax ? acc_row(r, 0) : jr;
          }
        }
        jstar = (jr < 64 && jstar >= (1 << 20)) ? 32 * t + jr + 4 * half : jstar;
#pragma unroll
        for (int sk = 0; sk < (LAST ? LAST_KSTEPS : 2); ++sk) {
          const bf16x8 af = acc_to_frag(dA, sk);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) Y[dt] = mfma32(tr_frag_at(ob_, lo.tr[sk][dt]), af, Y[dt]);
        }
      };
#pragma unroll 2
      for (int t = 0; t < NFT - 1; ++t) tile(t, std::false_type{});
      tile(NFT - 1, std::true_type{});
      const float rs_all = xhalf_sum(rs);
      const float ddiag = -rs_all;
      jstar = min(jstar, __shfl_xor(jstar, 32));
      // The stabiliser routes -sum(dA) to the arg-max feature j*: the net coefficient of Omega[j*] is
      // dA[j*] - sum(dA) = -(sum of the other dA), a difference that must not be left to bf16.  The matrix product
      // above carried bf16(dA[j*]) Omega[j*]; dA[j*] itself is recomputed here in f32 from its factors (the
      // exponential at the row max, the context row j* against this token's dO/den, z[j*]), and the write-back adds
      // (dA[j*] - bf16(dA[j*]) - sum(dA)) Omega[j*] in f32.
      float stabc = 0.f;
      if (jstar < NF) {
        float gs = 0.f;
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        struct Pairs { bf16x2_t p[4]; };   // a fragment's 16 bytes as four packed pairs (no element extraction)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const Pairs cr = __builtin_bit_cast(Pairs, *(const bf16x8*)(mim + img_off(jstar, 16 * ks + 8 * half)));
          const Pairs dd = __builtin_bit_cast(Pairs, dof[ks]);
#pragma unroll
          for (int j = 0; j < 4; ++j) gs = __builtin_amdgcn_fdot2_f32_bf16(cr.p[j], dd.p[j], gs, false);
        }
        gs = xhalf_sum(gs);
        const float da = exp2_fast(rmx * (CNORM * LOG2E) - off) * (gs + zv[jstar] * dden);
        stabc = (da - (float)(bf16)da) - rs_all;
      }
      __builtin_amdgcn_sched_barrier(0);   // the next tile's loads stay below: hoisted above, they cost spills
      bf16x8 xr[4];
      load_x_rows(xr, rx, a.x.sn, n0, n_hi);
      fetch(n0 + TRIP, nxf, nyf, ny2f, nden, nrmx);
      auto half_tile = [&](auto pass_c) {   // tokens 16 pass .. 16 pass + 15 of the tile through the 16-row stage
        constexpr int PASS = decltype(pass_c)::value;
        stage_rows_half(stage, Y, CNORM, PASS);
        if (l < 32 && (l >> 4) == PASS) {
          stage[(l & 15) * STAGE_PITCH + 64] = ddiag;
          // stabiliser gradient: rowmax = S[n, j*]  =>  dq += -c * (sum(dA) - dA[j*]) * Omega[j*]
          stage[(l & 15) * STAGE_PITCH + 65] = (jstar < NF) ? CNORM * stabc : 0.f;
          ((int*)stage)[(l & 15) * STAGE_PITCH + 66] = (jstar < NF) ? jstar : 0;
        }
        __builtin_amdgcn_wave_barrier();
        store_rows<2 * PASS, 2>(stage, ro0, a.out0.sn, xr, 2.f * HALF_C2, n0, n_hi, omg);
        __builtin_amdgcn_wave_barrier();
      };
      half_tile(std::integral_constant<int, 0>{});
      half_tile(std::integral_constant<int, 1>{});
      if (valid && l < 32) a.dden_out[srow + n] = dden;
    }

    if (MODE == 2) {
      const bf16x8 (&vf)[4] = yf;
      const float off = (HALF_C2 * norm2 + CNORM * gstab) * LOG2E;
      f32x16 Yk[2] = {zero16(), zero16()};
      f32x16 Yv[2] = {zero16(), zero16()};
      float rs = 0.f;
      auto tile = [&](int t, auto last_c) {
        constexpr bool LAST = decltype(last_c)::value;
        const char* ob_ = omg + t * 4096;
        const char* mb_ = mim + t * 4096;
        f32x16 X = zero16(), G = zero16();
        // X[m,token] = Omega_t k^T ; G[m,token] = dC_t v^T
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          X = mfma32(*(const bf16x8*)(ob_ + lo.row[ks]), xf[ks], X);
          G = mfma32(*(const bf16x8*)(mb_ + lo.row[ks]), vf[ks], G);
        }
        f32x16 E = zero16(), dA = zero16();
#pragma unroll
        for (int g = 0; g < (LAST ? LAST_GROUPS : 4); ++g) {
          const f32x4 z4 = *(const f32x4*)(zv + 32 * t + 8 * g + 4 * half);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = 4 * g + i;
            float e = exp2_fast(X[r] * (CNORM * LOG2E) - off);
            if (LAST) e = (acc_row(r, half) < NF - 32 * (NFT - 1)) ? e : 0.f;
            E[r] = e;
            const float v = e * (G[r] + z4[i]);
            dA[r] = v;
            rs += v;
          }
        }
#pragma unroll
        for (int sk = 0; sk < (LAST ? LAST_KSTEPS : 2); ++sk) {
          const bf16x8 af = acc_to_frag(dA, sk);
          const bf16x8 ef = acc_to_frag(E, sk);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            Yk[dt] = mfma32(tr_frag_at(ob_, lo.tr[sk][dt]), af, Yk[dt]);
            Yv[dt] = mfma32(tr_frag_at(mb_, lo.tr[sk][dt]), ef, Yv[dt]);
          }
        }
      };
#pragma unroll 2
      for (int t = 0; t < NFT - 1; ++t) tile(t, std::false_type{});
      tile(NFT - 1, std::true_type{});
      const float ddiag = -xhalf_sum(rs);
      if (valid && l < 32) ksum -= ddiag;
      bf16x8 xr[4];
      load_x_rows(xr, rx, a.x.sn, n0, n_hi);
      fetch(n0 + TRIP, nxf, nyf, ny2f, nden, nrmx);
      // dk first (it consumes the x rows), then dv = E dC + eps * colsum(dC)
      stage_rows(stage, Yk, CNORM, nullptr, 0.f);
      if (l < 32) stage[(l & 31) * STAGE_PITCH + 64] = ddiag;
      __builtin_amdgcn_wave_barrier();
      store_rows(stage, ro0, a.out0.sn, xr, 2.f * HALF_C2, n0, n_hi);
      __builtin_amdgcn_wave_barrier();
      stage_rows(stage, Yv, 1.f, sm, KEPS);
      __builtin_amdgcn_wave_barrier();
      store_rows(stage, ro1, a.out1.sn, nullptr, 0.f, n0, n_hi);
      __builtin_amdgcn_wave_barrier();
    }
    // rotate: the prefetched fragments become the current ones
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { xf[ks] = nxf[ks]; if (MODE != 0) yf[ks] = nyf[ks]; if (MODE == 1) y2f[ks] = ny2f[ks]; }
    den_c = nden; rmx_c = nrmx;
  }
  if (MODE == 2) {
    // fixed-order sum of the 8 wave totals -> one partial per workgroup
    ksum = wave_sum(ksum);
    __syncthreads();
    if (l == 0) sm[w] = ksum;
    __syncthreads();
    if (tid == 0) {
      float s = 0.f;
      for (int i = 0; i < 8; ++i) s += sm[i];
      a.ksum_part[blockIdx.y * gridDim.x + blockIdx.x] = s;
    }
  }
}

// ------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------
hipError_t favor_prep_omega(const float* omega, void* out, hipStream_t s) {
  Traced tr(K_PREP_OMEGA, 0, s);
  hipLaunchKernelGGL(favor_prep_omega_kernel, dim3((NFP * DH + 255) / 256), dim3(256), 0, s, omega, (bf16*)out);
  return hipGetLastError();
}

static int round_up(int v, int m) { return (v + m - 1) / m * m; }

hipError_t favor_kfix(const float* part_val, const long long* part_key, int stab_per_bh, const float* ksum_part,
                      int sum_per_bh, int BH, int group_bh, const void* omega_b, TensorBHND_out dk, int H, int N,
                      hipStream_t s) {
  Traced tr(K_KFIX, 0, s);
  hipLaunchKernelGGL(favor_kfix_kernel, dim3((BH + group_bh - 1) / group_bh), dim3(256), 0, s, part_val, part_key,
                     stab_per_bh, ksum_part, sum_per_bh, BH, group_bh, (const bf16*)omega_b, dk, H, N);
  return hipGetLastError();
}

hipError_t favor_ctx(int mode, CtxArgs a, int B, int nchunks, hipStream_t s) {
  a.chunk_len = round_up((a.N + nchunks - 1) / nchunks, CTX_TILE);
  const size_t lds = 2 * CTX_BUF + 80 * sizeof(float);
  // algorithmic GEMM-term FLOPs per (token, head): mode 0: scores 2*64*266 + K'^T V 2*266*64;
  // mode 1: q'^T dnum 2*266*64 + dz 2*266 (the recomputed scores are not counted)
  const double th = (double)B * a.H * a.N;
  // bytes per (token, head): mode 0 reads k, v; mode 1 reads q, dout, the centred output and 3 f32 row scalars
  Traced tr(mode == 0 ? K_CTX_FWD : K_CTX_BWD, th * (mode == 0 ? 68096.0 : 34580.0), s, th * (mode == 0 ? 256.0 : 396.0));
  if (mode == 0)
    hipLaunchKernelGGL((favor_ctx_kernel<0>), dim3(nchunks, B * a.H), dim3(768), lds, s, a);
  else
    hipLaunchKernelGGL((favor_ctx_kernel<1>), dim3(nchunks, B * a.H), dim3(768), lds, s, a);
  return hipGetLastError();
}

// scratch: BH * (288*64 + 288 + 9*80) floats placed by the caller after the chunk partials
hipError_t favor_ctx_finalize(const float* part, int nchunks, int BH, int mode, const float* stab, int stab_per_bh,
                              int group_bh, const bf16* omega_b, const float* sums_fwd, void* Mb, float* zf, float* sums, float* scratch, hipStream_t s) {
  float* M32 = scratch;
  float* zz = M32 + (long)BH * NFP * DH;
  float* colpart = zz + (long)BH * NFP;
  Traced tr(K_CTX_FINALIZE, 0, s);
  if (mode == 1) {   // backward: one launch (the consumer sums the column partials, favor_fin_colpart)
    if (nchunks == 8)
      hipLaunchKernelGGL((favor_ctx_reduce_kernel<8, true>), dim3(BH, NFT), dim3(256), 0, s, part, nchunks, stab, stab_per_bh, group_bh,
                         omega_b, sums, M32, zz, colpart, sums_fwd, (bf16*)Mb, zf);
    else
      hipLaunchKernelGGL((favor_ctx_reduce_kernel<0, true>), dim3(BH, NFT), dim3(256), 0, s, part, nchunks, stab, stab_per_bh, group_bh,
                         omega_b, sums, M32, zz, colpart, sums_fwd, (bf16*)Mb, zf);
    return hipGetLastError();
  }
  if (nchunks == 8)
    hipLaunchKernelGGL((favor_ctx_reduce_kernel<8, false>), dim3(BH, NFT), dim3(256), 0, s, part, nchunks, stab, stab_per_bh, group_bh,
                       omega_b, sums, M32, zz, colpart, sums_fwd, (bf16*)Mb, zf);
  else
    hipLaunchKernelGGL((favor_ctx_reduce_kernel<0, false>), dim3(BH, NFT), dim3(256), 0, s, part, nchunks, stab, stab_per_bh, group_bh,
                       omega_b, sums, M32, zz, colpart, sums_fwd, (bf16*)Mb, zf);
  hipLaunchKernelGGL(favor_ctx_finalize_kernel, dim3(BH, NFT), dim3(256), 0, s, M32, zz, colpart, mode, sums_fwd,
                     (bf16*)Mb, zf, sums);
  return hipGetLastError();
}

float* favor_fin_colpart(float* scratch, int BH) { return scratch + (long)BH * NFP * DH + (long)BH * NFP; }

size_t favor_apply_lds() { return APPLY_LDS; }

hipError_t favor_apply(int mode, ApplyArgs a, int B, int nchunks, hipStream_t s) {
  a.chunk_len = round_up((a.N + nchunks - 1) / nchunks, 32);
  const size_t lds = mode == 1 ? (size_t)APPLY_LDS_DQ : favor_apply_lds();
  static std::atomic<unsigned long long> ready[3];
  const void* kfn = mode == 0 ? (const void*)favor_apply_kernel<0>
                              : (mode == 1 ? (const void*)favor_apply_kernel<1> : (const void*)favor_apply_kernel<2>);
  if (hipError_t e = ensure_dynamic_lds(kfn, lds, ready[mode]); e != hipSuccess) return e;
  dim3 grid(nchunks, B * a.H);
  // algorithmic FLOPs per (token, head): mode 0: scores + q'C + q'.z = 68,628; mode 1: dnum C^T + dden z +
  // dS Omega = 68,628; mode 2: v dC^T + k' dC + dS Omega = 102,144 (recomputed scores not counted)
  const double th = (double)B * a.H * a.N;
  Traced tr(mode == 0 ? K_APPLY_FWD : (mode == 1 ? K_APPLY_BWD_Q : K_APPLY_BWD_K),
            th * (mode == 2 ? 102144.0 : 68628.0), s,
            // bytes per (token, head): mode 0 q in, out + centred out, 2 row scalars; mode 1 q, dout, centred out in,
            // dq out, 3 row scalars; mode 2 k, v in, dk, dv out
            th * (mode == 0 ? (a.out1.p ? 392.0 : 256.0) : (mode == 1 ? 524.0 : 512.0)));
  if (mode == 0) hipLaunchKernelGGL((favor_apply_kernel<0>), grid, dim3(512), lds, s, a);
  else if (mode == 1) hipLaunchKernelGGL((favor_apply_kernel<1>), grid, dim3(768), lds, s, a);
  else hipLaunchKernelGGL((favor_apply_kernel<2>), grid, dim3(512), lds, s, a);
  return hipGetLastError();
}

}  // namespace scm
