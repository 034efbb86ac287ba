// Shared device helpers for the gfx950 (MI355X) kernels of the Performer RNA-encoder path.
// Wave = 64 lanes.  All matrix products use v_mfma_f32_32x32x16_bf16 (bf16 in, f32 accumulate).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace scm {

typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// model constants of the path (performer_pytorch.py:550,201; rna_encoder.py:29-37)
constexpr int DH = 64;        // dim_head
constexpr int NF = 266;       // nb_features = int(64 * ln 64)
constexpr int NFP = 288;      // padded to 9 MFMA tiles of 32
constexpr int NFT = NFP / 32; // 9
constexpr float KEPS = 1e-4f; // softmax_kernel eps
constexpr float LOG2E = 1.4426950408889634f;
// data_normalizer = 64^-1/4 ; c^2/2 = 1/16
constexpr float CNORM = 0.35355339059327373f;
constexpr float HALF_C2 = 0.0625f;

#define SCM_DEV __device__ __forceinline__

SCM_DEV int lane_id() { return threadIdx.x & 63; }

// D = A(32x16) * B(16x32) + C.  Lane l holds A[l&31][8*(l>>5)+j], B[8*(l>>5)+j][l&31], j=0..7.
// C/D: element r of lane l is row (r&3) + 8*(r>>2) + 4*(l>>5), column l&31.
SCM_DEV f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

SCM_DEV int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

SCM_DEV f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

SCM_DEV float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }

SCM_DEV float xhalf_sum(float v) { return v + __shfl_xor(v, 32); }
SCM_DEV float xhalf_max(float v) { return fmaxf(v, __shfl_xor(v, 32)); }

// Lane exchange inside a row of 16 lanes through the data-parallel-primitive path of the vector ALU (no LDS-pipe
// round trip): quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror.
template <int CTRL>
SCM_DEV float dpp_move(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// Sum over the 64 lanes, result in every lane: four exchanges inside each row of 16 lanes, then two across rows.
SCM_DEV float wave_sum(float v) {
  v += dpp_move<0xB1>(v);
  v += dpp_move<0x4E>(v);
  v += dpp_move<0x141>(v);
  v += dpp_move<0x140>(v);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
// The same sum (the same additions in the same pairing, so the same bits) delivered as a wave-uniform value: the row
// totals travel to lane 63 by the two wave-level broadcasts of the DPP path (row_bcast:15 into rows 1 and 3,
// row_bcast:31 into rows 2 and 3) and are read from there into a scalar register -- no LDS-pipe round trip.
template <int CTRL, int ROW_MASK>
SCM_DEV float dpp_move_rows(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
SCM_DEV float wave_sum_uniform(float v) {
  v += dpp_move<0xB1>(v);
  v += dpp_move<0x4E>(v);
  v += dpp_move<0x141>(v);
  v += dpp_move<0x140>(v);
  v += dpp_move_rows<0x142, 0xA>(v);
  v += dpp_move_rows<0x143, 0xC>(v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
SCM_DEV float wave_max(float v) {
  v = fmaxf(v, dpp_move<0xB1>(v));
  v = fmaxf(v, dpp_move<0x4E>(v));
  v = fmaxf(v, dpp_move<0x141>(v));
  v = fmaxf(v, dpp_move<0x140>(v));
  v = fmaxf(v, __shfl_xor(v, 16));
  v = fmaxf(v, __shfl_xor(v, 32));
  return v;
}

// Registers 8s..8s+7 of a 32x32 accumulator tile as the bf16 fragment of k-step s of a
// following MFMA that contracts over the tile's ROW index.  Element j of lane-half h is row
// 16s + 8(j>>2) + 4h + (j&3) of the tile: the other operand must use the same k order.
SCM_DEV bf16x8 acc_to_frag(const f32x16& x, int s) {
  bf16x8 f;
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = (bf16)x[8 * s + j];
  return f;
}

// ---- LDS images with 128-byte rows (64 bf16), 16-byte chunks XOR-swizzled -----------------
// chunk' = chunk ^ f(row), f(row) = (((row>>1)&1)<<2) | ((row>>2)&3).
// Row reads (ds_read_b128, lanes = 32 consecutive rows, same chunk) and transposed reads
// (ds_read_b64_tr_b16 of 4-row x 16-column blocks) are both bank-conflict free.
SCM_DEV int swz128(int row, int chunk) {
  return chunk ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3));
}
// byte offset of element (row, col) in a swizzled [rows][64] bf16 image
SCM_DEV int img_off(int row, int col) {
  return row * 128 + swz128(row, col >> 3) * 16 + (col & 7) * 2;
}

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

// transposed 4x16 block read: returns, for this lane's column, the 4 consecutive rows.
SCM_DEV bf16x4 lds_tr_read(const void* lds_ptr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(lds_ptr));
}

// Fragment (A operand rows = image columns, or B operand cols = image columns) whose k index
// runs over image ROWS, read from a swizzled 128-byte-row image by two transposed reads.
//   k order "natural":  element j <- row  kbase + 8*half + j
//   k order "acc":      element j <- row  kbase + 8*(j>>2) + 4*half + (j&3)   (pairs with acc_to_frag)
// col_base: first of the 32 image columns this fragment spans.
template <bool ACC_ORDER>
SCM_DEV bf16x8 img_tr_frag(const char* img, int kbase, int col_base) {
  const int l = lane_id();
  const int g = l >> 4, i = l & 15, half = g >> 1;
  const int col = col_base + 16 * (g & 1) + 4 * (i & 3);
  const int r0 = ACC_ORDER ? (kbase + 4 * half) : (kbase + 8 * half);
  const int r1 = ACC_ORDER ? (kbase + 8 + 4 * half) : (kbase + 8 * half + 4);
  const int q = i >> 2;
  bf16x4 lo = lds_tr_read(img + img_off(r0 + q, col));
  bf16x4 hi = lds_tr_read(img + img_off(r1 + q, col));
  bf16x8 f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}

// Fragment whose k index runs along image COLUMNS (contiguous): lane reads 16 bytes of row
// row_base + (l&31) at columns kbase + 8*half .. +7.
SCM_DEV bf16x8 img_row_frag(const char* img, int row_base, int kbase) {
  const int l = lane_id();
  const int row = row_base + (l & 31);
  return *(const bf16x8*)(img + img_off(row, kbase + 8 * (l >> 5)));
}

// ---- raw buffer access: the hardware bounds check replaces per-lane branches around global memory ops.
// Loads at a byte offset >= the resource's size return 0 and stores there are dropped, so a lane that is out
// of range simply passes BUF_OOB: the instruction stream stays straight-line and the compiler can keep many
// loads in flight with exact s_waitcnt counts.  Resources cover less than 2 GiB (32-bit offsets).
typedef __amdgpu_buffer_rsrc_t brsrc;
constexpr uint32_t BUF_OOB = 0x80000000u;   // stays out of range when a small immediate offset is added
SCM_DEV brsrc make_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
SCM_DEV u32x4 buf_load16(brsrc r, uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0); }
SCM_DEV void buf_store16(brsrc r, uint32_t off, u32x4 v) { __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)off, 0, 0); }
SCM_DEV bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
SCM_DEV f32x4 as_f32x4(u32x4 v) { return __builtin_bit_cast(f32x4, v); }
SCM_DEV u32x4 as_u32x4(bf16x8 v) { return __builtin_bit_cast(u32x4, v); }

// Asynchronous global -> LDS copy of 16 bytes per lane (buffer_load_dwordx4 ... lds): lane i's bytes land at
// LDS byte address lds_wave_base + 16 i (wave-uniform base, placed in M0), no VGPR staging; lanes whose offset is
// out of the resource's range write zeros.  Written as inline assembly on purpose: the compiler then neither
// counts these copies nor orders LDS reads against them, so several stages can stay in flight; the consumer
// waits with wait_vmcnt<N>() (copies complete in issue order) and then synchronises the workgroup.
SCM_DEV u32x4 dma_rsrc(const void* p, uint32_t bytes) {
  const uint64_t a = (uint64_t)p;
  return u32x4{(uint32_t)a, (uint32_t)(a >> 32) & 0xFFFFu, bytes, 0x00020000u};
}
SCM_DEV uint32_t lds_addr_of(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
// (M0, a reserved register, is named as clobbered on purpose: the diagnostic about that is silenced for this helper only)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
SCM_DEV void dma16_to_lds(u32x4 rsrc, uint32_t lds_wave_base, uint32_t voffset, uint32_t soffset) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_wave_base), "v"(voffset), "s"(rsrc), "s"(soffset) : "memory", "m0");
}
#pragma clang diagnostic pop
// s_waitcnt vmcnt(N) only (expcnt/lgkmcnt left alone): gfx9 encoding vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4]<<14
template <int N>
SCM_DEV void wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}

// erf-GELU pieces as odd polynomials, x * P(x^2), evaluated by Horner in t = 2 x^2 / xmax^2 - 1 on the clamped
// argument: no reciprocal and no exponential (both quarter rate), which the GELU epilogues were bound by.
//   Phi(x)   = 0.5 + x P8(t),  |x| <= 4   : |error| <= 3e-5 (f32 evaluation, clamping included)
//   gelu'(x) = Phi(x) + x phi(x) = 0.5 + x Q10(t),  |x| <= 4.5 : |error| <= 8e-5
// Both are far below the bf16 resolution of the tensors these values are rounded into (2^-9 relative).
SCM_DEV float gelu_cdf(float x) {
  const float xc = __builtin_amdgcn_fmed3f(x, -4.0f, 4.0f);
  const float t = fmaf(xc * xc, 0.125f, -1.f);
  float p = 1.356703606e-03f;
  p = fmaf(p, t, -3.878374350e-03f);
  p = fmaf(p, t, 6.272368044e-03f);
  p = fmaf(p, t, -1.171882770e-02f);
  p = fmaf(p, t, 2.242440755e-02f);
  p = fmaf(p, t, -3.706041785e-02f);
  p = fmaf(p, t, 5.595684806e-02f);
  p = fmaf(p, t, -8.430877183e-02f);
  p = fmaf(p, t, 1.759490917e-01f);
  return fmaf(xc, p, 0.5f);
}
SCM_DEV float gelu_erf(float x) { return x * gelu_cdf(x); }
SCM_DEV float gelu_erf_grad(float x) {
  const float xc = __builtin_amdgcn_fmed3f(x, -4.5f, 4.5f);
  const float t = fmaf(xc * xc, 2.f / 20.25f, -1.f);
  float p = 1.190257076e-02f;
  p = fmaf(p, t, -2.947445965e-02f);
  p = fmaf(p, t, 2.863586895e-02f);
  p = fmaf(p, t, -3.601999706e-02f);
  p = fmaf(p, t, 6.953686171e-02f);
  p = fmaf(p, t, -9.394002518e-02f);
  p = fmaf(p, t, 9.800084290e-02f);
  p = fmaf(p, t, -9.390361139e-02f);
  p = fmaf(p, t, 8.697274791e-02f);
  p = fmaf(p, t, -9.001392591e-02f);
  p = fmaf(p, t, 1.594318619e-01f);
  return fmaf(xc, p, 0.5f);
}

}  // namespace scm
