// Shared pieces of the bf16 arithmetic mode (KD_PREC_BF16) of libkdiff_hip.so, gfx950 only.
//
// bf16 mode = what the reference runs under torch.autocast(bfloat16) (k_diffusion/models/image_transformer_v2.py:98-103:
// fp32 statistics, `.to(x.dtype)`; flash path :376-384 needs half types): activations travel through HBM as bf16
// (residual stream, qkv, attention output, FF hidden), every product is ONE v_mfma_f32_32x32x16_bf16 with fp32
// accumulation, norm statistics / softmax / GELU / RoPE are fp32 in registers, the image and the solver state stay fp32.
//
// Packed weight image ("kd_pack_weight_bf16"): blocks of [128 tile rows][64 k] bf16 = 16 KiB, ordered
// [n-tile][k-step], rows 128 bytes, the 16-byte chunk index q (8 k each) stored at q ^ ((row >> 1) & 7):
// a ds_read_b128 of an MFMA operand fragment (32 rows x one chunk; lane groups of 16 rows) then touches every 16-byte
// slot of the 256-byte bank row exactly once.  A block is the unit every bf16 GEMM kernel moves with global_load_lds
// (lane-linear copy: the image in HBM IS the LDS image).  GEGLU tiles interleave 32 value rows with their 32 gate rows.
//
// Operand roles ("swapped" product): D = W_frag (A operand, rows = output features n) x act_frag (B operand, columns =
// activation rows m), so a lane owns ONE activation row m = lane & 31 and, per 32-feature block, the 16 features
// n = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).  Everything row-wise in the epilogues (RMS-norm row factor, cosine-sim norm over a
// head vector, RoPE pairs (d, d + 16), GEGLU value/gate) is then in-lane arithmetic plus one cross-half exchange; stores are
// 16 bytes per lane (8 consecutive features) after a v_permlane32_swap pairing of the two half-waves.
#pragma once
#include "kd_common.h"

namespace kd {
namespace b16 {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u16 = unsigned short;

constexpr int WBLK = 16384;     // bytes of one packed block: [128 rows][64 k] bf16
constexpr int WROWS = 128, WKS = 64;

// byte offset of (row r, 16-byte chunk q in 0..7) inside a [rows][64] bf16 image with 128-byte rows
__device__ __host__ __forceinline__ int swz128(int r, int q) { return r * 128 + ((q ^ ((r >> 1) & 7)) << 4); }

// W row that feeds tile row r of n-tile nt (GEGLU: 64 outputs per tile, 32 value rows then their 32 gate rows, twice)
__device__ __host__ __forceinline__ int w_row_of_tile(int nt, int r, int N, bool geglu) {
  if (geglu) {
    const int n = nt * 64 + (r >> 6) * 32 + (r & 31);
    return (n < N) ? (((r >> 5) & 1) ? N + n : n) : -1;
  }
  return (nt * 128 + r < N) ? nt * 128 + r : -1;
}

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {   // v_cvt_pk_bf16_f32, round to nearest even; a -> low half
  bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }

// exchange between the half-waves: afterwards lanes 0-31 hold (x_own, x_of_lane+32) in (x, y) and lanes 32-63 hold
// (y_of_lane-32, y_own).  An involution.
__device__ __forceinline__ void half_swap(unsigned& x, unsigned& y) {
  const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
  x = r[0];
  y = r[1];
}

// One 32-feature block of a lane's row, fp32 in MFMA C-layout order (v[r], r < 16) -> bf16, paired with the other half-wave
// into 2 x 16-byte stores.  `crow` = &C[row][first feature of the block]; lh = lane >> 5.
// `nvalid` (multiple of 8): features of the block that exist (ragged last block of a row): pieces past it are not stored.
__device__ __forceinline__ void store_block_bf16(u16* crow, const float (&v)[16], int lh, bool ok, int nvalid = 32) {
  unsigned p[4][2];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    p[g][0] = pack_bf16(v[4 * g], v[4 * g + 1]);
    p[g][1] = pack_bf16(v[4 * g + 2], v[4 * g + 3]);
  }
#pragma unroll
  for (int gp = 0; gp < 4; gp += 2) {
    half_swap(p[gp][0], p[gp + 1][0]);
    half_swap(p[gp][1], p[gp + 1][1]);
    // lanes 0-31: features 8gp .. 8gp+7; lanes 32-63: features 8(gp+1) .. +7
    if (ok && 8 * (gp + lh) < nvalid) st16(crow + 8 * (gp + lh), u32x4{p[gp][0], p[gp][1], p[gp + 1][0], p[gp + 1][1]});
  }
}

// the same block of a bf16 row-major operand (residual / skip) brought INTO the C-layout as fp32: r[i] pairs with v[i]
__device__ __forceinline__ void load_block_bf16(const u16* rrow, float (&r)[16], int lh, int nvalid = 32) {
#pragma unroll
  for (int gp = 0; gp < 4; gp += 2) {
    const u32x4 q = *reinterpret_cast<const u32x4*>(rrow + (8 * (gp + lh) < nvalid ? 8 * (gp + lh) : 0));   // pieces past the row end: re-read piece 0 (unused)
    unsigned q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    half_swap(q0, q2);
    half_swap(q1, q3);
    r[4 * gp + 0] = bf_lo(q0); r[4 * gp + 1] = bf_hi(q0); r[4 * gp + 2] = bf_lo(q1); r[4 * gp + 3] = bf_hi(q1);
    r[4 * gp + 4] = bf_lo(q2); r[4 * gp + 5] = bf_hi(q2); r[4 * gp + 6] = bf_lo(q3); r[4 * gp + 7] = bf_hi(q3);
  }
}
// raw 16-byte pieces of that block (requested early, converted later with block_from_raw)
__device__ __forceinline__ void load_block_raw(const u16* rrow, u32x4 (&q)[2], int lh) {
  q[0] = *reinterpret_cast<const u32x4*>(rrow + 8 * lh);
  q[1] = *reinterpret_cast<const u32x4*>(rrow + 8 * (2 + lh));
}
__device__ __forceinline__ void block_from_raw(u32x4 (&qq)[2], float (&r)[16]) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    unsigned q0 = qq[h][0], q1 = qq[h][1], q2 = qq[h][2], q3 = qq[h][3];
    half_swap(q0, q2);
    half_swap(q1, q3);
    const int gp = 2 * h;
    r[4 * gp + 0] = bf_lo(q0); r[4 * gp + 1] = bf_hi(q0); r[4 * gp + 2] = bf_lo(q1); r[4 * gp + 3] = bf_hi(q1);
    r[4 * gp + 4] = bf_lo(q2); r[4 * gp + 5] = bf_hi(q2); r[4 * gp + 6] = bf_lo(q3); r[4 * gp + 7] = bf_hi(q3);
  }
}

// lane half lh picks 4 of a head's 8 RoPE frequencies held in SCALAR registers (s_load).  Written as `lh ? f[4 + u] : f[u]` hipcc turns the
// choice into an indexed extract from the 8-vector: seven v_cmp / v_cndmask pairs (plus their hazard s_nops) per element, 56 pairs per
// head vector in every qkv epilogue.  As a bit select it is one v_bfi_b32 per element.
__device__ __forceinline__ float pick_half(float f_lo, float f_hi, unsigned hi_mask /* 0 or ~0u */) {
  return __uint_as_float((__float_as_uint(f_hi) & hi_mask) | (__float_as_uint(f_lo) & ~hi_mask));
}

// ---- q / k preparation of one 64-dim head vector held as two C-layout blocks (dims 0..31 in a0, 32..63 in a1) -----------
// scale_for_cosine_sim (image_transformer_v2.py:106-114) + _apply_rotary_emb_inplace (:187-199) on the RAW accumulators of the
// row (true value = acc * rs, rs = the RMS-norm row factor): q <- rope(acc * g), g = rs * sqrt(scale_h) * rsqrt(rs^2 * sum acc^2 + eps).
// Rotary pairs (d, d + 16), d < 16, are registers (r, r + 8) of the SAME lane in a0.  Angles come from the token's axial
// position (py, px) and this lane's four frequencies fr[u] = freqs[head][4 lh + u] / (2 pi) (v_sin / v_cos take revolutions):
// dims d < 8 turn with y, 8 <= d < 16 with x (axial_rope.py / image_transformer_v2.py:234-248).
__device__ __forceinline__ void qk_prep_blocks(f32x16& a0, f32x16& a1, float rs, float sqrt_scale, float eps, float py, float px,
                                               const float (&fr)[4]) {
  // Every product / sum below is written as the instruction it must become (contraction off, explicit fmaf): this function is inlined
  // into several kernels -- the qkv projections and the fused attention block (block_bf16.hip), whose results are tested BIT-IDENTICAL
  // against each other -- and left to itself the compiler fuses `x1 * c - x2 * s` one way in one kernel and the other way in the next.
#pragma clang fp contract(off)
  float ss = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) ss = fmaf(a0[r], a0[r], fmaf(a1[r], a1[r], ss));
  ss += __shfl_xor(ss, 32, 64);
  const float g = rs * sqrt_scale * rsqrtf(fmaf(rs * rs, ss, eps));
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const float rev = ((r >> 2) ? px : py) * fr[r & 3];
    const float c = __builtin_amdgcn_cosf(rev), s = __builtin_amdgcn_sinf(rev);
    const float x1 = a0[r] * g, x2 = a0[r + 8] * g;
    const float t1 = x2 * s, t2 = x1 * s;
    a0[r] = fmaf(x1, c, -t1);
    a0[r + 8] = fmaf(x2, c, t2);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) a1[r] *= g;
}

}  // namespace b16
}  // namespace kd
