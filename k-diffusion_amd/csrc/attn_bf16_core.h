// Shared pieces of the bf16 attention cores (attn_bf16.hip) and of the rows-stationary block kernels built on the dense core (block_bf16.hip):
// operand images in LDS (swizzle, LDS-DMA helper), the V^T fragment reads, the softmax pieces and the output store.
#pragma once
#include "bf16_common.h"

namespace kd {
namespace b16 {

constexpr int DH = 64;

enum { MODE_GLOBAL = 0, MODE_WINDOW = 1, MODE_WINDOW4 = 2, MODE_WINDOW16 = 3 };
template <int MODE> struct WinLog2 { static constexpr int v = MODE == MODE_WINDOW ? 3 : (MODE == MODE_WINDOW4 ? 2 : 4); };

struct DArgs {
  const u16* qkv; u16* out;
  int batch, T, nh;          // T = tokens per sample
  int H, W, ws, shift;       // window modes
  int warm;                  // code warm-up workgroups (kd_common.h)
};

template <int MODE>
__device__ __forceinline__ int slot_token(const DArgs& a, int slot, int wi, int wj) {
  if (MODE == MODE_GLOBAL) return slot;
  constexpr int L = WinLog2<MODE>::v, WS = 1 << L;
  const int ai = slot >> L, bj = slot & (WS - 1);
  int i = wi * WS + ai - a.shift; if (i < 0) i += a.H;     // rolled[i] = orig[(i - shift) mod H]  (:274)
  int j = wj * WS + bj - a.shift; if (j < 0) j += a.W;
  return i * a.W + j;
}
template <int MODE>
__device__ __forceinline__ int slot_region(int slot, int wi, int wj, int shift) {      // make_shifted_window_masks (:285-316)
  constexpr int L = WinLog2<MODE>::v, WS = 1 << L;
  return ((wi == 0 && (slot >> L) < shift) ? 2 : 0) + ((wj == 0 && (slot & (WS - 1)) < shift) ? 1 : 0);
}

#define KD_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define KD_BARRIER() asm volatile("s_barrier" ::: "memory")

__device__ __forceinline__ void glds16(const void* src, void* dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

// 16-byte chunk swizzle of the K / V images (rows of 128 bytes): chunk q of row r sits at q ^ asw(r).  Bit 2 of the XOR word comes
// from r bit 1, so the four rows r .. r + 3 of a ds_read_b64_tr_b16 group land in four different 64-byte quarters of the 256-byte
// bank row (the GEMM images' (r >> 1) & 7 puts rows r, r + 2 into the same quarter: two-way conflicts on every V^T read), while
// 16 consecutive rows still take 16 different (half, slot) positions for the ds_read_b128 of the K fragments.
// asw(r + 8) = asw(r) ^ 2, asw(r + 16) = asw(r).
__device__ __forceinline__ int asw(int row) { return ((row & 2) << 1) | ((row >> 2) & 3); }

// V^T fragments of one k-step (8 k-slots = image rows key0 .. key0+3 and key0+8 .. key0+11; features 32 e + (lane & 31), e = 0, 1)
// through ds_read_b64_tr_b16.  `va` = vt_addr(row key0 + ((lane & 15) >> 2), lane): the e = 1 chunk is `^ 64`, the +8 row is
// `^ 32` and 1024 bytes further.
using s16x4 = short __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int vt_lane_row(int lane) { return (lane & 15) >> 2; }
__device__ __forceinline__ int vt_addr(int r0, int lane) {
  const int c = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);
  return r0 * 128 + ((c ^ asw(r0)) << 4) + (lane & 1) * 8;
}
__device__ __forceinline__ bf16x8 vt_read(const char* vimg, int a_lo) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(vimg + a_lo));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(vimg + (a_lo ^ 32) + 1024));
  const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
  return __builtin_bit_cast(bf16x8, u32x4{l2[0], l2[1], h2[0], h2[1]});
}
// O^T (two feature blocks) += V^T P^T for one k-step
__device__ __forceinline__ void pv_step(f32x16 (&O)[2], const char* vimg, int va, const bf16x8 pf) {
  O[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vt_read(vimg, va), pf, O[0], 0, 0, 0);
  O[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vt_read(vimg, va ^ 64), pf, O[1], 0, 0, 0);
}
// 8 probabilities (accumulator registers 8u .. 8u+7 of a score tile) -> B-operand fragment
__device__ __forceinline__ bf16x8 p_frag(const f32x16& S, int u) {
  return __builtin_bit_cast(bf16x8, u32x4{pack_bf16(S[8 * u], S[8 * u + 1]), pack_bf16(S[8 * u + 2], S[8 * u + 3]),
                                          pack_bf16(S[8 * u + 4], S[8 * u + 5]), pack_bf16(S[8 * u + 6], S[8 * u + 7])});
}
// v_max3_f32 / v_min3_f32 through the compiler's own pattern (NOT inline asm: the hazard recogniser does not look inside asm
// operands, and an asm VALU read of a just-written MFMA result misses its wait states -- seen as wrong scores on hardware)
__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
__device__ __forceinline__ float min3f(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }
// row maximum of a lane's (masked) scores over NT tiles, both half-waves
template <int NT>
__device__ __forceinline__ float score_max(const f32x16 (&S)[NT]) {
  float m = -INFINITY;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; i += 2) m = max3f(m, S[t][i], S[t][i + 1]);
  return fmaxf(m, __shfl_xor(m, 32, 64));
}
// S <- exp(S - m) in place (v_exp_f32 on a packed fma), returns this lane's partial row sum
template <int NT>
__device__ __forceinline__ float score_exp(f32x16 (&S)[NT], float m) {
  constexpr float LOG2E = 1.4426950408889634f;
  const f32x2 mb = {-m * LOG2E, -m * LOG2E};
  f32x2 l2 = {0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
      const f32x2 x = __builtin_elementwise_fma(f32x2{S[t][i], S[t][i + 1]}, f32x2{LOG2E, LOG2E}, mb);
      const f32x2 pv = {__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
      S[t][i] = pv.x;
      S[t][i + 1] = pv.y;
      l2 += pv;
    }
  return l2.x + l2.y;
}
__device__ __forceinline__ void store_o(u16* orow, const f32x16 (&O)[2], float inv, int lh, bool ok) {
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = O[e][r] * inv;
    store_block_bf16(orow + 32 * e, v, lh, ok);
  }
}

}  // namespace b16
}  // namespace kd
