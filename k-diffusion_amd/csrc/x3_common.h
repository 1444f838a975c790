// Shared pieces of the round-3 fp32-parity ("split3") kernels: gemm_x3.hip (norm -> projection), ffn_x3.hip (fused feed-forward block).
// Packed weight stage, counted waits, hi / lo splitting, compile-time loops and the named-AccVGPR helpers.
#pragma once
#include "bf16_common.h"
#include <utility>

namespace kd {
namespace x3 {

using b16::bf16x8;
using b16::u32x2;
using b16::u32x4;
using b16::pack_bf16;

constexpr int STG = 16384, IMG = 8192;      // one ring stage: [hi image | lo image] of [128 W rows][32 k] bf16

// byte offset of (row, 16-byte chunk c in 0..3) inside a [128][32] bf16 image (kd_pack_weight_bf16x3's swizzle)
__device__ __forceinline__ int swz64(int row, int c) { return row * 64 + ((c ^ ((row >> 2) & 3)) << 4); }

#define KD_BARRIER() asm volatile("s_barrier" ::: "memory")

// ---- per-workgroup time line (benchmarks/wg_timeline.py) ---------------------------------------------------------------------------------------
// kd_prof_clock_buffer hands the kernels a buffer of 16 stamps of ONE workgroup.  When the caller sets entry 15 to the magic value below the
// buffer holds 32 + 3 * grid entries and EVERY workgroup's first thread writes its entry / exit time (s_memrealtime: 100 MHz ticks, one clock
// for the whole chip) to entries 32 + 3 * blockIdx.x, + 1: launch ramp, rounds, tail and the spread of the workgroups' durations become
// visible.  Outside every loop; nothing is read or written when no buffer is set.
constexpr unsigned long long KD_WG_STAMP_MAGIC = 0x4b44ull;
struct WgStamp { unsigned long long* slot; };
__device__ __forceinline__ WgStamp wg_stamp_begin(unsigned long long* clk) {
  WgStamp w{nullptr};
  if (clk && threadIdx.x == 0 && clk[15] == KD_WG_STAMP_MAGIC) {
    w.slot = clk + 32 + 3 * blockIdx.x;
    w.slot[0] = __builtin_amdgcn_s_memrealtime();
  }
  return w;
}
__device__ __forceinline__ void wg_stamp_end(const WgStamp& w) {
  if (w.slot) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this thread's stores are out
    w.slot[1] = __builtin_amdgcn_s_memrealtime();
  }
}

// s_waitcnt vmcnt(n) for a run-time n (6-bit counter: anything above 60 waits for 60 outstanding, which is only more conservative)
__device__ __forceinline__ void wait_vm(int n) {
  switch (n < 60 ? n : 60) {
#define KD_C(v) case v: asm volatile("s_waitcnt vmcnt(" #v ")" ::: "memory"); break;
    KD_C(0) KD_C(1) KD_C(2) KD_C(3) KD_C(4) KD_C(5) KD_C(6) KD_C(7) KD_C(8) KD_C(9) KD_C(10) KD_C(11) KD_C(12) KD_C(13) KD_C(14) KD_C(15)
    KD_C(16) KD_C(17) KD_C(18) KD_C(19) KD_C(20) KD_C(21) KD_C(22) KD_C(23) KD_C(24) KD_C(25) KD_C(26) KD_C(27) KD_C(28) KD_C(29) KD_C(30) KD_C(31)
    KD_C(32) KD_C(33) KD_C(34) KD_C(35) KD_C(36) KD_C(37) KD_C(38) KD_C(39) KD_C(40) KD_C(41) KD_C(42) KD_C(43) KD_C(44) KD_C(45) KD_C(46) KD_C(47)
    KD_C(48) KD_C(49) KD_C(50) KD_C(51) KD_C(52) KD_C(53) KD_C(54) KD_C(55) KD_C(56) KD_C(57) KD_C(58) KD_C(59) KD_C(60)
#undef KD_C
  }
}

// 8 consecutive fp32 (two float4) -> hi / lo bf16 fragments: hi = bf16_rne(x), lo = bf16_rne(x - hi)
__device__ __forceinline__ void split8(const f32x4 v0, const f32x4 v1, u32x4& hi, u32x4& lo) {
  hi = u32x4{pack_bf16(v0[0], v0[1]), pack_bf16(v0[2], v0[3]), pack_bf16(v1[0], v1[1]), pack_bf16(v1[2], v1[3])};
  lo = u32x4{pack_bf16(v0[0] - b16::bf_lo(hi[0]), v0[1] - b16::bf_hi(hi[0])), pack_bf16(v0[2] - b16::bf_lo(hi[1]), v0[3] - b16::bf_hi(hi[1])),
             pack_bf16(v1[0] - b16::bf_lo(hi[2]), v1[1] - b16::bf_hi(hi[2])), pack_bf16(v1[2] - b16::bf_lo(hi[3]), v1[3] - b16::bf_hi(hi[3]))};
}
// 4 consecutive fp32 -> the 16 bytes [hi: 4 x bf16][lo: 4 x bf16] (KdGemm.qkv_packed: the operand format of the split attention cores)
__device__ __forceinline__ f32x4 pack_split4(const f32x4 v) {
  const unsigned h0 = pack_bf16(v[0], v[1]), h1 = pack_bf16(v[2], v[3]);
  const unsigned l0 = pack_bf16(v[0] - b16::bf_lo(h0), v[1] - b16::bf_hi(h0)), l1 = pack_bf16(v[2] - b16::bf_lo(h1), v[3] - b16::bf_hi(h1));
  return f32x4{__uint_as_float(h0), __uint_as_float(h1), __uint_as_float(l0), __uint_as_float(l1)};
}

// (K = 512: a lane's row takes 256 registers (hi + lo) next to 64 accumulators and the weight fragments, and hipcc -- which will not park
// long-lived MFMA operands in the AccVGPR half of the file, with builtins or with "a"-constrained asm -- spills ~150 of them to scratch, whose
// reloads also sit in the vmcnt queue the ring's counted waits assume to be theirs.  That width uses the named-AccVGPR path below.)
__device__ __forceinline__ void mfma_a(f32x16& acc, const bf16x8 w, const bf16x8 a) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, a, acc, 0, 0, 0);
}

template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f)); }

// ---- K = 512: the activation fragments live in NAMED AccVGPRs ------------------------------------------------------------------------------
// A lane's row takes 256 registers there (32 chunks x (hi + lo) x 4): exactly the AccVGPR half of the unified file, which leaves all 256
// ArchVGPRs to accumulators, weight fragments and the prologue / epilogue arithmetic.  hipcc will not make that assignment itself (see the
// note above mfma_a), so the kernel does it by hand: chunk c's hi fragment is a[8c .. 8c+3], its lo fragment a[8c+4 .. 8c+7], written with
// v_accvgpr_write_b32 in the prologue and named literally as the B operand of asm MFMAs.  Every statement that writes them lists all 256
// as clobbers (the compiler keeps nothing of its own there and the kernel descriptor allocates them); the build is audited for compiler
// v_accvgpr_* / scratch use (csrc/Makefile: check_x3_agpr).  Asm MFMAs get no hazard padding from the compiler: accumulate chains need
// none, the epilogue's first read of the accumulators is padded by hand.
#define KD_A16(b) "a" #b "0", "a" #b "1", "a" #b "2", "a" #b "3", "a" #b "4", "a" #b "5", "a" #b "6", "a" #b "7", "a" #b "8", "a" #b "9"
#define KD_AGPR_ALL                                                                                                                        \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", KD_A16(1), KD_A16(2), KD_A16(3), KD_A16(4), KD_A16(5), KD_A16(6), KD_A16(7),    \
      KD_A16(8), KD_A16(9), KD_A16(10), KD_A16(11), KD_A16(12), KD_A16(13), KD_A16(14), KD_A16(15), KD_A16(16), KD_A16(17), KD_A16(18),       \
      KD_A16(19), KD_A16(20), KD_A16(21), KD_A16(22), KD_A16(23), KD_A16(24), "a250", "a251", "a252", "a253", "a254", "a255"
template <int IDX>
__device__ __forceinline__ void areg_write4(const u32x4 v) {
  asm volatile("v_accvgpr_write_b32 a[%c4], %0\n\tv_accvgpr_write_b32 a[%c5], %1\n\tv_accvgpr_write_b32 a[%c6], %2\n\tv_accvgpr_write_b32 a[%c7], %3"
               :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "i"(IDX), "i"(IDX + 1), "i"(IDX + 2), "i"(IDX + 3) : KD_AGPR_ALL);
}
// kernels that run TWO waves per SIMD have the low 128 AccVGPRs (hipcc splits a 256-register budget 128 + 128 once AccVGPRs are used)
#define KD_A10(b) "a" #b "0", "a" #b "1", "a" #b "2", "a" #b "3", "a" #b "4", "a" #b "5", "a" #b "6", "a" #b "7", "a" #b "8", "a" #b "9"
#define KD_AGPR_LO128                                                                                                                     \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", KD_A10(1), KD_A10(2), KD_A10(3), KD_A10(4), KD_A10(5), KD_A10(6), KD_A10(7),   \
      KD_A10(8), KD_A10(9), KD_A10(10), KD_A10(11), "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127"
template <int IDX>
__device__ __forceinline__ void areg_write4_lo(const u32x4 v) {
  asm volatile("v_accvgpr_write_b32 a[%c4], %0\n\tv_accvgpr_write_b32 a[%c5], %1\n\tv_accvgpr_write_b32 a[%c6], %2\n\tv_accvgpr_write_b32 a[%c7], %3"
               :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "i"(IDX), "i"(IDX + 1), "i"(IDX + 2), "i"(IDX + 3) : KD_AGPR_LO128);
}
// all operands in ArchVGPRs, as asm: for kernels that name AccVGPRs themselves and cannot let the compiler choose the accumulator's file
__device__ __forceinline__ void mfma_vv(f32x16& acc, const bf16x8 w, const bf16x8 a) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(a));
}
// first MFMA of an accumulate chain: C = 0 (inline constant), the accumulator is written only
__device__ __forceinline__ void mfma_vv0(f32x16& acc, const bf16x8 w, const bf16x8 a) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(w), "v"(a));
}
template <int IDX>
__device__ __forceinline__ void mfma_ag0(f32x16& acc, const bf16x8 w) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], 0" : "=&v"(acc) : "v"(w), "i"(IDX), "i"(IDX + 3));
}
template <int IDX>
__device__ __forceinline__ void mfma_ag(f32x16& acc, const bf16x8 w) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(acc) : "v"(w), "i"(IDX), "i"(IDX + 3));
}

}  // namespace x3
}  // namespace kd
