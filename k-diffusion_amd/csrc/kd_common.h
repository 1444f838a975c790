// Shared host/device helpers for libkdiff_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/kdiff_hip.h"

namespace kd {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int WAVE = 64;

// 16-byte store of a kernel's RESULT rows.  Round 6 tried the write-through form here (`global_store_dwordx4 ... sc1`: the bytes leave the
// XCD's L2 while the kernel runs, so that the write-back at the kernel boundary -- (dirty bytes) / ~6 TB/s before the next dependent launch
// starts, MI355X_MICROARCH.md "boundary" -- finds nothing to do).  Same box, interleaved with the plain build: fp32-parity 212.1 / 213.1 ->
// 208.7 / 210.7 images/s, bf16 429.0 / 430.0 -> 367.8 / 366.9 (-14.5 %): a write-through store DROPS the line from the L2, and the next launch
// of a layer re-reads most of what its predecessor wrote on the same XCD (the panel / tile placements are XCD-aware for exactly that reason).
// The boundary's write-back is the cheaper side of that trade.  Plain stores it is (profiles/r06_wt_stores.md).
template <class V>
__device__ __forceinline__ void st16(void* p, const V& v) {
  static_assert(sizeof(V) == 16, "st16: 16-byte vectors");
  *reinterpret_cast<V*>(p) = v;
}

// the public descriptor plus what only the library sets (kept out of the ABI)
struct GemmP : KdGemm {
  int warm;         // code warm-up workgroups (code_warm_begin below; option "code_warm")
  int debug;        // benchmarks/ only (kd_set_option("gemm_debug")): 1 no C stores, 2 no MFMA, 8 GEGLU without erf, 32 conservative store wait
  int scale_tab;    // norm scales of a tile's sample staged in LDS
};

// ---- error reporting (thread-local message, C ABI returns a code) -------------------------------
char* err_buf();
int fail(int code, const char* fmt, ...);
// kd_set_option() values (tuning / A-B switches), thread-safe; `option("name", dflt)` with a string LITERAL costs one atomic load
int option_index(const char* name);
int option_at(int idx, int dflt);
#define option(name, dflt) ::kd::option_at([]() -> int { static const int idx_ = ::kd::option_index(name); return idx_; }(), (dflt))

// ---- per-launch profiling (bench.py): hipEvent pairs around launches when enabled ---------------
struct ProfRec { std::string name; hipEvent_t e0, e1; double flops, bytes; };
bool prof_on();
struct ProfTicket { int idx; unsigned gen; };     // record index + the generation of the record list it belongs to (kd_prof_reset starts a new one)
ProfTicket prof_begin(const char* name, double flops, double bytes, hipStream_t s);      // thread-safe
void prof_end(ProfTicket t, hipStream_t s);

struct LaunchScope {
  hipStream_t s; ProfTicket t;
  LaunchScope(const char* name, double flops, double bytes, hipStream_t st) : s(st), t(prof_on() ? prof_begin(name, flops, bytes, st) : ProfTicket{-1, 0}) {}
  ~LaunchScope() { if (t.idx >= 0) prof_end(t, s); }
};

// hipFuncAttributeMaxDynamicSharedMemorySize of a kernel, set ONCE PER KERNEL AND DEVICE from whichever host thread launches it first.
// One function-local `static LdsAttr` per launch site (= per kernel instantiation): a bit per device ordinal.  hipFuncSetAttribute is
// idempotent, so two threads racing on a kernel's first launch both set the same value before either publishes the device's bit; a thread
// that reads the bit (acquire) launches after the attribute call that published it (release).
struct LdsAttr {
  std::atomic<unsigned long long> done{0};
  void ensure(const void* kern, int bytes) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return;
    (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    done.fetch_or(bit, std::memory_order_release);
  }
};

// CUs of the CURRENT device (hipGetDevice), cached per device ordinal: the grid / split / kernel-selection rules of the launchers use it
int cu_count();

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(KD_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return KD_OK;
}

// ---- device helpers ---------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum_xor(float v, int width) {
#pragma unroll
  for (int o = width >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max_xor(float v, int width) {
#pragma unroll
  for (int o = width >> 1; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// exact (erf) GELU, F.gelu default (image_transformer_v2.py:95)
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// erf to 1e-7 ABSOLUTE error in one range, branch-free: erf(t) = 1 - exp(t * q(t)), q = log(erfc(t)) / t fitted by a
// degree-9 polynomial on [0, 4] (beyond 4 erfc < 2e-8 and the extrapolation stays below 2e-8 up to the clamp).
// The libm erff costs ~3x as many VALU issues; the GEGLU epilogue evaluates erf for every element of the widest
// activation of the network, and there only the absolute error matters (gelu = x/2 * (1 + erf)).
__device__ __forceinline__ float erf_fast(float a) {
  const float t = fminf(fabsf(a), 6.0f);
  float r = -5.2967772e-07f;
  r = fmaf(r, t, 1.1485352e-05f);
  r = fmaf(r, t, -1.0681756e-04f);
  r = fmaf(r, t, 5.4055965e-04f);
  r = fmaf(r, t, -1.4155075e-03f);
  r = fmaf(r, t, -1.7897904e-04f);
  r = fmaf(r, t, 1.9390738e-02f);
  r = fmaf(r, t, -1.0285765e-01f);
  r = fmaf(r, t, -6.3660932e-01f);
  r = fmaf(r, t, -1.1283793e+00f);
  return copysignf(1.0f - __expf(r * t), a);
}
__device__ __forceinline__ float gelu_erf_fast(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f)); }

// GEGLU on two outputs at once:  (value/2) * gate * (1 + erf(gate / sqrt 2)).
// erf(t / sqrt 2) = 1 - 2^(t * Q(t)) with Q = log2(erfc(t / sqrt 2)) / t fitted by a degree-5 polynomial in the gate magnitude
// t itself (the 1/sqrt 2 and log2 e live in the coefficients, the exponential is a bare v_exp_f32): |erf error| <= 3.1e-7 and
// |gelu error| <= 4.8e-7 in fp32 over the whole range (clamp at 8.5, beyond which 2^(tQ) underflows to erf = 1).  The GEGLU
// epilogues are bound by their VALU issue slots (profiles/r01_astat_ablation.md): this form needs ~17 per output where the
// degree-9 erf_fast + separate scaling took ~23.  `vh` = value * 0.5 (the caller folds the 1/2 into its row scale).
using f32x2 = float __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 geglu_pair(f32x2 vh, f32x2 g) {
  const f32x2 t = {fminf(fabsf(g.x), 8.5f), fminf(fabsf(g.y), 8.5f)};
  f32x2 r = f32x2{1.775372766132932e-05f, 1.775372766132932e-05f};
#define KD_PK_STEP(c) r = __builtin_elementwise_fma(r, t, f32x2{(c), (c)})
  KD_PK_STEP(-6.4774916972965e-04f);
  KD_PK_STEP(7.724025286734104e-03f);
  KD_PK_STEP(-5.2926722913980484e-02f);
  KD_PK_STEP(-4.590827524662018e-01f);
  KD_PK_STEP(-1.1511168479919434f);
#undef KD_PK_STEP
  const f32x2 e = r * t;
  const f32x2 om = f32x2{1.0f, 1.0f} - f32x2{__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
  const f32x2 er = {copysignf(om.x, g.x), copysignf(om.y, g.y)};
  const f32x2 vg = vh * g;
  return __builtin_elementwise_fma(vg, er, vg);
}

// ---- code warm-up ----------------------------------------------------------------------------------------------------------
// A kernel that has not run for ~1 ms finds its code evicted from the L2s (4 MB per XCD; several GB of activations went through
// since): the first launch of a forward then walks its code through one exposed instruction-cache miss after the other, ~1.2 us
// per KiB of code on boxes whose memory-side cache does not hold it either (the first layer of every level: 25-32 us on a 37 us
// launch of the 21-26 KiB A-stationary kernels, profiles/r02_level_entry.md).  gfx950 has no instruction prefetch, but
// instruction misses are served by the L2: the first wave of the first 8 workgroups of a launch (one per XCD: workgroups go to the
// XCDs round-robin; 8 measured better than 64, 512 or all) reads the BYTES after the kernel's entry as DATA, all lines at
// once, so that the instruction fetches behind them find the code in L2.  The values are only kept so that the loads stay ordered
// with the kernel's own (code_warm_end after its first vmcnt wait).
// Safety of the range: every code object of the library ends with kd_text_pad_kernel (KD_TEXT_PAD at the end of each .hip file),
// KD_CODE_WARM_MAX + 4 KiB of s_nop behind the last real kernel, so [entry, entry + BYTES) never leaves the loaded image;
// csrc/check_code_objects.py verifies that layout at build time (Makefile, __graft_entry__.build).
constexpr int KD_CODE_WARM_MAX = 32768;
// Validated on: ROCm 7.2.0 (HIP 7.2.26015, AMD clang 22.0.0git roc-7.2.0) -- the code-object layout by check_code_objects.py at every
// build, the behaviour (no result change, +9.4 % images/s) on MI355X with that toolchain (profiles/r02_level_entry.md).  The trick
// rests on how THAT linker lays out a code object (.kd_text_pad directly behind .text in one executable segment) and on
// s_getpc_b64 pointing into it; with any other compiler the default is OFF until someone re-validates (fail closed), and
// check_code_objects.py refuses a layout it does not recognise whatever the version.
#if defined(__clang_major__) && __clang_major__ == 22 && defined(HIP_VERSION_MAJOR) && HIP_VERSION_MAJOR == 7 && HIP_VERSION_MINOR == 2
constexpr int KD_CODE_WARM_DEFAULT = 8;     // workgroups (one per XCD) whose first wave reads the kernel's code range into L2
#else
constexpr int KD_CODE_WARM_DEFAULT = 0;
#endif
template <int TAG>
__global__ __attribute__((section(".kd_text_pad"))) void kd_text_pad_kernel() {            // its own section: the linker puts it
  asm volatile(".fill 9216, 4, 0xbf800000");                                              // behind .text, in the same segment
}                                                                                          // 36 KiB of s_nop; never launched
#define KD_TEXT_PAD(tag) \
  __attribute__((used)) static const void* const kd_text_pad_ref_##tag = reinterpret_cast<const void*>(&kd::kd_text_pad_kernel<0>);

template <int BYTES> struct CodeWarm { int v[(BYTES + 4095) / 4096]; };
template <int BYTES>
__device__ __forceinline__ CodeWarm<BYTES> code_warm_begin(bool on) {
  static_assert(BYTES % 64 == 0 && BYTES >= 64 && BYTES <= KD_CODE_WARM_MAX, "code warm-up range");
  constexpr int N = (BYTES + 4095) / 4096;
  CodeWarm<BYTES> w;
#pragma unroll
  for (int i = 0; i < N; ++i) w.v[i] = 0;
  if (on) {
    unsigned long long pc;
    asm volatile("s_getpc_b64 %0" : "=s"(pc));
    const char* base = reinterpret_cast<const char*>(pc & ~63ull);
    const int lo = (threadIdx.x & 63) * 64;
#pragma unroll
    for (int i = 0; i < N; ++i) w.v[i] = *reinterpret_cast<const volatile int*>(base + min(i * 4096 + lo, BYTES - 64));
  }
  return w;
}
template <int BYTES>
__device__ __forceinline__ void code_warm_end(const CodeWarm<BYTES>& w) {
#pragma unroll
  for (int i = 0; i < (BYTES + 4095) / 4096; ++i) asm volatile("" ::"v"(w.v[i]));
}

constexpr int KD_ROT = 16;       // rotary angles per head: dims [0,16) pair with [16,32)
// ---- DPP helpers: cross-lane moves inside a 16-lane row as plain VALU ops (no LDS round trip like ds_bpermute) ----
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_ROR4 = 0x124, DPP_ROR8 = 0x128, DPP_ROR12 = 0x12C;
// sum over the 16 lanes of a row, result in every lane
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<DPP_ROR8>(v);
  v += dpp_mov<DPP_ROR4>(v);
  v += dpp_mov<DPP_XOR2>(v);
  v += dpp_mov<DPP_XOR1>(v);
  return v;
}

// ---- q/k row preparation, 16 lanes per 64-float row: lane c = lane & 15 owns dims [4c, 4c+4) -------
// scale_for_cosine_sim (image_transformer_v2.py:106-114) then _apply_rotary_emb_inplace (:187-199): dims d < 16 pair
// with d + 16, i.e. lane c < 4 with lane c + 4 (row_ror:12 brings lane c+4's value to lane c, row_ror:4 lane c-4's).
// cs / sn: this lane's chunk (c & 3) of the row's 16 cos / sin values (lanes c >= 8 ignore them).
__device__ __forceinline__ f32x4 prep_row16_regs(f32x4 v, int c, float sqrt_scale, f32x4 cs, f32x4 sn, float eps) {
  const float ss = row16_sum(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
  const float f = sqrt_scale * rsqrtf(ss + eps);
  v = v * f;
  f32x4 up, dn;
#pragma unroll
  for (int u = 0; u < 4; ++u) { up[u] = dpp_mov<DPP_ROR12>(v[u]); dn[u] = dpp_mov<DPP_ROR4>(v[u]); }
  const f32x4 rot = (c < 4) ? (v * cs - up * sn) : (v * cs + dn * sn);
  return c < 8 ? rot : v;
}
__device__ __forceinline__ f32x4 prep_row16(f32x4 v, int c, float sqrt_scale, const float* cs_row, const float* sn_row, float eps) {
  const f32x4 cs = *reinterpret_cast<const f32x4*>(cs_row + 4 * (c & 3));
  const f32x4 sn = *reinterpret_cast<const f32x4*>(sn_row + 4 * (c & 3));
  return prep_row16_regs(v, c, sqrt_scale, cs, sn, eps);
}

// 32x32 MFMA C/D fragment: element `reg` of lane `lane` is C[row][col]
__device__ __forceinline__ int mfma32_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

}  // namespace kd
