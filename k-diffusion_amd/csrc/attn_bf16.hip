// Attention cores of the HDiT denoiser in bf16 mode (KD_PREC_BF16), gfx950.
//
//   kd_attn_global_bf16  dense softmax attention per (sample, head): T <= 256 keys held whole in LDS, longer sequences streamed
//                        in 128-key blocks with an online softmax          (SDPA / flash-attn, image_transformer_v2.py:383,392)
//   kd_attn_window_bf16  shifted-window attention, roll / window / mask / unwindow as index arithmetic      (:319-337, :253-316)
//   kd_attn_na2d_bf16    neighbourhood attention, clamped windows, kernel sizes 3, 5, 7, 9                     (natten na2d, :428)
//
// qkv is the bf16 output of the qkv GEMM [tokens, 3, nh, 64] with q, k ALREADY prepared (cosine-sim scale + RoPE in that GEMM's
// epilogue); out is bf16 [tokens, nh * 64].  One scheme for all three cores:
//   * K rows and V rows go HBM -> LDS by global_load_lds, 8 rows x 128 bytes per wave-instruction, as row-major [key][64] images
//     whose 16-byte chunks are XOR-swizzled on the source side (asw() below): no staging VALU work, no ds_write pass;
//   * S^T = K Q^T : K rows are the MFMA A operand (conflict-free ds_read_b128), Q the B operand straight from HBM registers; a
//     lane owns ONE query (column) and 16 keys per 32-key tile, so the softmax reductions are in-lane plus one half-wave shuffle;
//   * O^T = V^T P^T : P (fp32 -> bf16 in registers) is the B operand as it sits in the accumulators; the V^T fragments come out
//     of the ROW-major V image through ds_read_b64_tr_b16 (hardware 4x4 transpose: lane i of a 16-lane group receives column i
//     of the four rows the group addresses -- benchmarks/probe/ds_read_tr_probe.cpp records the mapping);
//   * the O^T accumulators are the C-layout of the bf16 GEMM epilogues: normalise, pair the half-waves, 16-byte stores.
// fp32: scores, softmax, accumulators.  bf16: q, k, v, P, out.
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

// ---- dense core, the whole key set in LDS: global (T <= 256) and windows --------------------------------------------------
// NT = key tiles of 32; QW = waves per workgroup, wave w owns queries 32 (qblk * QW + w) ..; several workgroups per (sample, head)
// when QW < NT (each stages the full K / V: L2 hits, and the chip sees 2+ workgroups per CU instead of one).
template <int MODE, int NT, int QW>
__global__ __launch_bounds__(QW * 64, (QW <= 4 && MODE != MODE_WINDOW16) ? 2 : 1) void attn_dense_bf16_kernel(const DArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TP = NT * 32;
  char* Kimg = smem;
  char* Vimg = smem + TP * 128;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, h2 = lane >> 5;
  const auto warm = code_warm_begin<(MODE == MODE_WINDOW16 ? 16 : 10) * 1024>((int)blockIdx.x < a.warm && tid < 64);
  constexpr int NQB = (NT + QW - 1) / QW;            // query blocks per problem
  int r = blockIdx.x;
  const int qblk = r % NQB; r /= NQB;
  int b, head, wi = 0, wj = 0;
  if (MODE == MODE_GLOBAL) {
    head = r % a.nh; b = r / a.nh;
  } else {
    const int nww = a.W >> WinLog2<MODE>::v, nwh = a.H >> WinLog2<MODE>::v;
    wj = r % nww; r /= nww; wi = r % nwh; r /= nwh; head = r % a.nh; b = r / a.nh;
  }
  const int T = MODE == MODE_GLOBAL ? a.T : (1 << (2 * WinLog2<MODE>::v));          // key slots of this problem
  const size_t row_bytes = (size_t)3 * a.nh * DH * 2;
  const char* base = reinterpret_cast<const char*>(a.qkv) + (size_t)b * a.T * row_bytes + head * (DH * 2);

  // ---- K and V rows -> LDS images (8 rows per wave-instruction) ----------------------------------------------------------
  for (int pc = wid; pc < TP / 8; pc += QW) {
    const int row = 8 * pc + (lane >> 3);
    const int tok = slot_token<MODE>(a, min(row, T - 1), wi, wj);
    const int q = (lane & 7) ^ asw(row);
    const char* src = base + (size_t)tok * row_bytes + q * 16;
    glds16(src + a.nh * DH * 2, Kimg + pc * 1024);
    glds16(src + 2 * a.nh * DH * 2, Vimg + pc * 1024);
  }
  // ---- this lane's query: dims 16 st + 8 h2 .. +7 ------------------------------------------------------------------------------
  const int q_slot = (qblk * QW + wid) * 32 + l31;
  const bool q_ok = q_slot < T;
  const int q_tok = slot_token<MODE>(a, min(q_slot, T - 1), wi, wj);
  bf16x8 qf[4];
  {
    const u32x4* qp = reinterpret_cast<const u32x4*>(base + (size_t)q_tok * row_bytes + 16 * h2);
#pragma unroll
    for (int st = 0; st < 4; ++st) qf[st] = __builtin_bit_cast(bf16x8, qp[2 * st]);
  }
  KD_WAIT_VM(0);
  code_warm_end(warm);
  KD_BARRIER();
  if ((qblk * QW + wid) * 32 >= T) return;            // a wave without queries (no barrier follows)

  // ---- S^T = K Q^T ----------------------------------------------------------------------------------------------------------------
  f32x16 S[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) S[t][i] = 0.f;
  const int ka = l31 * 128 + ((h2 ^ asw(l31)) << 4);          // K fragment of tile t, k-step st: (ka ^ 32 st) + 4096 t
#pragma unroll
  for (int st = 0; st < 4; ++st) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Kimg + (ka ^ (32 * st)) + 4096 * t);
      S[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[st], S[t], 0, 0, 0);
    }
  }
  // ---- masks + softmax over keys -----------------------------------------------------------------------------------------------
  const int q_region = (MODE != MODE_GLOBAL) ? slot_region<MODE>(min(q_slot, T - 1), wi, wj, a.shift) : 0;
  if (MODE != MODE_GLOBAL || (T & 31)) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int ks = t * 32 + mfma32_row(i, lane);
        if (MODE == MODE_GLOBAL) {
          if (t * 32 + 32 > T) S[t][i] += (ks < T) ? 0.f : -INFINITY;
        } else {
          if ((1 << (2 * WinLog2<MODE>::v)) < TP) S[t][i] += (ks < T) ? 0.f : -INFINITY;
          if (a.shift) S[t][i] += (slot_region<MODE>(min(ks, T - 1), wi, wj, a.shift) == q_region) ? 0.f : -INFINITY;
        }
      }
  }
  const float m = score_max<NT>(S);
  float l = score_exp<NT>(S, m);
  l += __shfl_xor(l, 32, 64);

  // ---- O^T = V^T P^T ---------------------------------------------------------------------------------------------------------------
  f32x16 O[2];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int i = 0; i < 16; ++i) O[e][i] = 0.f;
  const int va = vt_addr(4 * h2 + vt_lane_row(lane), lane);      // k-step (t, u): rows + 32 t + 16 u, same swizzle word
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) pv_step(O, Vimg + (32 * t + 16 * u) * 128, va, p_frag(S[t], u));
  store_o(a.out + ((size_t)b * a.T + q_tok) * (a.nh * DH) + head * DH, O, 1.0f / l, h2, q_ok);
}

// ---- global-attention block in ONE launch: AdaRMSNorm -> qkv projection of one head -> cosine-sim + RoPE -> dense attention ----------
// (image_transformer_v2.py:370-396: norm, qkv_proj, scale_for_cosine_sim_qkv, apply_rotary_emb_, attention -- everything in front of
// out_proj.)  At the level-2 shape (8192 rows, K = 512) the two-launch form spent 29.5 us in the projection (a 4 us K loop per wave
// between a row prologue repeated per n-split and the q / k epilogues) and 13.5 us in a core that has 1.7 us of work, with a 25 MB
// qkv round trip through HBM between them.  Here a workgroup owns ONE (sample, head) problem with T = 256 tokens:
//   * its 8 waves take 32 rows each: the sample's rows come in by LDS-DMA once per workgroup, are normalised and scaled into
//     MFMA B-operand fragments held in registers (the A-stationary form of csrc/gemm_bf16.hip, same arithmetic, same order);
//   * the head's 64 rows of W_k, W_v, W_q stream past them (three passes over K, 8 KiB half blocks of the packed image, 4-slot ring of
//     two k-steps each, one barrier per 16 MFMAs per wave); k and v leave their epilogues as bf16 rows of the K / V images in LDS
//     (the dense core's swizzle), q -- the last pass -- stays in registers as the B fragments of S^T = K Q^T;
//   * then the dense core above, unchanged: scores, softmax, O^T = V^T P^T, 16-byte stores of the attention output.
// Neither q, k nor v ever reaches HBM.  The arithmetic is that of the two-launch form operation for operation (the results are
// bit-identical: tests/test_ops_gpu.py::test_attn_block_bf16_matches_two_launches).  144 KiB of LDS, one workgroup per CU.
struct BArgs {
  const u16* x; const char* Wp; u16* out;
  const float* scale; int scale_stride; float eps;
  int batch, nh;                      // 256 tokens per sample
  const float* qk_scale; const float* pos; const float* freq;
  int warm;
  unsigned long long* clk;            // kd_prof_clock_buffer (TS instantiation only): workgroup 0's time line, s_memtime stamps
  const char* Wp_out; u16* xio; int* sync;      // OUTP: packed out-projection weight, the residual stream (= x, updated in place), per-sample counters
};
extern unsigned long long* g_clk;     // gemm_bf16.hip

__device__ __forceinline__ void wait_vm_n(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
  }
}

// one 64-dim row held as two C-layout blocks -> its 128-byte row of a [token][64] bf16 image in LDS (chunk c at c ^ asw(row))
__device__ __forceinline__ void row_to_image(char* img, int rowi, const f32x16& b0, const f32x16& b1, float mul, int lh) {
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    unsigned pk[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x16& v = e ? b1 : b0;
      pk[g][0] = pack_bf16(v[4 * g] * mul, v[4 * g + 1] * mul);
      pk[g][1] = pack_bf16(v[4 * g + 2] * mul, v[4 * g + 3] * mul);
    }
#pragma unroll
    for (int gp = 0; gp < 4; gp += 2) {
      half_swap(pk[gp][0], pk[gp + 1][0]);
      half_swap(pk[gp][1], pk[gp + 1][1]);
      const int c = 4 * e + gp + lh;                       // dims 8 c .. 8 c + 7 of the row
      *reinterpret_cast<u32x4*>(img + rowi * 128 + ((c ^ asw(rowi)) << 4)) = u32x4{pk[gp][0], pk[gp][1], pk[gp + 1][0], pk[gp + 1][1]};
    }
  }
}

// Rendezvous of the n workgroups that share a sample (OUTP): one thread per workgroup arrives and waits until all n have.  They are resident
// together -- ids 8 apart inside one group of 8 n consecutive ids, one workgroup per CU, dispatched in id order -- so nobody waits for a
// workgroup that cannot start; a bounded wait (~1 s) guards the assumption: on expiry the flag behind the counters is set and the workgroup goes
// on (wrong numbers, reported by the host wrapper's check, no hang).
// Memory order.  With the XCD-aware placement (batch % 8 == 0) the n workgroups sit on ONE XCD and exchange through its L2: a store is
// acknowledged by the L2 (vmcnt), atomics execute there, and the readers fetch the attention rows with sc1 loads that do not stop in their
// CU's vector cache -- no cache maintenance at all (`same_l2`).  An agent-scope release / acquire pair instead writes back and invalidates the
// WHOLE L2 (buffer_wbl2 / buffer_inv sc1): measured 15 k clocks per rendezvous and an out-projection pass that then missed on every line; it is
// kept for the plain placement only, where the workgroups of a sample are spread over the XCDs.
__device__ __forceinline__ void sample_rendezvous(int* arrive, int target, int* flag, bool same_l2) {
  if (!same_l2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __hip_atomic_fetch_add(arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int spins = 0;
  while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
    __builtin_amdgcn_s_sleep(2);
    if (++spins > (1 << 23)) { __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
  }
  if (!same_l2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
__device__ __forceinline__ void glds16_sc1(const void* src, void* dst) {       // sc1: served by the L2, not by this CU's vector cache
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 16);
}

// TS: in-kernel time line of workgroup 0 (kd_prof_clock_buffer; a template flag so that the measured kernel's loops stay what they are):
// [0] entry, [4] rows normalised, [5] / [6] / [8] end of the k / v / q pass, [9] scores + softmax done, [2] exit, [1] / [3] s_memrealtime.
// The 32 rows of one wave (rows 32 wid .. of sample b) -> normalised, scaled MFMA B fragments `a[NC]` held in registers, and the rows' RMS
// factor `rs`.  gemm_astat_kernel's arithmetic (same products, same order of the sum of squares), another staging schedule: the rows come in by
// K-HALVES of 256 elements -- all 32 rows x 512 bytes per round into the wave's 16 KiB slot, every lane reads its own row's 16 chunks (at
// K = 512 the row-halves schedule of the projection kernel leaves half the lanes idle in each round) -- and the second half is in flight while
// the first is converted; scales two chunks at a time, one pair ahead (32 registers of scales: the block kernels carry 2 waves per SIMD).
// A MACRO, not a function: as a forceinline function taking / returning the fragments the same code compiled to 40 more registers and, at
// K = 512, 600 bytes of scratch per lane.  Uses the enclosing kernel's smem, wid, lane, l31, lh and the constants NC, K; declares a, rs.
// ROW0: first of the workgroup's 256 rows.
#define KD_ROWS_TO_FRAGMENTS(XPTR, ROW0, SVEC, EPS) \
  bf16x8 a[NC]; \
  float rs; \
  { \
    constexpr int NH = K / 256; \
    char* stage = smem + wid * WBLK; \
    char* scl = smem + 8 * WBLK + wid * (K * 4); \
    const char* ssrc = reinterpret_cast<const char*>((SVEC)) + lane * 16; \
_Pragma("unroll") \
    for (int i = 0; i < K * 4 / 1024; ++i) glds16(ssrc + i * 1024, scl + i * 1024); \
    auto request = [&](int h) { \
_Pragma("unroll") \
      for (int i = 0; i < 16; ++i) { \
        const int rr = (i * 64 + lane) >> 5, qs = (i * 64 + lane) & 31; \
        const size_t grow = (size_t)(ROW0) + wid * 32 + rr; \
        glds16(reinterpret_cast<const char*>((XPTR) + grow * K) + h * 512 + ((qs ^ (rr & 15)) << 4), stage + i * 1024); \
      } \
    }; \
    float ssq = 0.f; \
    const float* spl = reinterpret_cast<const float*>(scl) + 8 * lh; \
    const char* rowp = stage + l31 * 512; \
    request(0); \
_Pragma("unroll") \
    for (int h = 0; h < NH; ++h) { \
      u32x4 raw[16]; \
      KD_WAIT_VM(0); \
_Pragma("unroll") \
      for (int c = 0; c < 16; ++c) raw[c] = *reinterpret_cast<const u32x4*>(rowp + (((2 * c + lh) ^ (l31 & 15)) << 4)); \
      if (h + 1 < NH) { \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        request(h + 1); \
      } \
      f32x4 s0[2][2], s1[2][2]; \
      auto load_scales = [&](int c0, int g) { \
_Pragma("unroll") \
        for (int u = 0; u < 2; ++u) { \
          s0[g][u] = *reinterpret_cast<const f32x4*>(spl + 16 * (16 * h + c0 + u)); \
          s1[g][u] = *reinterpret_cast<const f32x4*>(spl + 16 * (16 * h + c0 + u) + 4); \
        } \
      }; \
      load_scales(0, 0); \
      __builtin_amdgcn_sched_barrier(0); \
_Pragma("unroll") \
      for (int c0 = 0; c0 < 16; c0 += 2) { \
        const int g = (c0 >> 1) & 1; \
        if (c0 + 2 < 16) load_scales(c0 + 2, g ^ 1); \
_Pragma("unroll") \
        for (int u = 0; u < 2; ++u) { \
          float x[8]; \
_Pragma("unroll") \
          for (int e = 0; e < 4; ++e) { x[2 * e] = bf_lo(raw[c0 + u][e]); x[2 * e + 1] = bf_hi(raw[c0 + u][e]); } \
_Pragma("unroll") \
          for (int e = 0; e < 8; ++e) ssq = fmaf(x[e], x[e], ssq); \
          u32x4 o = {pack_bf16(x[0] * s0[g][u][0], x[1] * s0[g][u][1]), pack_bf16(x[2] * s0[g][u][2], x[3] * s0[g][u][3]), \
                     pack_bf16(x[4] * s1[g][u][0], x[5] * s1[g][u][1]), pack_bf16(x[6] * s1[g][u][2], x[7] * s1[g][u][3])}; \
          asm volatile("" : "+v"(o)); \
          a[16 * h + c0 + u] = __builtin_bit_cast(bf16x8, o); \
        } \
        __builtin_amdgcn_sched_barrier(0); \
      } \
    } \
    ssq += __shfl_xor(ssq, 32, 64); \
    rs = rsqrtf(ssq / (float)K + (EPS)); \
  }

// OUTP: the block's out projection + residual in the same launch (image_transformer_v2.py:393-396).  The attention output of a sample is
// complete when its n_heads workgroups have stored their 64 columns; after a rendezvous of those workgroups (same XCD: the exchange stays in
// one L2) workgroup (sample, h) takes the sample's 256 attention rows as B fragments (LDS-DMA, no arithmetic) and runs ONE more pass: the 64
// rows 64 h .. of W_out, i.e. the output columns 64 h .. of the new residual stream, x[:, 64 h ..] += att W_out^T -- written in place (every
// workgroup of the sample finished reading x before it arrived; column slices are disjoint).  Same products in the same order as the tiled
// projection kernel it replaces: bit-identical.
template <int NC /* K / 16 */, bool OUTP = false, bool TS = false>
__global__ __launch_bounds__(512, 1) void attn_block_bf16_kernel(const BArgs p) {
  constexpr int K = NC * 16, NK = NC / 4, SPP = NK / 2, NSTAGE = 3 * SPP, NSLOT = 4, PDIST = 3, T = 256, NT = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Kimg = smem + NSLOT * WBLK;
  char* Vimg = Kimg + T * 128;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lh = lane >> 5;
  const auto warm = code_warm_begin<24 * 1024>((int)blockIdx.x < p.warm && tid < 64);
  const bool probe = TS && blockIdx.x == 0 && tid == 0;
  if (probe) { p.clk[0] = __builtin_amdgcn_s_memtime(); p.clk[1] = __builtin_amdgcn_s_memrealtime(); }
  // extended form (x3_common.h: wg_stamp_begin): every workgroup's entry / exit in 100 MHz ticks at clk[32 + 3 b], + 1
  unsigned long long* wg_slot = (TS && tid == 0 && p.clk[15] == 0x4b44ull) ? p.clk + 32 + 3 * blockIdx.x : nullptr;
  if (TS && wg_slot) wg_slot[0] = __builtin_amdgcn_s_memrealtime();
  // workgroup -> (sample, head): ids go to the 8 XCDs round-robin, so the heads of one sample get ids 8 apart -- one XCD's L2 fetches the
  // sample's rows from HBM once for all of them
  int b, head;
  if ((p.batch & 7) == 0) {
    const int j = blockIdx.x >> 3;
    b = (j / p.nh) * 8 + (blockIdx.x & 7);
    head = j % p.nh;
  } else {
    b = blockIdx.x / p.nh;
    head = blockIdx.x % p.nh;
  }
  const int tok = wid * 32 + l31;                         // this lane's token of the sample: its row, later its query
  const size_t row = (size_t)b * T + tok;

  // ---- the wave's 32 rows -> normalised, scaled B fragments ---------------------------------------------------------------------------
  KD_ROWS_TO_FRAGMENTS(p.x, (size_t)b * T, p.scale + (size_t)b * p.scale_stride, p.eps)
  float py = p.pos[2 * tok], px = p.pos[2 * tok + 1];
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(py), "+v"(px) : : "memory");
  code_warm_end(warm);
  KD_BARRIER();                                        // every wave has taken its rows out of the slot it borrowed
  if (probe) p.clk[4] = __builtin_amdgcn_s_memtime();

  // ---- the head's W_k, W_v, W_q rows past the fragments -------------------------------------------------------------------------------
  // stage s = pass s / SPP (k, v, q), k-steps 2 kk and 2 kk + 1 of that pass: two 8 KiB half blocks (rows 64 (head & 1) .. + 63 of the
  // block (n-tile of the head's rows, k-step)); wave w brings piece w of each
  const char* wbase = p.Wp + (head & 1) * 8192 + wid * 1024 + lane * 16;
  auto issue = [&](int s) {
    const int pass = s / SPP, kk = s % SPP;
    const int which = pass == 0 ? 1 : (pass == 1 ? 2 : 0);
    const int nt = (which * K + head * 64) >> 7;
    char* dst = smem + (s % NSLOT) * WBLK + wid * 1024;
    glds16(wbase + ((size_t)nt * NK + 2 * kk) * WBLK, dst);
    glds16(wbase + ((size_t)nt * NK + 2 * kk + 1) * WBLK, dst + 8192);
  };
#pragma unroll
  for (int s = 0; s < PDIST; ++s) issue(s);

  int off4[4];
#pragma unroll
  for (int cc = 0; cc < 4; ++cc) off4[cc] = swz128(l31, 2 * cc + lh);
  // the head's constants through the scalar cache (see gemm_astat_kernel)
  typedef float f32x8s __attribute__((ext_vector_type(8)));
  f32x8s fq;
  float qsc;
  asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dword %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
               : "=s"(fq), "=s"(qsc) : "s"(p.freq + head * 8), "s"(p.qk_scale + head) : "memory");
  float fr[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) fr[u] = pick_half(fq[u], fq[4 + u], 0u - (unsigned)lh);
  const float sqs = sqrtf(qsc);

  bf16x8 qf[4];
  f32x16 acc[2];
#pragma unroll
  for (int s = 0; s < NSTAGE; ++s) {
    const int pass = s / SPP, kk = s % SPP;
    if (kk == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    }
    wait_vm_n(2 * min(PDIST - 1, NSTAGE - 1 - s));
    KD_BARRIER();                        // every wave's pieces of stage s are in; everyone is done reading slot (s - 1) % NSLOT
    if (s + PDIST < NSTAGE) issue(s + PDIST);
    const char* st = smem + (s % NSLOT) * WBLK;
    bf16x8 wf[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) wf[0][j] = *reinterpret_cast<const bf16x8*>(st + j * 32 * 128 + off4[0]);
#pragma unroll
    for (int c8 = 0; c8 < 8; ++c8) {     // 8 chunks of 16 k: half block h = c8 / 4, chunk cc = c8 % 4
      if (c8 + 1 < 8) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
          wf[(c8 + 1) & 1][j] = *reinterpret_cast<const bf16x8*>(st + ((c8 + 1) >> 2) * 8192 + j * 32 * 128 + off4[(c8 + 1) & 3]);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[c8 & 1][j], a[4 * (2 * kk + (c8 >> 2)) + (c8 & 3)], acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kk == SPP - 1) {
      if (pass == 0) {                   // k: cosine-sim scale + RoPE, then its row of the K image
        qk_prep_blocks(acc[0], acc[1], rs, sqs, p.eps, py, px, fr);
        row_to_image(Kimg, tok, acc[0], acc[1], 1.0f, lh);
      } else if (pass == 1) {            // v
        row_to_image(Vimg, tok, acc[0], acc[1], rs, lh);
      } else {                           // q: stays in registers as the B fragments of the score products (dims 16 st + 8 lh .. + 7)
        qk_prep_blocks(acc[0], acc[1], rs, sqs, p.eps, py, px, fr);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          unsigned pk[4][2];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            pk[g][0] = pack_bf16(acc[e][4 * g], acc[e][4 * g + 1]);
            pk[g][1] = pack_bf16(acc[e][4 * g + 2], acc[e][4 * g + 3]);
          }
#pragma unroll
          for (int gp = 0; gp < 4; gp += 2) {
            half_swap(pk[gp][0], pk[gp + 1][0]);
            half_swap(pk[gp][1], pk[gp + 1][1]);
            qf[2 * e + gp / 2] = __builtin_bit_cast(bf16x8, u32x4{pk[gp][0], pk[gp][1], pk[gp + 1][0], pk[gp + 1][1]});
          }
        }
      }
      if (probe) p.clk[pass == 0 ? 5 : (pass == 1 ? 6 : 8)] = __builtin_amdgcn_s_memtime();
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  KD_BARRIER();                                        // all 256 rows of the K and V images are written

  // ---- S^T = K Q^T, softmax, O^T = V^T P^T: the dense core ------------------------------------------------------------------------------
  f32x16 S[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) S[t][i] = 0.f;
  const int ka = l31 * 128 + ((lh ^ asw(l31)) << 4);
#pragma unroll
  for (int st = 0; st < 4; ++st) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Kimg + (ka ^ (32 * st)) + 4096 * t);
      S[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[st], S[t], 0, 0, 0);
    }
  }
  const float m = score_max<NT>(S);
  float l = score_exp<NT>(S, m);
  l += __shfl_xor(l, 32, 64);
  if (probe) p.clk[9] = __builtin_amdgcn_s_memtime();
  f32x16 O[2];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int i = 0; i < 16; ++i) O[e][i] = 0.f;
  const int va = vt_addr(4 * lh + vt_lane_row(lane), lane);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) pv_step(O, Vimg + (32 * t + 16 * u) * 128, va, p_frag(S[t], u));
  store_o(p.out + row * (size_t)(p.nh * DH) + head * DH, O, 1.0f / l, lh, true);
  if (OUTP) {
    if (probe) p.clk[10] = __builtin_amdgcn_s_memtime();
    // the residual operand: this lane's row, columns 64 head .. + 63 of the OLD x (nobody writes them but this workgroup, below)
    u16* xrow = p.xio + row * (size_t)K + head * DH;
    u32x4 rraw[2][2];
    load_block_raw(xrow, rraw[0], lh);
    load_block_raw(xrow + 32, rraw[1], lh);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's attention rows are out (and the residual pieces in)
    KD_BARRIER();
    if (tid == 0) {
      sample_rendezvous(p.sync + 2 * b, p.nh, p.sync + 2 * p.batch, (p.batch & 7) == 0);
      // everybody of this sample is past its wait once all have departed: the last one clears the counters for the next launch
      if (__hip_atomic_fetch_add(p.sync + 2 * b + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == p.nh - 1) {
        __hip_atomic_store(p.sync + 2 * b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p.sync + 2 * b + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    KD_BARRIER();
    if (probe) p.clk[11] = __builtin_amdgcn_s_memtime();
    bf16x8 a2[NC];                                    // (its own array: the live range of the projection fragments ends with the q pass)
    // ---- the sample's attention rows -> B fragments (the staging of the prologue, no arithmetic) --------------------------------------------
    {
      constexpr int RPR = WBLK / (2 * K), NR = 32 / RPR, CPR = K / 8;
      char* stage = smem + wid * WBLK;
      int lane_o = lane, l31_o = l31, lh_o = lh;      // opaque copies: the source offsets and read addresses below are those of the prologue, and
      asm volatile("" : "+v"(lane_o), "+v"(l31_o), "+v"(lh_o));   // CSE kept all of them alive (in scratch) across the whole kernel instead of recomputing them
#pragma unroll
      for (int r = 0; r < NR; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int rr = (i * 64 + lane_o) / CPR, qs = (i * 64 + lane_o) % CPR;
          const size_t grow = (size_t)b * T + wid * 32 + r * RPR + rr;
          glds16_sc1(reinterpret_cast<const char*>(p.out + grow * K) + ((qs ^ (rr & 15)) << 4), stage + i * 1024);
        }
        KD_WAIT_VM(0);
        if (NR == 1 || (l31_o / RPR) == r) {
          const int rr = l31_o % RPR;
          const char* rowp = stage + rr * (2 * K);
#pragma unroll
          for (int c = 0; c < NC; ++c) a2[c] = *reinterpret_cast<const bf16x8*>(rowp + (((2 * c + lh_o) ^ (rr & 15)) << 4));
          // a real branch (exec mask), not 128 selects between the old and the new fragments: an if-converted form needs both sets live
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
    KD_BARRIER();                                      // the staging slots become the ring again
    const char* wo = p.Wp_out + ((size_t)(head >> 1) * NK) * WBLK + (head & 1) * 8192 + wid * 1024 + lane * 16;
    auto issue_o = [&](int s) {
      char* dst = smem + (s % NSLOT) * WBLK + wid * 1024;
      glds16(wo + (size_t)(2 * s) * WBLK, dst);
      glds16(wo + (size_t)(2 * s + 1) * WBLK, dst + 8192);
    };
    constexpr int PD2 = SPP < PDIST ? SPP : PDIST;
#pragma unroll
    for (int s = 0; s < PD2; ++s) issue_o(s);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int s = 0; s < SPP; ++s) {
      wait_vm_n(2 * min(PD2 - 1, SPP - 1 - s));
      KD_BARRIER();
      if (s + PD2 < SPP) issue_o(s + PD2);
      const char* st = smem + (s % NSLOT) * WBLK;
      bf16x8 wf[2][2];
#pragma unroll
      for (int j = 0; j < 2; ++j) wf[0][j] = *reinterpret_cast<const bf16x8*>(st + j * 32 * 128 + off4[0]);
#pragma unroll
      for (int c8 = 0; c8 < 8; ++c8) {
        if (c8 + 1 < 8) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
            wf[(c8 + 1) & 1][j] = *reinterpret_cast<const bf16x8*>(st + ((c8 + 1) >> 2) * 8192 + j * 32 * 128 + off4[(c8 + 1) & 3]);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[c8 & 1][j], a2[4 * (2 * s + (c8 >> 2)) + (c8 & 3)], acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      float v[16], rr_[16];
      block_from_raw(rraw[e], rr_);
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = acc[e][r] + rr_[r];
      store_block_bf16(xrow + 32 * e, v, lh, true);
    }
  }
  if (probe) { p.clk[2] = __builtin_amdgcn_s_memtime(); p.clk[3] = __builtin_amdgcn_s_memrealtime(); }
  if (TS && wg_slot) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); wg_slot[1] = __builtin_amdgcn_s_memrealtime(); }
}

// ---- AdaRMSNorm -> wide projection (GEGLU up projection, qkv) in the attention block's form ------------------------------------------------
// (image_transformer_v2.py:487-491: norm, up_proj / linear_geglu; :370-380 / :415-425: norm, qkv_proj, cosine-sim scale, RoPE -- for the levels
// whose attention core is a separate launch.)  The A-stationary projection kernel (gemm_bf16.hip: gemm_astat_kernel) runs these shapes --
// 8 192 rows x K = 512 x 3 072 W rows, 32 768 rows x K = 256 x 1 536 / 768 W rows -- as 128-row panels x n-splits: every split repeats the
// panel's row prologue, and two workgroups per CU re-stream up to 1 MiB of rows + weights through a 37 - 50 bytes / clock L2 -> LDS path.
// Here a workgroup owns (256-row group, slice of six 64-row half blocks of the packed image): its 8 waves normalise the group's rows ONCE into
// register fragments (KD_ROWS_TO_FRAGMENTS) and then run six passes over K through the attention block's 4-slot ring; a pass ends with the
// epilogue of the lane's own row (GEGLU of 32 value / 32 gate columns; cosine-sim scale + RoPE of a q / k head vector; the row factor for v) and
// its 16-byte stores.  Same products in the same order as gemm_astat_kernel<NC, EPI>: bit-identical.
struct UArgs {
  const u16* x; const char* Wp; u16* out;
  const float* scale; int scale_stride; float eps;
  int groups, groups_per_sample, slices, n_out;      // 256-row groups; groups of one sample; slices of 6 half blocks; output row width
  int n_heads; const float* qk_scale; const float* pos; const float* freq;      // EPI_QKV
  int warm;
};

__device__ __forceinline__ void wait_vm_any(int n) {
  switch (n) {
#define KD_C(v) case v: asm volatile("s_waitcnt vmcnt(" #v ")" ::: "memory"); break;
    KD_C(0) KD_C(1) KD_C(2) KD_C(3) KD_C(4) KD_C(5) KD_C(6) KD_C(7) KD_C(8) KD_C(9) KD_C(10) KD_C(11) KD_C(12) KD_C(13) KD_C(14) KD_C(15) KD_C(16)
#undef KD_C
    default: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
  }
}

template <int NC /* K / 16 */, int EPI>
__global__ __launch_bounds__(512, 1) void proj_block_bf16_kernel(const UArgs p) {
  constexpr int K = NC * 16, NK = NC / 4, SPP = NK / 2, NPASS = 6, NSTAGE = NPASS * SPP, NSLOT = 4, PDIST = 3;
  constexpr int NST = EPI == KD_EPI_GEGLU ? 2 : 4;      // 16-byte stores per lane at the end of a pass
  constexpr int COLS = EPI == KD_EPI_GEGLU ? 32 : 64;   // output columns of a half block
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lh = lane >> 5;
  const auto warm = code_warm_begin<24 * 1024>((int)blockIdx.x < p.warm && tid < 64);
  int grp, slice;                                        // the slices of one row group on one XCD (ids 8 apart), as the attention block's heads
  if ((p.groups & 7) == 0) {
    const int j = blockIdx.x >> 3;
    grp = (j / p.slices) * 8 + (blockIdx.x & 7);
    slice = j % p.slices;
  } else {
    grp = blockIdx.x / p.slices;
    slice = blockIdx.x % p.slices;
  }
  const int b = grp / p.groups_per_sample;
  const size_t row0 = (size_t)grp * 256;
  const size_t row = row0 + wid * 32 + l31;
  KD_ROWS_TO_FRAGMENTS(p.x, row0, p.scale + (size_t)b * p.scale_stride, p.eps)
  float py = 0.f, px = 0.f;
  if (EPI == KD_EPI_QKV) {
    const int tok = (int)(row - (size_t)b * p.groups_per_sample * 256);
    py = p.pos[2 * tok];
    px = p.pos[2 * tok + 1];
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(py), "+v"(px) : : "memory");
  }
  code_warm_end(warm);
  KD_BARRIER();                                        // every wave has taken its rows out of the slot it borrowed

  // stage s = pass s / SPP (half block 6 slice + pass), k-steps 2 kk and 2 kk + 1; wave w brings piece w of each half block
  const char* wbase = p.Wp + wid * 1024 + lane * 16;
  auto issue = [&](int s) {
    const int hb = NPASS * slice + s / SPP, kk = s % SPP;
    const char* src = wbase + ((size_t)(hb >> 1) * NK + 2 * kk) * WBLK + (hb & 1) * 8192;
    char* dst = smem + (s % NSLOT) * WBLK + wid * 1024;
    glds16(src, dst);
    glds16(src + WBLK, dst + 8192);
  };
#pragma unroll
  for (int s = 0; s < PDIST; ++s) issue(s);
  int off4[4];
#pragma unroll
  for (int cc = 0; cc < 4; ++cc) off4[cc] = swz128(l31, 2 * cc + lh);
  u16* crow = p.out + row * (size_t)p.n_out + (size_t)NPASS * slice * COLS;
  f32x16 acc[2];
#pragma unroll
  for (int s = 0; s < NSTAGE; ++s) {
    const int pass = s / SPP, kk = s % SPP;
    if (kk == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    }
    {
      // behind stage s in the queue (loads and stores retire in issue order): the stages requested after it and the stores of every pass
      // that ended since its request (iterations s - PDIST .. s - 1)
      int allow = 2 * min(PDIST - 1, NSTAGE - 1 - s);
#pragma unroll
      for (int e = s - PDIST; e <= s - 1; ++e)
        if (e >= 0 && e % SPP == SPP - 1) allow += NST;
      wait_vm_any(allow);
    }
    KD_BARRIER();
    if (s + PDIST < NSTAGE) issue(s + PDIST);
    const char* st = smem + (s % NSLOT) * WBLK;
    bf16x8 wf[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) wf[0][j] = *reinterpret_cast<const bf16x8*>(st + j * 32 * 128 + off4[0]);
#pragma unroll
    for (int c8 = 0; c8 < 8; ++c8) {
      if (c8 + 1 < 8) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
          wf[(c8 + 1) & 1][j] = *reinterpret_cast<const bf16x8*>(st + ((c8 + 1) >> 2) * 8192 + j * 32 * 128 + off4[(c8 + 1) & 3]);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[c8 & 1][j], a[4 * (2 * kk + (c8 >> 2)) + (c8 & 3)], acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kk == SPP - 1) {
      if (EPI == KD_EPI_GEGLU) {                        // value block acc[0], gate block acc[1] (gemm_astat_kernel's epilogue)
        float v[16];
        const float rsh = 0.5f * rs;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2 o = geglu_pair(f32x2{acc[0][r], acc[0][r + 1]} * rsh, f32x2{acc[1][r], acc[1][r + 1]} * rs);
          v[r] = o.x;
          v[r + 1] = o.y;
        }
        store_block_bf16(crow + 32 * pass, v, lh, true);
      } else {                                          // one 64-column vector of q, k or v: dims 0..31 in acc[0], 32..63 in acc[1]
        const int vec = NPASS * slice + pass;
        const int which = vec / p.n_heads, head = vec - which * p.n_heads;
        if (which < 2) {
          typedef float f32x8s __attribute__((ext_vector_type(8)));
          f32x8s fq;
          float qsc;
          asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dword %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                       : "=s"(fq), "=s"(qsc) : "s"(p.freq + head * 8), "s"(p.qk_scale + head) : "memory");
          float fr[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) fr[u] = pick_half(fq[u], fq[4 + u], 0u - (unsigned)lh);
          qk_prep_blocks(acc[0], acc[1], rs, sqrtf(qsc), p.eps, py, px, fr);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) { acc[0][r] *= rs; acc[1][r] *= rs; }
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          float v[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = acc[jj][r];
          store_block_bf16(crow + 64 * pass + 32 * jj, v, lh, true);
        }
      }
    }
  }
}

// ---- global core for T > 256: 128-key blocks double-buffered through LDS, online softmax ---------------------------------------
constexpr int GL_QW = 8, GL_KB = 128, GL_NTK = GL_KB / 32;
constexpr int GL_IMG = GL_KB * 128, GL_BUF = 2 * GL_IMG, GL_LDS = 2 * GL_BUF;

__global__ __launch_bounds__(GL_QW * 64) void attn_long_bf16_kernel(const DArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, h2 = lane >> 5;
  const auto warm = code_warm_begin<6144>((int)blockIdx.x < a.warm && tid < 64);
  const int T = a.T, nqb = (T + GL_QW * 32 - 1) / (GL_QW * 32);
  int r = blockIdx.x;
  const int qb = r % nqb; r /= nqb;
  const int head = r % a.nh, b = r / a.nh;
  const size_t row_bytes = (size_t)3 * a.nh * DH * 2;
  const char* base = reinterpret_cast<const char*>(a.qkv) + (size_t)b * T * row_bytes + head * (DH * 2);
  const int q_slot = qb * (GL_QW * 32) + wid * 32 + l31;
  const bool q_ok = q_slot < T;
  const int q_tok = min(q_slot, T - 1);
  // block kb -> buffer kb & 1: this wave moves pieces 2 wid, 2 wid + 1 of the K image and of the V image (4 per block)
  auto issue = [&](int kb) {
    char* buf = smem + (kb & 1) * GL_BUF;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int pc = 2 * wid + j, row = 8 * pc + (lane >> 3);
      const int tok = min(kb * GL_KB + row, T - 1);
      const int q = (lane & 7) ^ asw(row);
      const char* src = base + (size_t)tok * row_bytes + q * 16;
      glds16(src + a.nh * DH * 2, buf + pc * 1024);
      glds16(src + 2 * a.nh * DH * 2, buf + GL_IMG + pc * 1024);
    }
  };
  bf16x8 qf[4];
  {
    const u32x4* qp = reinterpret_cast<const u32x4*>(base + (size_t)q_tok * row_bytes + 16 * h2);
#pragma unroll
    for (int st = 0; st < 4; ++st) qf[st] = __builtin_bit_cast(bf16x8, qp[2 * st]);
  }
  KD_WAIT_VM(0);                                   // q in registers before any block is in flight (counted waits below see only blocks)
  issue(0);
  f32x16 O[2];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int i = 0; i < 16; ++i) O[e][i] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const int ka = l31 * 128 + ((h2 ^ asw(l31)) << 4);
  const int va = vt_addr(4 * h2 + vt_lane_row(lane), lane);
  const int nkb = (T + GL_KB - 1) / GL_KB;
  code_warm_end(warm);
  for (int kb = 0; kb < nkb; ++kb) {
    if (kb + 1 < nkb) { issue(kb + 1); KD_WAIT_VM(4); } else { KD_WAIT_VM(0); }
    KD_BARRIER();                                  // block kb is in for every wave
    const char* Kimg = smem + (kb & 1) * GL_BUF;
    const char* Vimg = Kimg + GL_IMG;
    const int k0 = kb * GL_KB;
    f32x16 S[GL_NTK];
#pragma unroll
    for (int t = 0; t < GL_NTK; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) S[t][i] = 0.f;
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int t = 0; t < GL_NTK; ++t) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Kimg + (ka ^ (32 * st)) + 4096 * t);
        S[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[st], S[t], 0, 0, 0);
      }
    if (k0 + GL_KB > T) {
#pragma unroll
      for (int t = 0; t < GL_NTK; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) S[t][i] += (k0 + t * 32 + mfma32_row(i, lane) < T) ? 0.f : -INFINITY;
    }
    const float m_new = fmaxf(m_run, score_max<GL_NTK>(S));
    const float alpha = __expf(m_run - m_new);       // first block: exp(-inf) = 0
    m_run = m_new;
    l_run = l_run * alpha + score_exp<GL_NTK>(S, m_new);
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int i = 0; i < 16; ++i) O[e][i] *= alpha;
#pragma unroll
    for (int t = 0; t < GL_NTK; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u) pv_step(O, Vimg + (32 * t + 16 * u) * 128, va, p_frag(S[t], u));
    KD_BARRIER();                                  // every wave is done with buffer kb & 1 before block kb + 2 overwrites it
  }
  const float l = l_run + __shfl_xor(l_run, 32, 64);
  store_o(a.out + ((size_t)b * T + q_tok) * (a.nh * DH) + head * DH, O, 1.0f / l, h2, q_ok);
}

// ---- neighbourhood core ---------------------------------------------------------------------------------------------------------------
// One 256-thread workgroup per (sample, head, 8x16 query tile); KS = kernel size (3, 5, 7, 9).  The (8 + KS - 1) x (16 + KS - 1)
// key halo of the tile goes to LDS once (K image + V image).  Wave (wy, wx) owns the 4x8 query block at rows 4wy.., columns
// 8wx..: the clamped windows of its queries lie inside a PR x 16-or-32-column patch of the halo (PR = 4 + KS - 1 rows), walked
// as local keys kl = PW * r + c (PW = 16 for KS <= 9, else 32) in tiles of 32; keys outside a query's window get a -inf bias
// from per-lane bit words (one word per tile).  Out-of-image halo positions (images smaller than the halo) are clamped to a real
// token: they are outside every window, so their probability is exactly 0.
struct NArgs {
  const u16* qkv; u16* out;
  int batch, H, W, nh;
  int warm;                  // code warm-up workgroups (kd_common.h)
};
constexpr int NA_TH = 8, NA_TW = 16;

template <int KS>
struct NaGeo {
  static constexpr int HR = NA_TH + KS - 1, HC = NA_TW + KS - 1;            // halo
  static constexpr int PR = 4 + KS - 1;                                      // patch rows of a wave
  static constexpr int PW = (8 + KS - 1 <= 16) ? 16 : 32;                    // patch width (keys per patch row)
  static constexpr int NKT = (PR * PW + 31) / 32;                            // key tiles per wave
  // image rows: a wave's patch may poke past the halo's last key -- the highest row any fragment read touches is
  // (HR - 1) HC + (HC - (8 + KS - 1)) + 15 = HR HC - KS + 8.  Kept tight: at KS = 7 the two images take 78 KiB, TWO workgroups per CU
  static constexpr int ROWS = ((HR * HC - KS + 9 + 7) / 8) * 8;
  static constexpr int LDS = 2 * ROWS * 128;
};

template <int KS>
__global__ __launch_bounds__(256, (NaGeo<KS>::LDS <= 80 * 1024) ? 2 : 1) void attn_na2d_bf16_kernel(const NArgs a) {
  using G = NaGeo<KS>;
  constexpr int HR = G::HR, HC = G::HC, PR = G::PR, PW = G::PW, NKT = G::NKT, ROWS = G::ROWS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Kimg = smem;
  char* Vimg = smem + ROWS * 128;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, h2 = lane >> 5;
  const auto warm = code_warm_begin<7168>((int)blockIdx.x < a.warm && tid < 64);
  const int wy_ = wid >> 1, wx_ = wid & 1;
  const int tiles_x = (a.W + NA_TW - 1) / NA_TW, tiles_y = (a.H + NA_TH - 1) / NA_TH;
  int r;
  {   // XCD-aware tile order: neighbouring tiles (overlapping halos) run on ONE L2
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int q = nwg >> 3, rem = nwg & 7;
    r = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + k;
  }
  const int tx = r % tiles_x; r /= tiles_x;
  const int ty = r % tiles_y; r /= tiles_y;
  const int head = r % a.nh, b = r / a.nh;
  const int T = a.H * a.W;
  const size_t row_bytes = (size_t)3 * a.nh * DH * 2;
  const char* base = reinterpret_cast<const char*>(a.qkv) + (size_t)b * T * row_bytes + head * (DH * 2);
  const int ty0 = ty * NA_TH, tx0 = tx * NA_TW;
  const int hy0 = max(0, min(ty0 - KS / 2, a.H - HR)), hx0 = max(0, min(tx0 - KS / 2, a.W - HC));

  // ---- halo rows -> K / V images: image row 8 pc + (lane >> 3) = halo position (y, x), walked 32 rows at a time ----------------
  static_assert(PW == 16, "kernel sizes up to 9: 16-key patch rows");
  {
    int row = 8 * wid + (lane >> 3);
    int y = row / HC, x = row % HC;
    constexpr int DY = 32 / HC, DX = 32 % HC;
    for (int pc = wid; pc < ROWS / 8; pc += 4) {
      // rows past the halo's last key (a patch may poke there) take any real token: they are outside every window
      const int ky = min(hy0 + y, a.H - 1), kx = min(hx0 + x, a.W - 1);
      const char* src = base + (size_t)(unsigned)((ky * a.W + kx) * (int)row_bytes + (((lane & 7) ^ asw(row)) << 4));
      glds16(src + a.nh * DH * 2, Kimg + pc * 1024);
      glds16(src + 2 * a.nh * DH * 2, Vimg + pc * 1024);
      row += 32; x += DX; y += DY;
      if (x >= HC) { x -= HC; ++y; }
    }
  }
  // ---- this lane's query ------------------------------------------------------------------------------------------------------------
  const int qy_raw = ty0 + 4 * wy_ + (l31 >> 3), qx_raw = tx0 + 8 * wx_ + (l31 & 7);
  const bool q_ok = qy_raw < a.H && qx_raw < a.W;
  const int qy = min(qy_raw, a.H - 1), qx = min(qx_raw, a.W - 1);
  const int q_tok = qy * a.W + qx;
  bf16x8 qf[4];
  {
    const u32x4* qp = reinterpret_cast<const u32x4*>(base + (size_t)q_tok * row_bytes + 16 * h2);
#pragma unroll
    for (int st = 0; st < 4; ++st) qf[st] = __builtin_bit_cast(bf16x8, qp[2 * st]);
  }
  // clamped window start (NATTEN: start = clamp(i - KS/2, 0, L - KS)) relative to the halo; patch origin of this wave
  const int wy = max(0, min(qy - KS / 2, a.H - KS)) - hy0, wx = max(0, min(qx - KS / 2, a.W - KS)) - hx0;
  const int row_lo = min(max(0, min(min(ty0 + 4 * wy_, a.H - 1) - KS / 2, a.H - KS)) - hy0, HR - PR);
  const int col_lo = min(max(0, min(min(tx0 + 8 * wx_, a.W - 1) - KS / 2, a.W - KS)) - hx0, HC - (8 + KS - 1));
  const int korg = row_lo * HC + col_lo;          // halo index of patch key (0, 0); local key 16 r + c is image row korg + HC r + c
  // validity of patch column / patch row for THIS lane's query as +inf (inside the window) / -inf: v_min3 applies both at once.
  // Accumulator register i of a tile holds local key (i & 3) + 8 (i >> 2) + 4 h2: column (i & 3) + 8 ((i >> 2) & 1) + 4 h2 of patch
  // row 2 t + (i >> 3).
  float colv[8], rowv[2 * NKT];
  {
    const int r0 = wy - row_lo, c0 = wx - col_lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) colv[j] = ((unsigned)((j & 3) + 8 * (j >> 2) + 4 * h2 - c0) < (unsigned)KS) ? INFINITY : -INFINITY;
#pragma unroll
    for (int p = 0; p < 2 * NKT; ++p) rowv[p] = ((unsigned)(p - r0) < (unsigned)KS) ? INFINITY : -INFINITY;
  }
  KD_WAIT_VM(0);
  code_warm_end(warm);
  KD_BARRIER();

  // ---- S^T = K Q^T over the wave's key tiles: tile t, local key 32 t + i = patch row 2 t + (i >> 4), column i & 15 -------------
  f32x16 S[NKT];
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) S[t][i] = 0.f;
  int ka[NKT];
  {
    const int kr0 = korg + (l31 >> 4) * HC + (l31 & 15);
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
      const int kr = kr0 + 2 * t * HC;
      ka[t] = kr * 128 + ((h2 ^ asw(kr)) << 4);
    }
  }
#pragma unroll
  for (int st = 0; st < 4; ++st) {
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Kimg + (ka[t] ^ (32 * st)));
      S[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[st], S[t], 0, 0, 0);
    }
  }
  // ---- window mask + softmax -------------------------------------------------------------------------------------------------------
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) S[t][i] = min3f(S[t][i], colv[i & 7], rowv[2 * t + (i >> 3)]);
  const float m = score_max<NKT>(S);
  float l = score_exp<NKT>(S, m);
  l += __shfl_xor(l, 32, 64);

  // ---- O^T = V^T P^T: k-slots of lane-half h2 at step (t, u) are patch row 2 t + u, columns 4 h2 + {0..3} and + 8 ----------------
  f32x16 O[2];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int i = 0; i < 16; ++i) O[e][i] = 0.f;
  const int vr0 = korg + 4 * h2 + vt_lane_row(lane);
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) pv_step(O, Vimg, vt_addr(vr0 + (2 * t + u) * HC, lane), p_frag(S[t], u));
  store_o(a.out + ((size_t)b * T + q_tok) * (a.nh * DH) + head * DH, O, 1.0f / l, h2, q_ok);
}

// ---- neighbourhood core, kernel sizes 11 and 13 ------------------------------------------------------------------------------------
// The 4x8 query block of a wave now needs a patch of (4 + KS - 1) rows x (8 + KS - 1) = 18 / 20 columns: no longer a power-of-two
// width, so the patch is walked DENSELY: local key kl = 20 r + c (columns padded to 20, a multiple of 4 so that the 4-key groups of
// the V^T reads never straddle two patch rows), tiles of 32 keys, every key's (row, column) by constant division, validity per
// accumulator register from those.  One workgroup per CU (122 / 145 KiB of halo images), the whole register file per wave.
// Correctness form for the sizes no shipped config uses; the 3..9 kernel above is the tuned one.
template <int KS>
struct NaWide {
  static constexpr int HR = NA_TH + KS - 1, HC = NA_TW + KS - 1;
  static constexpr int PR = 4 + KS - 1, PC = 20;                            // patch rows; padded patch width
  static constexpr int NKT = (PR * PC + 31) / 32;
  static constexpr int ROWS = ((HR * HC + 2 * PC + 7) / 8) * 8;             // slack: padded columns and the last tile's tail poke past the halo
  static constexpr int LDS = 2 * ROWS * 128;
  static_assert(8 + KS - 1 <= PC, "patch width");
};

template <int KS>
__global__ __launch_bounds__(256, 1) void attn_na2d_wide_bf16_kernel(const NArgs a) {
  using G = NaWide<KS>;
  constexpr int HR = G::HR, HC = G::HC, PR = G::PR, PC = G::PC, NKT = G::NKT, ROWS = G::ROWS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Kimg = smem;
  char* Vimg = smem + ROWS * 128;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, h2 = lane >> 5;
  const auto warm = code_warm_begin<24576>((int)blockIdx.x < a.warm && tid < 64);
  const int wy_ = wid >> 1, wx_ = wid & 1;
  const int tiles_x = (a.W + NA_TW - 1) / NA_TW, tiles_y = (a.H + NA_TH - 1) / NA_TH;
  int r = blockIdx.x;
  const int tx = r % tiles_x; r /= tiles_x;
  const int ty = r % tiles_y; r /= tiles_y;
  const int head = r % a.nh, b = r / a.nh;
  const int T = a.H * a.W;
  const size_t row_bytes = (size_t)3 * a.nh * DH * 2;
  const char* base = reinterpret_cast<const char*>(a.qkv) + (size_t)b * T * row_bytes + head * (DH * 2);
  const int ty0 = ty * NA_TH, tx0 = tx * NA_TW;
  const int hy0 = max(0, min(ty0 - KS / 2, a.H - HR)), hx0 = max(0, min(tx0 - KS / 2, a.W - HC));
  for (int pc = wid; pc < ROWS / 8; pc += 4) {
    const int row = 8 * pc + (lane >> 3);
    const int ky = min(hy0 + row / HC, a.H - 1), kx = min(hx0 + row % HC, a.W - 1);       // rows past the halo: any real token (masked)
    const char* src = base + (size_t)(ky * a.W + kx) * row_bytes + (((lane & 7) ^ asw(row)) << 4);
    glds16(src + a.nh * DH * 2, Kimg + pc * 1024);
    glds16(src + 2 * a.nh * DH * 2, Vimg + pc * 1024);
  }
  const int qy_raw = ty0 + 4 * wy_ + (l31 >> 3), qx_raw = tx0 + 8 * wx_ + (l31 & 7);
  const bool q_ok = qy_raw < a.H && qx_raw < a.W;
  const int qy = min(qy_raw, a.H - 1), qx = min(qx_raw, a.W - 1);
  const int q_tok = qy * a.W + qx;
  bf16x8 qf[4];
  {
    const u32x4* qp = reinterpret_cast<const u32x4*>(base + (size_t)q_tok * row_bytes + 16 * h2);
#pragma unroll
    for (int st = 0; st < 4; ++st) qf[st] = __builtin_bit_cast(bf16x8, qp[2 * st]);
  }
  const int wy = max(0, min(qy - KS / 2, a.H - KS)) - hy0, wx = max(0, min(qx - KS / 2, a.W - KS)) - hx0;
  const int row_lo = min(max(0, min(min(ty0 + 4 * wy_, a.H - 1) - KS / 2, a.H - KS)) - hy0, HR - PR);
  const int col_lo = min(max(0, min(min(tx0 + 8 * wx_, a.W - 1) - KS / 2, a.W - KS)) - hx0, HC - (8 + KS - 1));
  const int korg = row_lo * HC + col_lo;
  const int r0 = wy - row_lo, c0 = wx - col_lo;
  auto img_row = [&](int kl) -> int { return korg + (kl / PC) * HC + (kl % PC); };     // LDS row of local key kl
  KD_WAIT_VM(0);
  code_warm_end(warm);
  KD_BARRIER();

  f32x16 S[NKT];
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) S[t][i] = 0.f;
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    const int kr = min(img_row(32 * t + l31), ROWS - 1);
    const int ka = kr * 128 + ((h2 ^ asw(kr)) << 4);
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Kimg + (ka ^ (32 * st)));
      S[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[st], S[t], 0, 0, 0);
    }
  }
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int kl = 32 * t + (i & 3) + 8 * (i >> 2) + 4 * h2;
      const int pr = kl / PC, pcn = kl % PC;
      const bool valid = (unsigned)(pr - r0) < (unsigned)KS && (unsigned)(pcn - c0) < (unsigned)KS && pr < PR;
      S[t][i] = valid ? S[t][i] : -INFINITY;
    }
  const float m = score_max<NKT>(S);
  float l = score_exp<NKT>(S, m);
  l += __shfl_xor(l, 32, 64);

  f32x16 O[2];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int i = 0; i < 16; ++i) O[e][i] = 0.f;
  const int c = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1), sub = (lane & 1) * 8;
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const bf16x8 pf = p_frag(S[t], u);
      // k-slots of lane-half h2: keys 32 t + 16 u + 4 h2 + {0..3} and the same + 8 -- two 4-key groups, each inside one patch row
      const int k0 = 32 * t + 16 * u + 4 * h2;
      const int ra = min(img_row(k0) + vt_lane_row(lane), ROWS - 1), rb = min(img_row(k0 + 8) + vt_lane_row(lane), ROWS - 1);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (s16x4 __attribute__((address_space(3)))*)(Vimg + ra * 128 + (((4 * e + c) ^ asw(ra)) << 4) + sub));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (s16x4 __attribute__((address_space(3)))*)(Vimg + rb * 128 + (((4 * e + c) ^ asw(rb)) << 4) + sub));
        const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2v = __builtin_bit_cast(u32x2, hi);
        O[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, u32x4{l2[0], l2[1], h2v[0], h2v[1]}), pf, O[e], 0, 0, 0);
      }
    }
  store_o(a.out + ((size_t)b * T + q_tok) * (a.nh * DH) + head * DH, O, 1.0f / l, h2, q_ok);
}

template <int KS>
static int launch_na_wide(const NArgs& a, hipStream_t s) {
  auto k = attn_na2d_wide_bf16_kernel<KS>;
  static LdsAttr set;
  set.ensure(reinterpret_cast<const void*>(k), NaWide<KS>::LDS);
  const long nb = (long)a.batch * a.nh * ((a.H + NA_TH - 1) / NA_TH) * ((a.W + NA_TW - 1) / NA_TW);
  char nm[64] = "attn_na2d_bf16";
  if (prof_on()) snprintf(nm, sizeof(nm), "attn_na2d_bf16 k%d %dx%d nh=%d", KS, a.H, a.W, a.nh);
  LaunchScope prof(nm, 4.0 * a.batch * (double)a.H * a.W * a.nh * DH * KS * KS, 8.0 * a.batch * (double)a.H * a.W * a.nh * DH, s);
  hipLaunchKernelGGL(k, dim3((unsigned)nb), dim3(256), NaWide<KS>::LDS, s, a);
  return check_launch("kd_attn_na2d_bf16");
}

template <int MODE, int NT, int QW>
static int launch_dense(const DArgs& a, long nproblems, const char* name, hipStream_t s) {
  constexpr int LDS = NT * 32 * 256, NQB = (NT + QW - 1) / QW;
  auto k = attn_dense_bf16_kernel<MODE, NT, QW>;
  static LdsAttr set;
  set.ensure(reinterpret_cast<const void*>(k), LDS);
  const int n_slots = MODE == MODE_GLOBAL ? a.T : (1 << (2 * WinLog2<MODE>::v));
  LaunchScope prof(name, 4.0 * (double)nproblems * n_slots * n_slots * DH, 2.0 * (double)a.batch * a.T * a.nh * DH * 4.0, s);
  hipLaunchKernelGGL(k, dim3((unsigned)(nproblems * NQB)), dim3(QW * 64), LDS, s, a);
  return check_launch(name);
}

template <int KS>
static int launch_na(const NArgs& a, hipStream_t s) {
  auto k = attn_na2d_bf16_kernel<KS>;
  static LdsAttr set;
  set.ensure(reinterpret_cast<const void*>(k), NaGeo<KS>::LDS);
  const long nb = (long)a.batch * a.nh * ((a.H + NA_TH - 1) / NA_TH) * ((a.W + NA_TW - 1) / NA_TW);
  char nm[64] = "attn_na2d_bf16";
  if (prof_on()) snprintf(nm, sizeof(nm), "attn_na2d_bf16 k%d %dx%d nh=%d", KS, a.H, a.W, a.nh);
  LaunchScope prof(nm, 4.0 * a.batch * (double)a.H * a.W * a.nh * DH * KS * KS, 8.0 * a.batch * (double)a.H * a.W * a.nh * DH, s);
  hipLaunchKernelGGL(k, dim3((unsigned)nb), dim3(256), NaGeo<KS>::LDS, s, a);
  return check_launch("kd_attn_na2d_bf16");
}

}  // namespace b16
}  // namespace kd

using namespace kd;
using namespace kd::b16;

extern "C" int kd_attn_global_bf16(const void* qkv, void* out, int batch, int T, int nh, void* stream) {
  if (!qkv || !out || batch <= 0 || nh <= 0 || T <= 0) return fail(KD_EINVAL, "kd_attn_global_bf16: bad arguments");
  DArgs a{reinterpret_cast<const u16*>(qkv), reinterpret_cast<u16*>(out), batch, T, nh, 0, 0, 0, 0, option("code_warm", KD_CODE_WARM_DEFAULT)};
  hipStream_t s = (hipStream_t)stream;
  const long nb = (long)batch * nh;
  if (T > 256) {
    static LdsAttr set;
    set.ensure(reinterpret_cast<const void*>(attn_long_bf16_kernel), GL_LDS);
    const long nqb = (T + GL_QW * 32 - 1) / (GL_QW * 32);
    LaunchScope prof("attn_global_bf16", 4.0 * (double)nb * T * T * DH, 2.0 * (double)batch * T * nh * DH * 4.0, s);
    hipLaunchKernelGGL(attn_long_bf16_kernel, dim3((unsigned)(nb * nqb)), dim3(GL_QW * 64), GL_LDS, s, a);
    return check_launch("kd_attn_global_bf16");
  }
  const int qw = option("attn_global_qw", 8);
  if (T <= 32) return launch_dense<MODE_GLOBAL, 1, 1>(a, nb, "attn_global_bf16", s);
  if (T <= 64) return launch_dense<MODE_GLOBAL, 2, 2>(a, nb, "attn_global_bf16", s);
  if (T <= 128) return launch_dense<MODE_GLOBAL, 4, 4>(a, nb, "attn_global_bf16", s);
  if (qw >= 8) return launch_dense<MODE_GLOBAL, 8, 8>(a, nb, "attn_global_bf16", s);
  if (qw == 2) return launch_dense<MODE_GLOBAL, 8, 2>(a, nb, "attn_global_bf16", s);
  return launch_dense<MODE_GLOBAL, 8, 4>(a, nb, "attn_global_bf16", s);
}

// The global-attention block in front of its out projection as ONE launch (attn_block_bf16_kernel above).  `d` is the descriptor of the
// block's qkv projection exactly as kd_gemm_bf16 takes it (A = the residual stream, Wp = the packed qkv weight, scale / scale_stride /
// rows_per_sample = the AdaRMSNorm scale table, qk_scale / rope_pos / rope_freq / n_heads) -- except that C receives the ATTENTION OUTPUT
// [M, n_heads * 64] bf16 instead of qkv.  Shapes: 256 tokens per sample, K = n_heads * 64 in {256, 512}, N = 3 K.
extern "C" int kd_attn_block_bf16_supported(int tokens_per_sample, int width, int n_heads) {
  return tokens_per_sample == 256 && (width == 256 || width == 512) && n_heads * 64 == width && option("attn_block_bf16", 1) ? 1 : 0;
}

extern "C" int kd_attn_block_bf16(const KdGemm* dp, const KdGemm* op, int* sync, void* stream) {
  if (!dp) return fail(KD_EINVAL, "kd_attn_block_bf16: null descriptor");
  const KdGemm& d = *dp;
  if (!d.A || !d.Wp || !d.C || !d.scale || !d.qk_scale || !d.rope_pos || !d.rope_freq) return fail(KD_EINVAL, "kd_attn_block_bf16: null operand");
  if (d.epi != KD_EPI_QKV || !d.norm || d.a_mode != KD_A_PLAIN || d.precision != KD_PREC_BF16)
    return fail(KD_EINVAL, "kd_attn_block_bf16: the descriptor must be a bf16 norm -> qkv projection");
  if (!kd_attn_block_bf16_supported(d.rows_per_sample, d.K, d.n_heads) || d.N != 3 * d.K || d.M <= 0 || d.M % 256)
    return fail(KD_EINVAL, "kd_attn_block_bf16: shape M=%d N=%d K=%d, %d tokens per sample, %d heads is not taken (256 tokens per sample, "
                "K = 64 heads in {256, 512})", d.M, d.N, d.K, d.rows_per_sample, d.n_heads);
  if (op) {
    const KdGemm& o = *op;
    if (o.epi != KD_EPI_RESIDUAL || o.norm || o.a_mode != KD_A_PLAIN || o.precision != KD_PREC_BF16 || !o.Wp || o.M != d.M || o.N != d.K || o.K != d.K)
      return fail(KD_EINVAL, "kd_attn_block_bf16: the second descriptor must be the block's bf16 out projection + residual (M=%d, N=K=%d)", d.M, d.K);
    if (o.A != d.C || o.C != d.A || o.R != d.A)
      return fail(KD_EINVAL, "kd_attn_block_bf16: the out projection must read the attention output and update the residual stream in place");
    if (!sync) return fail(KD_EINVAL, "kd_attn_block_bf16: the fused out projection needs the counter workspace (2 * batch + 1 ints, zeroed once)");
  }
  BArgs a{reinterpret_cast<const u16*>(d.A), reinterpret_cast<const char*>(d.Wp), reinterpret_cast<u16*>(d.C), d.scale, d.scale_stride, d.eps,
          d.M / 256, d.n_heads, d.qk_scale, d.rope_pos, d.rope_freq, option("code_warm", KD_CODE_WARM_DEFAULT), g_clk,
          op ? reinterpret_cast<const char*>(op->Wp) : nullptr, op ? reinterpret_cast<u16*>(op->C) : nullptr, sync};
  hipStream_t s = (hipStream_t)stream;
  constexpr int LDS = 8 * WBLK + 8 * 512 * 4;          // prologue: 8 wave-private staging slots + 8 scale vectors; later ring + K / V images
  const double flops = 2.0 * d.M * 3.0 * d.K * d.K + 4.0 * (double)a.batch * a.nh * 256.0 * 256.0 * DH + (op ? 2.0 * d.M * (double)d.K * d.K : 0.0);
  const double bytes = 2.0 * ((double)d.M * d.K * 2.0 + 3.0 * d.K * d.K) + (op ? 2.0 * ((double)d.M * d.K + (double)d.K * d.K) : 0.0);
  char nm[96] = "attn_block_bf16";
  if (prof_on()) snprintf(nm, sizeof(nm), "attn_block_bf16%s M=%d K=%d nh=%d", op ? "+out" : "", d.M, d.K, d.n_heads);
  LaunchScope prof(nm, flops, bytes, s);
#define KD_BLK(NCV, OV, TSV) { static LdsAttr set; set.ensure(reinterpret_cast<const void*>(attn_block_bf16_kernel<NCV, OV, TSV>), LDS); \
    hipLaunchKernelGGL((attn_block_bf16_kernel<NCV, OV, TSV>), dim3((unsigned)(a.batch * a.nh)), dim3(512), LDS, s, a); }
#define KD_BLK2(NCV) { if (op) { if (a.clk) KD_BLK(NCV, true, true) else KD_BLK(NCV, true, false) } \
                       else { if (a.clk) KD_BLK(NCV, false, true) else KD_BLK(NCV, false, false) } }
  if (d.K == 512) KD_BLK2(32) else KD_BLK2(16)
#undef KD_BLK2
#undef KD_BLK
  return check_launch("kd_attn_block_bf16");
}

// AdaRMSNorm -> wide projection in the attention block's form (proj_block_bf16_kernel above).  `d` is the projection's descriptor exactly as
// kd_gemm_bf16 takes it -- epi = KD_EPI_GEGLU (N = d_ff) or KD_EPI_QKV (N = 3 K; the qkv tensor is written, for the levels whose attention core
// is its own launch), norm = 1 -- and the results are bit-identical to that call.  Rows per sample a multiple of 256, K in {256, 512}, the W rows
// a multiple of 6 half blocks (d_ff % 192 == 0; 3 K % 384 == 0 holds for both widths).
extern "C" int kd_proj_block_bf16_supported(int tokens_per_sample, int width, int n, int epi) {
  if (tokens_per_sample <= 0 || tokens_per_sample % 256 || (width != 256 && width != 512) || !option("proj_block_bf16", 1)) return 0;
  if (epi == KD_EPI_GEGLU) return n > 0 && n % 192 == 0;
  if (epi == KD_EPI_QKV) return n == 3 * width;
  return 0;
}

extern "C" int kd_proj_block_bf16(const KdGemm* dp, void* stream) {
  if (!dp) return fail(KD_EINVAL, "kd_proj_block_bf16: null descriptor");
  const KdGemm& d = *dp;
  if (!d.A || !d.Wp || !d.C || !d.scale) return fail(KD_EINVAL, "kd_proj_block_bf16: null operand");
  if ((d.epi != KD_EPI_GEGLU && d.epi != KD_EPI_QKV) || !d.norm || d.a_mode != KD_A_PLAIN || d.precision != KD_PREC_BF16)
    return fail(KD_EINVAL, "kd_proj_block_bf16: the descriptor must be a bf16 norm -> GEGLU or norm -> qkv projection");
  if (d.epi == KD_EPI_QKV && (!d.qk_scale || !d.rope_pos || !d.rope_freq || d.n_heads * 64 != d.K))
    return fail(KD_EINVAL, "kd_proj_block_bf16: the qkv projection needs qk_scale, rope_pos, rope_freq and n_heads * 64 == K");
  if (!kd_proj_block_bf16_supported(d.rows_per_sample, d.K, d.N, d.epi) || d.M <= 0 || d.M % d.rows_per_sample)
    return fail(KD_EINVAL, "kd_proj_block_bf16: shape M=%d N=%d K=%d, %d tokens per sample is not taken (tokens per sample a multiple of 256, K in "
                "{256, 512}, d_ff a multiple of 192)", d.M, d.N, d.K, d.rows_per_sample);
  const int w_rows = d.epi == KD_EPI_GEGLU ? 2 * d.N : d.N;
  UArgs a{reinterpret_cast<const u16*>(d.A), reinterpret_cast<const char*>(d.Wp), reinterpret_cast<u16*>(d.C), d.scale, d.scale_stride, d.eps,
          d.M / 256, d.rows_per_sample / 256, w_rows / 384, d.N, d.n_heads, d.qk_scale, d.rope_pos, d.rope_freq, option("code_warm", KD_CODE_WARM_DEFAULT)};
  hipStream_t s = (hipStream_t)stream;
  constexpr int LDS = 8 * WBLK + 8 * 512 * 4;
  const double flops = 2.0 * d.M * (double)w_rows * d.K;
  const double bytes = 2.0 * ((double)d.M * d.K + (double)w_rows * d.K + (double)d.M * d.N);
  char nm[96] = "proj_block_bf16";
  if (prof_on()) snprintf(nm, sizeof(nm), "proj_block_bf16<e%d> M=%d N=%d K=%d", d.epi, d.M, d.N, d.K);
  LaunchScope prof(nm, flops, bytes, s);
#define KD_PB(NCV, EP) { static LdsAttr set; set.ensure(reinterpret_cast<const void*>(proj_block_bf16_kernel<NCV, EP>), LDS); \
    hipLaunchKernelGGL((proj_block_bf16_kernel<NCV, EP>), dim3((unsigned)(a.groups * a.slices)), dim3(512), LDS, s, a); }
  if (d.K == 512) { if (d.epi == KD_EPI_GEGLU) KD_PB(32, KD_EPI_GEGLU) else KD_PB(32, KD_EPI_QKV) }
  else { if (d.epi == KD_EPI_GEGLU) KD_PB(16, KD_EPI_GEGLU) else KD_PB(16, KD_EPI_QKV) }
#undef KD_PB
  return check_launch("kd_proj_block_bf16");
}

extern "C" int kd_attn_window_bf16(const void* qkv, void* out, int batch, int H, int W, int nh, int ws, int shift, void* stream) {
  if (!qkv || !out || batch <= 0 || nh <= 0 || H <= 0 || W <= 0) return fail(KD_EINVAL, "kd_attn_window_bf16: bad arguments");
  if (ws != 4 && ws != 8 && ws != 16) return fail(KD_EINVAL, "kd_attn_window_bf16: window_size %d unsupported (4, 8 or 16)", ws);
  if ((H % ws) || (W % ws)) return fail(KD_EINVAL, "kd_attn_window_bf16: grid %dx%d not divisible by the window", H, W);
  if (shift < 0 || shift >= ws) return fail(KD_EINVAL, "kd_attn_window_bf16: bad shift %d", shift);
  DArgs a{reinterpret_cast<const u16*>(qkv), reinterpret_cast<u16*>(out), batch, H * W, nh, H, W, ws, shift, option("code_warm", KD_CODE_WARM_DEFAULT)};
  const long nb = (long)batch * nh * (H / ws) * (W / ws);
  hipStream_t s = (hipStream_t)stream;
  if (ws == 8) return launch_dense<MODE_WINDOW, 2, 2>(a, nb, "attn_window_bf16", s);
  if (ws == 4) return launch_dense<MODE_WINDOW4, 1, 1>(a, nb, "attn_window_bf16", s);
  return launch_dense<MODE_WINDOW16, 8, 4>(a, nb, "attn_window_bf16", s);
}

extern "C" int kd_attn_na2d_bf16(const void* qkv, void* out, int batch, int H, int W, int nh, int ks, void* stream) {
  if (!qkv || !out || batch <= 0 || nh <= 0) return fail(KD_EINVAL, "kd_attn_na2d_bf16: bad arguments");
  if (ks < 3 || ks > 13 || !(ks & 1)) return fail(KD_EINVAL, "kd_attn_na2d_bf16: kernel_size %d unsupported (3, 5, 7, 9, 11 or 13)", ks);
  if (H < ks || W < ks) return fail(KD_EINVAL, "kd_attn_na2d_bf16: grid %dx%d smaller than the %dx%d neighbourhood", H, W, ks, ks);
  NArgs a{reinterpret_cast<const u16*>(qkv), reinterpret_cast<u16*>(out), batch, H, W, nh, option("code_warm", KD_CODE_WARM_DEFAULT)};
  hipStream_t s = (hipStream_t)stream;
  switch (ks) {
    case 3: return launch_na<3>(a, s);
    case 5: return launch_na<5>(a, s);
    case 7: return launch_na<7>(a, s);
    case 9: return launch_na<9>(a, s);
    case 11: return launch_na_wide<11>(a, s);
    default: return launch_na_wide<13>(a, s);
  }
}

KD_TEXT_PAD(attn_bf16)      // last function of this code object: kd_common.h, code warm-up
