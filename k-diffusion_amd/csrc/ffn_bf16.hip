// Fused feed-forward block of the HDiT denoiser in bf16 mode, gfx950:
//     y = x + down_proj( GEGLU( up_proj( AdaRMSNorm(x, cond) ) ) )                      (image_transformer_v2.py:479-493, :95 GEGLU)
// in ONE kernel: the d_ff-wide hidden activation never goes to HBM.  At the level-0 shape of the headline config (131 072 tokens,
// width 128, d_ff 384) the two-kernel form moves 33 + 100 MB (up) and 100 + 33 + 33 MB (down) per layer and both kernels sit
// at the HBM rate; fused, the block reads x once and writes it once (67 MB) and is bound by its matrix / VALU work instead.
//
// Structure ("A-stationary", lane-owns-row like every bf16 GEMM here, see bf16_common.h):
//   * a workgroup of 8 waves owns a panel of 256 rows; wave w keeps the normalised rows 32 w .. 32 w + 31 as MFMA B-operand
//     fragments in registers for the whole kernel (K / 16 fragments) and the K-wide fp32 output accumulators beside them;
//   * d_ff is walked in tiles of 64 hidden features.  The weights of a tile -- the GEGLU-interleaved up-projection tile
//     [128 rows: 32 value | 32 gate | 32 value | 32 gate][K] and the down-projection k-step [K rows][64 hidden] -- form one
//     48 KiB unit (K = 128) that all waves copy HBM/L2 -> LDS with global_load_lds into a 3-slot ring: one barrier per tile,
//     counted vmcnt, the copy of tile t + 2 in flight while tile t is multiplied;
//   * per tile: 4 K/16 MFMAs give value / gate accumulators, the GEGLU runs in the lane that owns the row, and its 64 outputs
//     are ALREADY the B operand of the down-projection: accumulator registers 8u .. 8u+7 of a 32-feature block are the 8 k-slots
//     of lane-half lh for the hidden features {16u + 4 lh + 0..3, 16u + 8 + 4 lh + 0..3}.  The down-projection weight is packed
//     with that k order inside every group of 16 (kd_pack_weight_bf16 layout 2), so its fragments stay one ds_read_b128;
//   * epilogue: + x (from the raw row chunks kept in registers, one half-wave exchange per dword pair), bf16, 16-byte stores.
#include "bf16_common.h"

namespace kd {
namespace b16 {

struct FArgs {
  const u16* X; u16* Y; const char* Wu; const char* Wd;
  const float* scale; int scale_stride, rows_per_sample; float eps;
  int M, n_tiles;            // n_tiles = d_ff / 64
  unsigned long long* clk;   // kd_prof_clock_buffer: time line of workgroup 0
  int warm;                  // code warm-up workgroups (kd_common.h)
  const u16* Att; const char* Wo;   // fused out projection in front of the block (round 3, width 128): X <- X + Att Wo^T first; NULL = none
};

#define KD_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define KD_BARRIER() asm volatile("s_barrier" ::: "memory")

extern unsigned long long* g_clk;                     // gemm_bf16.hip (kd_prof_clock_buffer)
__device__ __forceinline__ void wait_vm_count(int n) {      // s_waitcnt vmcnt(n) for the few run-time values the half-unit ring needs
  if (n >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else if (n >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
constexpr int FF_NW = 8;

// SKEW: waves 4..7 (the second wave of every SIMD) run half a tile behind waves 0..3 -- in interval t they finish tile t - 1
// (GEGLU, down projection) and then start tile t (up projection), while their SIMD partner does up(t), GEGLU(t), down(t).  The
// barrier per tile otherwise keeps both waves of a SIMD in the SAME phase (both want the matrix pipe, then both want the VALU,
// and the per-tile costs add up); skewed, one wave's MFMAs run beside the other's GEGLU.  The down-projection k-step of tile t - 1
// has to stay one interval longer: its ring gets a fourth slot (3 x 32 KiB + 4 x 16 KiB = all 160 KiB of LDS).
// OUTP (round 3, as in ffn_x3.hip): the attention block's out projection runs first in the same workgroup (x <- x + att Wo^T,
// image_transformer_v2.py:473-476).  The lane's ATTENTION row is the B operand as it is stored (bf16), the fp32 output accumulators start
// from x (read into the C layout) and take the 32 MFMAs of Wo (32 KiB, parked in the ring slot the third weight unit will use); what they
// then hold is the new residual stream in the layout of an MFMA result, which under pack layout 3 of the up projection's weight IS its B
// operand: norm statistics, scale and rounding in registers, and the down projection keeps accumulating on top of the new x -- the out
// projection's result and the skip operand never cross HBM (and the new x reaches the norm in fp32, not rounded to bf16).
template <int NC /* K / 16 */, bool SKEW, bool OUTP = false>
__global__ __launch_bounds__(FF_NW * 64) void ffn_kernel(const FArgs p) {
  static_assert(!OUTP || (NC == 8 && !SKEW), "fused out projection: width 128, plain variant");
  constexpr int K = NC * 16, NKU = NC / 4, KB = K / 32, NTD = K / 128;
  constexpr int UP_BYTES = NKU * WBLK, DN_BYTES = NTD * WBLK, UNIT = UP_BYTES + DN_BYTES;
  constexpr int NDS = SKEW ? 4 : 3;                   // slots of the down-projection ring
  constexpr int PCS = UNIT / 1024 / FF_NW;            // 1 KiB pieces per wave per tile
  static_assert(K % 128 == 0 && (UNIT / 1024) % FF_NW == 0, "unit = whole pieces per wave");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wid = SKEW ? tid >> 6 : __builtin_amdgcn_readfirstlane(tid >> 6);     // (scalar branches on `late` cost the skewed variant 130 spilled registers)
  const auto warm = code_warm_begin<(SKEW ? 12 : 9) * 1024>((int)blockIdx.x < p.warm && tid < 64);     // kd_common.h
  const int row = blockIdx.x * (FF_NW * 32) + wid * 32 + l31;
  const bool ok = row < p.M;
  const int rowc = ok ? row : p.M - 1;
  const int T = p.n_tiles;
  const bool probe = p.clk && blockIdx.x == 0 && tid == 0;
  if (probe) { p.clk[0] = __builtin_amdgcn_s_memtime(); p.clk[1] = __builtin_amdgcn_s_memrealtime(); }

  const bool late = SKEW && wid >= FF_NW / 2;
  auto up_slot = [&](int t) -> char* { return smem + (t % 3) * UP_BYTES; };
  auto dn_slot = [&](int t) -> char* { return smem + 3 * UP_BYTES + (t % NDS) * DN_BYTES; };
  auto issue = [&](int t) {
#pragma unroll
    for (int i = 0; i < PCS; ++i) {
      const int pi = wid + FF_NW * i;
      const char* src;
      char* dst;
      if (pi < NKU * 16) {
        src = p.Wu + ((size_t)t * NKU) * WBLK + pi * 1024;
        dst = up_slot(t) + pi * 1024;
      } else {
        const int pd = pi - NKU * 16;
        src = p.Wd + ((size_t)(pd >> 4) * T + t) * WBLK + (pd & 15) * 1024;
        dst = dn_slot(t) + pd * 1024;
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane * 16),
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };

  // ---- this lane's row: raw bf16 + the sample's scale vector requested first, then the first two weight units ----------------
  u32x4 raw[NC];
  f32x4 s0[NC], s1[NC];                               // OUTP: the scale in the C layout (features 32 ob + 8 g + 4 lh .. + 3: s0 / s1[2 ob + (g >> 1)], g & 1)
  {
    const u32x4* ap = reinterpret_cast<const u32x4*>((OUTP ? p.Att : p.X) + (size_t)rowc * K + 8 * lh);
    const float* sp = p.scale + (size_t)(rowc / p.rows_per_sample) * p.scale_stride + (OUTP ? 4 : 8) * lh;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      raw[c] = ap[2 * c];
      s0[c] = *reinterpret_cast<const f32x4*>(sp + 16 * c);
      s1[c] = *reinterpret_cast<const f32x4*>(sp + 16 * c + (OUTP ? 8 : 4));
    }
  }
  f32x16 acc_o[K / 32];
  if constexpr (OUTP) {
    // x of the lane's row -> the output accumulators (C layout); requested BEFORE the weight copies so that the counted wait below covers it
    const u16* xrow = p.X + (size_t)rowc * K;
#pragma unroll
    for (int ob = 0; ob < K / 32; ++ob) {
      float sk[16];
      load_block_bf16(xrow + 32 * ob, sk, lh);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc_o[ob][r] = sk[r];
    }
    // Wo: two [128 rows][64 k] blocks -> the third up slot (free until unit 2 is requested)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.Wo + (wid + FF_NW * i) * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void*)(up_slot(2) + (wid + FF_NW * i) * 1024), 16, 0, 0);
  }
  issue(0);
  if (T > 1) issue(1);
  bf16x8 a[NC];
  float rs;
  if constexpr (OUTP) {
    if (T > 1) { KD_WAIT_VM(12); } else { KD_WAIT_VM(6); }      // rows, x and Wo are in (the two weight units may still be in flight)
    KD_BARRIER();
    int offo[4];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) offo[cc] = swz128(l31, 2 * cc + lh);
    const char* wo = up_slot(2);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const bf16x8 af = __builtin_bit_cast(bf16x8, raw[c]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wo + (c >> 2) * WBLK + offo[c & 3] + j * 32 * 128);
        acc_o[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, af, acc_o[j], 0, 0, 0);
      }
    }
    // new x (fp32, C layout) -> norm statistics, scale, bf16 fragments in the k order of pack layout 3: chunk 2 ob + hc = registers 8 hc .. + 7
    float ssq = 0.f;
#pragma unroll
    for (int ob = 0; ob < K / 32; ++ob) {
#pragma unroll
      for (int r = 0; r < 16; ++r) ssq = fmaf(acc_o[ob][r], acc_o[ob][r], ssq);
#pragma unroll
      for (int hc = 0; hc < 2; ++hc) {
        const f32x4 sa = s0[2 * ob + hc], sb = s1[2 * ob + hc];      // features 32 ob + 16 hc + 4 lh + 0..3, and + 8
        const f32x16& v = acc_o[ob];
        u32x4 o = {pack_bf16(v[8 * hc] * sa[0], v[8 * hc + 1] * sa[1]), pack_bf16(v[8 * hc + 2] * sa[2], v[8 * hc + 3] * sa[3]),
                   pack_bf16(v[8 * hc + 4] * sb[0], v[8 * hc + 5] * sb[1]), pack_bf16(v[8 * hc + 6] * sb[2], v[8 * hc + 7] * sb[3])};
        asm volatile("" : "+v"(o));
        a[2 * ob + hc] = __builtin_bit_cast(bf16x8, o);
      }
    }
    ssq += __shfl_xor(ssq, 32, 64);
    rs = rsqrtf(ssq / (float)K + p.eps);
  } else {
    float ssq = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      float x[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { x[2 * e] = bf_lo(raw[c][e]); x[2 * e + 1] = bf_hi(raw[c][e]); }
#pragma unroll
      for (int e = 0; e < 8; ++e) ssq = fmaf(x[e], x[e], ssq);
      u32x4 o = {pack_bf16(x[0] * s0[c][0], x[1] * s0[c][1]), pack_bf16(x[2] * s0[c][2], x[3] * s0[c][3]),
                 pack_bf16(x[4] * s1[c][0], x[5] * s1[c][1]), pack_bf16(x[6] * s1[c][2], x[7] * s1[c][3])};
      asm volatile("" : "+v"(o));
      a[c] = __builtin_bit_cast(bf16x8, o);
    }
    ssq += __shfl_xor(ssq, 32, 64);
    rs = rsqrtf(ssq / (float)K + p.eps);
  }
  const float rsh = 0.5f * rs;
  code_warm_end(warm);
  if (probe) p.clk[4] = __builtin_amdgcn_s_memtime();            // rows normalised

  int off4[4];
#pragma unroll
  for (int cc = 0; cc < 4; ++cc) off4[cc] = swz128(l31, 2 * cc + lh);
  if constexpr (!OUTP) {
#pragma unroll
    for (int ob = 0; ob < KB; ++ob)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc_o[ob][r] = 0.f;
  }

  f32x16 acc[4];
  bf16x8 hf[4];
  auto up = [&](int t) {                              // value / gate accumulators of the 2 x 32 hidden features of tile t
    const char* wu = up_slot(t);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const char* wk = wu + (c >> 2) * WBLK + off4[c & 3];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wk + j * 32 * 128);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, a[c], acc[j], 0, 0, 0);
      }
    }
  };
  auto glu = [&]() {                                  // GEGLU in the lane that owns the row; the outputs are the down projection's B operand
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      unsigned pk[8];
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 o = geglu_pair(f32x2{acc[2 * jj][r], acc[2 * jj][r + 1]} * rsh, f32x2{acc[2 * jj + 1][r], acc[2 * jj + 1][r + 1]} * rs);
        pk[r >> 1] = pack_bf16(o.x, o.y);
      }
      hf[2 * jj] = __builtin_bit_cast(bf16x8, u32x4{pk[0], pk[1], pk[2], pk[3]});
      hf[2 * jj + 1] = __builtin_bit_cast(bf16x8, u32x4{pk[4], pk[5], pk[6], pk[7]});
    }
  };
  auto down = [&](int t) {                            // k-step t (64 hidden features) into the K output features
    const char* wd = dn_slot(t);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int ob = 0; ob < KB; ++ob) {
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wd + (ob >> 2) * WBLK + (ob & 3) * 32 * 128 + off4[g]);
        acc_o[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, hf[g], acc_o[ob], 0, 0, 0);
      }
  };
  static_assert(PCS == 6, "the counted waits below are written for 6 pieces per wave per tile");
  const int n_int = SKEW ? T + 1 : T;                 // the late waves finish the last tile in one more interval
  for (int t = 0; t < n_int; ++t) {
    if (t + 1 < T) { KD_WAIT_VM(6); } else { KD_WAIT_VM(0); }
    KD_BARRIER();                                      // unit t is in for every wave; every wave is done with interval t - 1
    if (t + 2 < T) issue(t + 2);
    if (!late) {
      if (t < T) { up(t); glu(); down(t); }
    } else {
      if (t > 0) { glu(); down(t - 1); }
      if (t < T) up(t);
    }
  }
  if (probe) { p.clk[5] = __builtin_amdgcn_s_memtime(); p.clk[6] = p.clk[5]; }       // tiles done
  // ---- + skip, store -------------------------------------------------------------------------------------------------------------
  // The skip operand is the row this lane read in the prologue: its raw bf16 chunks are still in registers, only in the B-operand
  // order (chunk c: k = 16 c + 8 lh .. + 7) where the accumulators want the C-layout (features 8 g + 4 lh .. + 3 of a 32-block).
  // One half-wave exchange per dword pair puts them there -- lanes 0-31 keep their first 4 k of a chunk and get the partner's first 4,
  // lanes 32-63 get the partner's last 4 and keep their own -- instead of reading the row a second time (round 2 did: 94.8 MB per
  // launch for 67 MB of algorithmic traffic, half of the re-reads missed L2).
  u16* yrow = p.Y + (size_t)rowc * K;
  if constexpr (OUTP) {       // the accumulators started from the new x: nothing to add
#pragma unroll
    for (int ob = 0; ob < KB; ++ob) {
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = acc_o[ob][r];
      store_block_bf16(yrow + 32 * ob, v, lh, ok);
    }
  } else if (SKEW) {        // the skewed variant (on request only) has no registers to spare for the raw row: it reads it again
    const u16* xrow = p.X + (size_t)rowc * K;
#pragma unroll
    for (int ob = 0; ob < KB; ++ob) {
      float sk[16], v[16];
      load_block_bf16(xrow + 32 * ob, sk, lh);
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = acc_o[ob][r] + sk[r];
      store_block_bf16(yrow + 32 * ob, v, lh, ok);
    }
  } else
#pragma unroll
  for (int ob = 0; ob < KB; ++ob) {
    float v[16];
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      unsigned x0 = raw[2 * ob + cc][0], x1 = raw[2 * ob + cc][1], y0 = raw[2 * ob + cc][2], y1 = raw[2 * ob + cc][3];
      half_swap(x0, y0);
      half_swap(x1, y1);
      const float sk[8] = {bf_lo(x0), bf_hi(x0), bf_lo(x1), bf_hi(x1), bf_lo(y0), bf_hi(y0), bf_lo(y1), bf_hi(y1)};
#pragma unroll
      for (int r = 0; r < 8; ++r) v[8 * cc + r] = acc_o[ob][8 * cc + r] + sk[r];
    }
    store_block_bf16(yrow + 32 * ob, v, lh, ok);
  }
  if (probe) { p.clk[2] = __builtin_amdgcn_s_memtime(); p.clk[3] = __builtin_amdgcn_s_memrealtime(); p.clk[7] = (unsigned long long)T; }
}

// ---- width 256 --------------------------------------------------------------------------------------------------------------
// The same block at K = 256 (level 1 of the headline config: 32 768 tokens, d_ff 768).  A row's fragments (64 registers) plus
// the 256-wide fp32 output accumulators (128) no longer fit two waves per SIMD, so a workgroup is 4 waves x 32 rows with the
// whole register file per wave, and a d_ff tile's weights (64 KiB up + 32 KiB down) stream as TWO 48 KiB half units through the
// same 3-slot ring:   A_t = up-projection k-steps 0, 1, 2      B_t = up-projection k-step 3 + both down-projection blocks.
//   top of tile: wait A_t, barrier (B_t-1's slot is free: request A_t+1) | up chunks 0..11 from A_t |
//   wait B_t, barrier (A_t's slot is free: request B_t+1) | up chunks 12..15, GEGLU, down projection from B_t
// A_t is released after 40 % of the tile and B_t at its end, so every request has a whole tile (~7 000 clocks) to arrive.
// Weight fragments are read one chunk ahead of their MFMAs, across the phase boundaries too (explicit double buffer: one wave
// per SIMD has no partner to cover LDS latency).
constexpr int F2_NW = 4;

__global__ __launch_bounds__(F2_NW * 64) void ffn256_kernel(const FArgs p) {
  constexpr int NC = 16, K = 256, KB = 8;
  constexpr int HU = 3 * WBLK;                        // half unit: 2 up-projection blocks + 1 down-projection block
  constexpr int PCS = HU / 1024 / F2_NW;              // 12 pieces per wave per half unit
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lh = lane >> 5;
  const int row = blockIdx.x * (F2_NW * 32) + wid * 32 + l31;
  const bool ok = row < p.M;
  const int rowc = ok ? row : p.M - 1;
  const int T = p.n_tiles, U = 2 * T;
  const bool probe = p.clk && blockIdx.x == 0 && tid == 0;
  if (probe) { p.clk[0] = __builtin_amdgcn_s_memtime(); p.clk[1] = __builtin_amdgcn_s_memrealtime(); }

  auto issue = [&](int u) {
    const int t = u >> 1, half = u & 1;
    char* slot = smem + (u % 3) * HU;
#pragma unroll
    for (int i = 0; i < PCS; ++i) {
      const int pi = wid + F2_NW * i;                 // 0..47
      const char* src;
      if (half == 0) src = p.Wu + ((size_t)t * 4) * WBLK + pi * 1024;                       // up k-steps 0, 1, 2: 48 contiguous KiB
      else if (pi < 16) src = p.Wu + ((size_t)t * 4 + 3) * WBLK + pi * 1024;                // up k-step 3
      else src = p.Wd + ((size_t)((pi - 16) >> 4) * T + t) * WBLK + ((pi - 16) & 15) * 1024;   // down rows 0..127, 128..255
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane * 16),
                                       (__attribute__((address_space(3))) void*)(slot + pi * 1024), 16, 0, 0);
    }
  };
  int issued = 0;
  for (; issued < 3 && issued < U; ++issued) issue(issued);

  bf16x8 a[NC];
  float rs;
  {
    const u32x4* ap = reinterpret_cast<const u32x4*>(p.X + (size_t)rowc * K + 8 * lh);
    const float* sp = p.scale + (size_t)(rowc / p.rows_per_sample) * p.scale_stride + 8 * lh;
    u32x4 raw[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) raw[c] = ap[2 * c];
    float ssq = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(sp + 16 * c), s1 = *reinterpret_cast<const f32x4*>(sp + 16 * c + 4);
      float x[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { x[2 * e] = bf_lo(raw[c][e]); x[2 * e + 1] = bf_hi(raw[c][e]); }
#pragma unroll
      for (int e = 0; e < 8; ++e) ssq = fmaf(x[e], x[e], ssq);
      u32x4 o = {pack_bf16(x[0] * s0[0], x[1] * s0[1]), pack_bf16(x[2] * s0[2], x[3] * s0[3]),
                 pack_bf16(x[4] * s1[0], x[5] * s1[1]), pack_bf16(x[6] * s1[2], x[7] * s1[3])};
      asm volatile("" : "+v"(o));
      a[c] = __builtin_bit_cast(bf16x8, o);
    }
    ssq += __shfl_xor(ssq, 32, 64);
    rs = rsqrtf(ssq / (float)K + p.eps);
  }
  const float rsh = 0.5f * rs;
  if (probe) p.clk[4] = __builtin_amdgcn_s_memtime();

  int off4[4];
#pragma unroll
  for (int cc = 0; cc < 4; ++cc) off4[cc] = swz128(l31, 2 * cc + lh);
  f32x16 acc_o[KB];
#pragma unroll
  for (int ob = 0; ob < KB; ++ob)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_o[ob][r] = 0.f;

  f32x16 acc[4];
  bf16x8 hf[4];
  // weight fragment of the up projection: chunk c (0..15) of block j; chunks 0..11 in half unit A, 12..15 in B's first block
  auto up_frag = [&](const char* sa, const char* sb, int c, int j) -> bf16x8 {
    const char* blk = c < 12 ? sa + (c >> 2) * WBLK : sb;
    return *reinterpret_cast<const bf16x8*>(blk + off4[c & 3] + j * 32 * 128);
  };
  auto down_frag = [&](const char* sb, int half, int g, int ob) -> bf16x8 {
    return *reinterpret_cast<const bf16x8*>(sb + (1 + half) * WBLK + ob * 32 * 128 + off4[g]);
  };
  for (int t = 0; t < T; ++t) {
    const int ua = 2 * t, ub = 2 * t + 1;
    wait_vm_count(PCS * (issued - 1 - ua));            // A_t is in (later half units may stay in flight)
    KD_BARRIER();                                      // ... for every wave; every wave is done with B_{t-1}'s slot
    if (t > 0 && issued < U) { issue(issued); ++issued; }       // A_{t+1} (requested in the prologue for t = 0)
    const char* sa = smem + (ua % 3) * HU;
    const char* sb = smem + (ub % 3) * HU;
    // ---- up projection, 16 chunks, fragments one chunk ahead ---------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    bf16x8 wf[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) wf[0][j] = up_frag(sa, sb, 0, j);
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      if (c == 11) {
        // the last fragments of A_t are in flight (chunk 11's were requested a step ago): once they are in registers A_t's slot
        // goes to B_{t+1}; B_t has to be in for chunk 12
        __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0)
        wait_vm_count(PCS * (issued - 1 - ub));
        KD_BARRIER();
        if (issued < U) { issue(issued); ++issued; }
      }
      if (c + 1 < 16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[(c + 1) & 1][j] = up_frag(sa, sb, c + 1, j);
      } else {
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) wf[(c + 1) & 1][ob] = down_frag(sb, 0, 0, ob);  // first fragments of the down projection: they
      }                                                                                //  arrive while the GEGLU below runs
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[c & 1][j], a[c], acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      unsigned pk[8];
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 o = geglu_pair(f32x2{acc[2 * jj][r], acc[2 * jj][r + 1]} * rsh, f32x2{acc[2 * jj + 1][r], acc[2 * jj + 1][r + 1]} * rs);
        pk[r >> 1] = pack_bf16(o.x, o.y);
      }
      hf[2 * jj] = __builtin_bit_cast(bf16x8, u32x4{pk[0], pk[1], pk[2], pk[3]});
      hf[2 * jj + 1] = __builtin_bit_cast(bf16x8, u32x4{pk[4], pk[5], pk[6], pk[7]});
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- down projection: 8 steps (2 halves of the output x 4 hidden chunks), wf[0] holds step 0's fragments ------------------------
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      const int half = st >> 2, g = st & 3;
      if (st + 1 < 8) {
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) wf[(st + 1) & 1][ob] = down_frag(sb, (st + 1) >> 2, (st + 1) & 3, ob);
      }
#pragma unroll
      for (int ob = 0; ob < 4; ++ob)
        acc_o[4 * half + ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[st & 1][ob], hf[g], acc_o[4 * half + ob], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);                // B_t's fragments are in registers before the next tile's barrier releases its slot
  }
  if (probe) { p.clk[5] = __builtin_amdgcn_s_memtime(); p.clk[6] = p.clk[5]; }
  const u16* xrow = p.X + (size_t)rowc * K;
  u16* yrow = p.Y + (size_t)rowc * K;
#pragma unroll
  for (int ob = 0; ob < KB; ++ob) {
    float sk[16], v[16];
    load_block_bf16(xrow + 32 * ob, sk, lh);
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = acc_o[ob][r] + sk[r];
    store_block_bf16(yrow + 32 * ob, v, lh, ok);
  }
  if (probe) { p.clk[2] = __builtin_amdgcn_s_memtime(); p.clk[3] = __builtin_amdgcn_s_memrealtime(); p.clk[7] = (unsigned long long)T; }
}

}  // namespace b16
}  // namespace kd

using namespace kd;
using namespace kd::b16;

extern "C" int kd_ffn_bf16_supported(int M, int K, int d_ff) {
  // below ~16k rows the panels do not fill the chip and the two-kernel form (row-parallel over more, smaller units) is faster
  // (option "ffn_bf16_min_rows": A/B runs at small batches)
  if (!(M >= option("ffn_bf16_min_rows", 16384) && d_ff > 0 && d_ff % 64 == 0 && option("ffn_fused", 1))) return 0;
  // K = 256: correct, but measured level with the two-kernel form (65.1 vs 65.6 us at the level-1 shape; 8 300 clocks per tile for 3 072
  // matrix clocks with one wave per SIMD) -- on request only ("ffn_fused_256")
  return (K == 128 || (K == 256 && option("ffn_fused_256", 0))) ? 1 : 0;
}

extern "C" int kd_ffn_bf16(const KdFfn* dp, void* stream) {
  if (!dp) return fail(KD_EINVAL, "kd_ffn_bf16: null descriptor");
  const KdFfn& d = *dp;
  if (!d.x || !d.out || !d.scale || !d.Wp_up || !d.Wp_down) return fail(KD_EINVAL, "kd_ffn_bf16: null x / out / scale / Wp_up / Wp_down");
  if (d.M <= 0 || d.rows_per_sample <= 0 || (d.scale_stride & 3)) return fail(KD_EINVAL, "kd_ffn_bf16: needs M, rows_per_sample > 0, scale_stride %% 4 == 0");
  if ((d.K != 128 && d.K != 256) || d.d_ff <= 0 || d.d_ff % 64) return fail(KD_EINVAL, "kd_ffn_bf16: K = %d, d_ff = %d not supported (K == 128 or 256, d_ff %% 64 == 0)", d.K, d.d_ff);
  hipStream_t s = (hipStream_t)stream;
  FArgs a{};
  a.X = reinterpret_cast<const u16*>(d.x); a.Y = reinterpret_cast<u16*>(d.out);
  a.Wu = reinterpret_cast<const char*>(d.Wp_up); a.Wd = reinterpret_cast<const char*>(d.Wp_down);
  a.scale = d.scale; a.scale_stride = d.scale_stride; a.rows_per_sample = d.rows_per_sample; a.eps = d.eps;
  a.M = d.M; a.n_tiles = d.d_ff / 64;
  a.clk = g_clk;
  a.warm = option("code_warm", KD_CODE_WARM_DEFAULT);
  if (d.attn && d.K != 128) return fail(KD_EINVAL, "kd_ffn_bf16: the fused out projection needs K == 128 (K=%d)", d.K);
  if (d.K == 256) {
    constexpr int LDS256 = 9 * WBLK;
    static LdsAttr attr256;
    attr256.ensure(reinterpret_cast<const void*>(ffn256_kernel), LDS256);
    char nm2[96] = "ffn_bf16";
    if (prof_on()) snprintf(nm2, sizeof(nm2), "ffn_bf16 M=%d K=%d dff=%d", d.M, d.K, d.d_ff);
    LaunchScope prof2(nm2, 2.0 * d.M * (double)d.K * (3.0 * d.d_ff), 4.0 * d.M * (double)d.K + 6.0 * d.d_ff * (double)d.K, s);
    hipLaunchKernelGGL(ffn256_kernel, dim3((unsigned)((d.M + F2_NW * 32 - 1) / (F2_NW * 32))), dim3(F2_NW * 64), LDS256, s, a);
    return check_launch("kd_ffn_bf16");
  }
  // 1 (default) plain, 3 skewed wave pairs.  Measured equal within noise (64.5 / 68.6 us at the level-0 shape; a third form with
  // one wave per SIMD and the two row blocks' MFMA / GEGLU streams interleaved instruction by instruction took 71 us):
  // profiles/r02_ffn_fused.md -- under this kernel the chip runs against its power limit and re-arranging the same work buys nothing.
  const int variant = option("ffn_variant", 1);
  const bool outp = d.attn != nullptr;
  if (outp) {
    if (!d.Wp_out) return fail(KD_EINVAL, "kd_ffn_bf16: attn without Wp_out");
    a.Att = reinterpret_cast<const u16*>(d.attn); a.Wo = reinterpret_cast<const char*>(d.Wp_out);
  }
  auto kern = outp ? ffn_kernel<8, false, true> : (variant == 3 ? ffn_kernel<8, true> : ffn_kernel<8, false>);
  const int panel = FF_NW * 32, threads = FF_NW * 64;
  const int LDS = (variant == 3 && !outp ? 10 : 9) * WBLK;
  static LdsAttr attr_set[3];
  attr_set[0].ensure(reinterpret_cast<const void*>(ffn_kernel<8, false>), 9 * WBLK);
  attr_set[1].ensure(reinterpret_cast<const void*>(ffn_kernel<8, true>), 10 * WBLK);
  attr_set[2].ensure(reinterpret_cast<const void*>(ffn_kernel<8, false, true>), 9 * WBLK);
  char nm[96] = "ffn_bf16";
  if (prof_on()) snprintf(nm, sizeof(nm), "%s M=%d K=%d dff=%d", outp ? "ffn_bf16+out" : "ffn_bf16", d.M, d.K, d.d_ff);
  const double flops = 2.0 * d.M * (double)d.K * (3.0 * d.d_ff + (outp ? d.K : 0));
  const double bytes = (outp ? 6.0 : 4.0) * d.M * (double)d.K + 6.0 * d.d_ff * (double)d.K + (outp ? 2.0 * d.K * d.K : 0.0);
  LaunchScope prof(nm, flops, bytes, s);
  hipLaunchKernelGGL(kern, dim3((unsigned)((d.M + panel - 1) / panel)), dim3(threads), LDS, s, a);
  return check_launch("kd_ffn_bf16");
}

KD_TEXT_PAD(ffn_bf16)      // last function of this code object: kd_common.h, code warm-up
