// Fused feed-forward block in fp32-parity ("split3") arithmetic:  out = x + down_proj( GEGLU( up_proj( AdaRMSNorm(x, cond) ) ) )
// (image_transformer_v2.py:479-493; norm :155-166, LinearGEGLU :132-139 + :89-95, Linear :126-129).  fp32 activations in HBM, every product
// as hi*hi + hi*lo + lo*hi on the bf16 MFMA with fp32 accumulation, and the d_ff-wide hidden activation never leaves the chip: as separate
// kernels the level-0 block of the headline config writes and re-reads 2 x 201 MB of it per layer, next to 134 MB for x in and out.
//
// Built on the A-stationary projection of gemm_x3.hip (same prologue, same stage-pipelined K loop, same ring):
//   * a 4-wave workgroup owns 128 rows; a lane owns ONE row: its normalised, scaled, split row (a_hi / a_lo) stays in registers
//     (K = 128: 64 VGPRs; K = 256: the 128 AccVGPRs a0..a127, named by hand -- x3_common.h) for all of d_ff;
//   * per 64-feature d_ff tile the packed weights stream through the LDS ring as 16 KiB stages: K / 32 stages of the up projection's
//     GEGLU tile (32 value rows + 32 gate rows, twice), then 2 K / 128 stages of the down projection's k-steps for those 64 features
//     (pack layout 2: inside a group of 16 the k order of the GEGLU accumulators, so its outputs ARE the down projection's B operand);
//   * GEGLU in the lane that owns the row, split into hi / lo fragments in registers, then straight into the down projection's MFMAs;
//   * the K-wide fp32 output accumulators of a row block live in named AccVGPRs for the whole kernel (64 at K = 128, 128 at K = 256):
//     they are the C operand of asm MFMAs and never compete with the up projection's accumulators for ArchVGPRs;
//   * epilogue: + x (re-read: 512 / 1024 bytes per row, the only second touch of x), fp32, through the wave's LDS strip.
// One workgroup per CU (the register file is the wave's; at K = 128 the wave would fit 256 registers with 64 AccVGPRs, but once a kernel
// uses AccVGPRs hipcc splits a two-wave budget 128 + 128 whatever it is told, and spills): the GEGLU phase of a tile (packed fp32 math, no MFMA in flight: v_pk_* beside
// an MFMA costs 20 cycles each, profiles/r03_issue_model.md) is the only part of a tile the matrix pipe sits out.
#include "x3_common.h"

namespace kd {
namespace x3 {

struct FArgs3 {
  const float* X; float* Y;
  const char* Wu; const char* Wd;
  const float* scale; int scale_stride, rows_per_sample; float eps;
  int M, d_ff, n_tiles;
  int warm;
  unsigned long long* clk;
  const float* Att; const char* Wo;   // fused out projection in front of the block (round 3): X <- X + Att Wo^T first; NULL = none
};

// acc_o block ob (32 output features) of this lane's row: AccVGPRs AO + 16 ob .. + 15
template <int IDX>
__device__ __forceinline__ void mfma_acc_ag(const bf16x8 w, const bf16x8 h) {
  asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" :: "v"(w), "v"(h), "i"(IDX), "i"(IDX + 15) : KD_AGPR_ALL);
}
template <int IDX>
__device__ __forceinline__ void areg_zero16() {
  static_for<16>([&](auto i_) { asm volatile("v_accvgpr_write_b32 a[%c0], 0" :: "i"(IDX + decltype(i_)::value) : KD_AGPR_ALL); });
}
template <int IDX>
__device__ __forceinline__ f32x4 areg_read4() {
  f32x4 v;
  asm volatile("v_accvgpr_read_b32 %0, a[%c4]\n\tv_accvgpr_read_b32 %1, a[%c5]\n\tv_accvgpr_read_b32 %2, a[%c6]\n\tv_accvgpr_read_b32 %3, a[%c7]"
               : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]) : "i"(IDX), "i"(IDX + 1), "i"(IDX + 2), "i"(IDX + 3));
  return v;
}

// output accumulators (named AccVGPRs) += W fragment x activation fragment held in AccVGPRs too (fused out projection, see ffn_x3h_kernel)
template <int IDX, int BIDX>
__device__ __forceinline__ void mfma_acc_aa(const bf16x8 w) {
  asm volatile("v_mfma_f32_32x32x16_bf16 a[%c1:%c2], %0, a[%c3:%c4], a[%c1:%c2]" :: "v"(w), "i"(IDX), "i"(IDX + 15), "i"(BIDX), "i"(BIDX + 3) : KD_AGPR_ALL);
}

template <int NC /* K / 16: 8 or 16 */, bool OUTP = false /* K = 256: the attention block's out projection first (FArgs3.Att / Wo) */>
__global__ __launch_bounds__(256, 1) void ffn_x3_kernel(const FArgs3 p) {
  static_assert(!OUTP || NC == 16, "fused out projection: width 256 here, width 128 in ffn_x3h_kernel");
  constexpr int K = NC * 16, NKU = NC / 2;                      // ring stages of a tile's up projection
  constexpr int NOB = K / 32, NKD = 2 * (K / 128);              // output blocks of a row; ring stages of a tile's down-projection k-steps
  constexpr int UNIT = NKU + NKD;                               // stages per d_ff tile
  constexpr int NSTG = 8, PDIST = NSTG - 1, PB = 4;
  constexpr bool AG = NC >= 16;                                 // activation fragments in a0 .. a(8 NC - 1)
  constexpr int AO = AG ? 8 * NC : 0;                           // first AccVGPR of the output accumulators
  constexpr int WAREA = (NSTG / 4) * STG;
  constexpr int KH = WAREA / 128 >= K ? K : WAREA / 128, NR = K / KH, CPR = KH / 4, PIECES = 32 * CPR / 64;
  constexpr int SCL = K * 4 < 1024 ? 1024 : K * 4;
  // K = 128: a wave's rows take 16 KiB; they borrow ring slots 4 .. 7 and the first EARLY weight stages are requested into slots 0 .. 3 before
  // the rows have even arrived (the first MFMA then waits for nothing).  K = 256: the rows need all 8 slots.
  constexpr int EARLY = K == 128 ? 4 : 0;
  static_assert(NR == 1 && (K == 128 || K == 256), "widths 128 and 256");
  static_assert(UNIT <= 2 * NSTG, "slot arithmetic below assumes at most two trips round the ring per tile");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const auto warm = code_warm_begin<(NC <= 8 ? 16 : 28) * 1024>((int)blockIdx.x < p.warm && tid < 64);
  const int m0 = blockIdx.x * 128;
  const int T = p.n_tiles, total = T * UNIT;
  const bool probe = p.clk && blockIdx.x == (gridDim.x * 5) / 8 && tid == 0;      // a workgroup of a later round: the chip under load
  if (probe) { p.clk[0] = __builtin_amdgcn_s_memtime(); p.clk[1] = __builtin_amdgcn_s_memrealtime(); }
  const WgStamp wgs = wg_stamp_begin(p.clk);

  // The weight stream: tile t contributes UNIT stages -- u < NKU: up block (t, u); else down block (n-tile (u - NKU) / 2, k-step
  // 2 t + (u - NKU) % 2).  Stage (t, u) sits in ring slot (t UNIT + u) % NSTG.  Requests past the end re-request the last stage (uniform
  // 4 pieces per wave per stage: gemm_x3.hip).  `off` (compile-time where it matters) is relative to tile t's first stage, so that no
  // run-time division is needed.
  auto issue_rel = [&](int t, int off, int j) {
    int tt = t + off / UNIT, u = off % UNIT;
    if (tt >= T) { tt = T - 1; u = UNIT - 1; }
    const char* src = (u < NKU ? p.Wu + ((size_t)tt * NKU + u) * STG
                               : p.Wd + ((size_t)((u - NKU) >> 1) * (2 * T) + 2 * tt + ((u - NKU) & 1)) * STG) + wid * (PB * 1024) + lane * 16;
    char* dst = smem + ((t * UNIT + off) % NSTG) * STG + wid * (PB * 1024);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + j * 1024),
                                     (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, 0, 0);
  };

  // ---- this wave's 32 rows -> a_hi / a_lo (the prologue of gemm_x3.hip) ---------------------------------------------------------------
  const int row = m0 + wid * 32 + l31;
  const bool ok = row < p.M;
  const int rowc = ok ? row : p.M - 1;
  bf16x8 a_hi[AG ? 1 : NC], a_lo[AG ? 1 : NC];
  float rs = 1.f;
  f32x4 xres[OUTP ? NOB : 1][4];                       // OUTP: x of the lane's row in the C layout (features 32 ob + 8 g + 4 lh + 0..3)
  if constexpr (OUTP) {
    const float* xr = p.X + (size_t)rowc * K + 4 * lh;
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
      for (int g = 0; g < 4; ++g) xres[ob][g] = *reinterpret_cast<const f32x4*>(xr + 32 * ob + 8 * g);
  }
  {
    char* stage = smem + (EARLY ? EARLY * STG + wid * STG : wid * WAREA);
    char* scl = smem + NSTG * STG + wid * SCL;
    const int r_first = min(m0 + wid * 32, p.M - 1), r_last = min(m0 + wid * 32 + 31, p.M - 1);
    const bool uni = p.scale_stride == 0 || r_first / p.rows_per_sample == r_last / p.rows_per_sample;
    if (uni) {
      const char* ssrc = reinterpret_cast<const char*>(p.scale + (size_t)(r_first / p.rows_per_sample) * p.scale_stride);
#pragma unroll
      for (int i = 0; i < SCL / 1024; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ssrc + (i * 1024 + lane * 16) % (K * 4)),
                                         (__attribute__((address_space(3))) void*)(scl + i * 1024), 16, 0, 0);
    }
    const float* sp = p.scale + (size_t)(rowc / p.rows_per_sample) * p.scale_stride + 8 * lh;
    const float* spl = reinterpret_cast<const float*>(scl) + 8 * lh;
    float ssq = 0.f;
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const int ci = i * 64 + lane, rr = ci / CPR, qs = ci % CPR;
      const int grow = min(m0 + wid * 32 + rr, p.M - 1);
      const char* src = reinterpret_cast<const char*>((OUTP ? p.Att : p.X) + (size_t)grow * K) + ((qs ^ (rr & 15)) << 4);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(stage + i * 1024), 16, 0, 0);
    }
    if constexpr (EARLY > 0) {                       // the first weight stages go to the slots no row borrows, behind the rows in the queue
#pragma unroll
      for (int s = 0; s < EARLY; ++s)
#pragma unroll
        for (int j = 0; j < PB; ++j) issue_rel(0, s, j);
      wait_vm(PB * EARLY);
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const char* rowp = stage + l31 * (KH * 4);
    static_for<NC / 4>([&](auto c4_) {
      constexpr int c0 = 4 * decltype(c4_)::value;
      f32x4 x0[4], x1[4], s0[4], s1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = 4 * (c0 + u) + 2 * lh;
        x0[u] = *reinterpret_cast<const f32x4*>(rowp + ((q ^ (l31 & 15)) << 4));
        x1[u] = *reinterpret_cast<const f32x4*>(rowp + (((q + 1) ^ (l31 & 15)) << 4));
        if constexpr (!OUTP) {
          if (uni) {
            s0[u] = *reinterpret_cast<const f32x4*>(spl + 16 * (c0 + u));
            s1[u] = *reinterpret_cast<const f32x4*>(spl + 16 * (c0 + u) + 4);
          } else {
            s0[u] = *reinterpret_cast<const f32x4*>(sp + 16 * (c0 + u));
            s1[u] = *reinterpret_cast<const f32x4*>(sp + 16 * (c0 + u) + 4);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      static_for<4>([&](auto u_) {
        constexpr int u = decltype(u_)::value;
        u32x4 hi, lo;
        if constexpr (OUTP) {
          split8(x0[u], x1[u], hi, lo);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) ssq = fmaf(x0[u][e], x0[u][e], fmaf(x1[u][e], x1[u][e], ssq));
          split8(x0[u] * s0[u], x1[u] * s1[u], hi, lo);
        }
        if constexpr (AG) {
          areg_write4<8 * (c0 + u)>(hi);
          areg_write4<8 * (c0 + u) + 4>(lo);
        } else {
          asm volatile("" : "+v"(hi), "+v"(lo));
          a_hi[c0 + u] = __builtin_bit_cast(bf16x8, hi);
          a_lo[c0 + u] = __builtin_bit_cast(bf16x8, lo);
        }
      });
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (!OUTP) {
      ssq += __shfl_xor(ssq, 32, 64);
      rs = rsqrtf(ssq / (float)K + p.eps);
    }
  }
  if constexpr (OUTP) {                                // the output accumulators start from x
    static_for<NOB>([&](auto ob_) {
      static_for<4>([&](auto g_) {
        constexpr int ob = decltype(ob_)::value, gq = decltype(g_)::value;
        areg_write4<AO + 16 * ob + 4 * gq>(__builtin_bit_cast(u32x4, xres[ob][gq]));
      });
    });
  } else {
    static_for<NOB>([&](auto ob_) { areg_zero16<AO + 16 * decltype(ob_)::value>(); });
  }
  code_warm_end(warm);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  KD_BARRIER();                                        // every wave has taken its rows out of the slots it borrowed
  // OUTP: the stream starts with Wo's 2 NKU stages ([n-tile of 128 out rows][32-k stage], plain layout): twice round the ring, so the
  // block's own stages keep their ring positions
  constexpr int NWO = OUTP ? 2 * NKU : 0;
  auto issue_wo = [&](int q, int j) {
    const char* src = p.Wo + (size_t)q * STG + wid * (PB * 1024) + j * 1024 + lane * 16;
    char* dst = smem + (q % NSTG) * STG + wid * (PB * 1024) + j * 1024;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };
  if constexpr (OUTP) {
#pragma unroll
    for (int q = 0; q < PDIST; ++q)
#pragma unroll
      for (int j = 0; j < PB; ++j) issue_wo(q, j);
  } else {
#pragma unroll
    for (int s = EARLY; s < PDIST; ++s)
#pragma unroll
      for (int j = 0; j < PB; ++j) issue_rel(0, s, j);
  }
  if (probe) p.clk[4] = __builtin_amdgcn_s_memtime();

  const int o0 = swz64(l31, lh), o1 = swz64(l31, 2 + lh);
  f32x16 acc[4];
  bf16x8 wh[2][4], wl[2][4];
  auto read_frags = [&](int slot, int h, bf16x8 (&fh)[4], bf16x8 (&fl)[4]) {
    const char* st = smem + slot * STG + (h ? o1 : o0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      fh[j] = *reinterpret_cast<const bf16x8*>(st + j * 32 * 64);
      fl[j] = *reinterpret_cast<const bf16x8*>(st + IMG + j * 32 * 64);
    }
  };
  // stage s in (its slot's pieces landed for every wave) and stage s - 1's slot released: the wait + barrier in the MIDDLE of a stage
  auto next_stage_in = [&]() {
    wait_vm(PB * (PDIST - 2));                        // behind stage s + 1: the PDIST - 2 stages requested after it (no stores until the end)
    KD_BARRIER();
  };
  wait_vm(PB * (PDIST - 1));
  KD_BARRIER();
  read_frags(0, 0, wh[0], wl[0]);
  if constexpr (OUTP) {
    // ================= out projection: x (in the output accumulators) += att Wo^T, 2 n-tiles x NKU stages of 24 MFMAs ======================
    constexpr std::integral_constant<int, 0> I0{};
    constexpr std::integral_constant<int, 1> I1{};
    constexpr std::integral_constant<int, 2> I2{};
    constexpr std::integral_constant<int, 3> I3{};
    static_for<NWO>([&](auto q_) {
      constexpr int q = decltype(q_)::value, ntile = q / NKU, ks = q % NKU;
      auto oo = [&](auto b_, auto j_, bool w_lo, auto al_) {
        constexpr int b = decltype(b_)::value, j = decltype(j_)::value, al = decltype(al_)::value, c = 2 * ks + b;
        mfma_acc_aa<AO + 16 * (4 * ntile + j), 8 * c + 4 * al>(w_lo ? wl[b][j] : wh[b][j]);
      };
      auto request = [&](int j) {                      // stream position q + PDIST: Wo's later stages, then the block's first ones
        if constexpr (q + PDIST < NWO) issue_wo(q + PDIST, j);
        else issue_rel(0, q + PDIST - NWO, j);
      };
      oo(I0, I0, true, I0);
      __builtin_amdgcn_sched_barrier(0);
      read_frags(q % NSTG, 1, wh[1], wl[1]);
      __builtin_amdgcn_sched_barrier(0);
      oo(I0, I1, true, I0); oo(I0, I2, true, I0); oo(I0, I3, true, I0);
      oo(I0, I0, false, I1); oo(I0, I1, false, I1); oo(I0, I2, false, I1); oo(I0, I3, false, I1);
      oo(I0, I0, false, I0); oo(I0, I1, false, I0); oo(I0, I2, false, I0); oo(I0, I3, false, I0);
      __builtin_amdgcn_sched_barrier(0);
      next_stage_in();
      oo(I1, I0, true, I0);
      __builtin_amdgcn_sched_barrier(0);
      read_frags((q + 1) % NSTG, 0, wh[0], wl[0]);
      __builtin_amdgcn_sched_barrier(0);
      oo(I1, I1, true, I0); oo(I1, I2, true, I0);
      request(0);
      __builtin_amdgcn_sched_barrier(0);
      oo(I1, I3, true, I0); oo(I1, I0, false, I1); oo(I1, I1, false, I1);
      request(1);
      __builtin_amdgcn_sched_barrier(0);
      oo(I1, I2, false, I1); oo(I1, I3, false, I1); oo(I1, I0, false, I0);
      request(2);
      __builtin_amdgcn_sched_barrier(0);
      oo(I1, I1, false, I0); oo(I1, I2, false, I0);
      request(3);
      oo(I1, I3, false, I0);
      __builtin_amdgcn_sched_barrier(0);
    });
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");  // asm MFMA results -> v_accvgpr_read
    // ---- new x (C layout, in the accumulators) -> norm statistics, scale, hi / lo fragments in the k order of pack layout 3 -----------
    {
      const char* scl = smem + NSTG * STG + wid * SCL;
      const int r_first = min(m0 + wid * 32, p.M - 1), r_last = min(m0 + wid * 32 + 31, p.M - 1);
      const bool uni = p.scale_stride == 0 || r_first / p.rows_per_sample == r_last / p.rows_per_sample;
      const float* sg = uni ? nullptr : p.scale + (size_t)(rowc / p.rows_per_sample) * p.scale_stride + 4 * lh;
      const float* sl = reinterpret_cast<const float*>(scl) + 4 * lh;
      float ssq = 0.f;
      static_for<NOB>([&](auto ob_) {
        constexpr int ob = decltype(ob_)::value;
        f32x4 v[4], sc[4];
        static_for<4>([&](auto g_) { v[decltype(g_)::value] = areg_read4<AO + 16 * ob + 4 * decltype(g_)::value>(); });
#pragma unroll
        for (int g = 0; g < 4; ++g) sc[g] = uni ? *reinterpret_cast<const f32x4*>(sl + 32 * ob + 8 * g) : *reinterpret_cast<const f32x4*>(sg + 32 * ob + 8 * g);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int e = 0; e < 4; ++e) ssq = fmaf(v[g][e], v[g][e], ssq);
        static_for<2>([&](auto hc_) {
          constexpr int hc = decltype(hc_)::value;
          u32x4 hi, lo;
          split8(v[2 * hc] * sc[2 * hc], v[2 * hc + 1] * sc[2 * hc + 1], hi, lo);
          areg_write4<8 * (2 * ob + hc)>(hi);
          areg_write4<8 * (2 * ob + hc) + 4>(lo);
        });
      });
      ssq += __shfl_xor(ssq, 32, 64);
      rs = rsqrtf(ssq / (float)K + p.eps);
    }
  }
  const float rsh = 0.5f * rs;

  for (int t = 0; t < T; ++t) {
    const int sbase = t * UNIT;
    const int slot0 = sbase % NSTG;                   // ring slot of the tile's first stage (UNIT is not a multiple of NSTG at K = 128)
    if (probe && t == 2) p.clk[8] = __builtin_amdgcn_s_memtime();
    // (no zeroing of the accumulators: the first MFMA of each chain takes the constant 0 as its C operand -- 64 v_mov per tile less, and
    // no v_mov -> SrcC hazard, which the compiler pads for its own MFMAs only)
    // ================= up projection of tile t: value / gate accumulators of its 2 x 32 hidden features (K / 32 stages) =================
    static_for<NKU>([&](auto ks_) {
      constexpr int ks = decltype(ks_)::value;
      const int slot = (slot0 + ks) % NSTG, nslot = (slot0 + ks + 1) % NSTG;
      auto mm = [&](auto b_, int j, bool w_lo, auto a_lo_) {
        constexpr int b = decltype(b_)::value, al = decltype(a_lo_)::value, c = 2 * ks + b;
        const bf16x8& w = w_lo ? wl[b][j] : wh[b][j];
        constexpr bool first = ks == 0 && b == 0;             // the chain of block j starts with its w_lo x a_hi term of chunk 0
        if constexpr (AG) {
          if (first && w_lo) mfma_ag0<8 * c + 4 * al>(acc[j], w);
          else mfma_ag<8 * c + 4 * al>(acc[j], w);
        } else {                                              // (asm: the accumulators must stay out of the AccVGPRs this kernel names)
          if (first && w_lo) mfma_vv0(acc[j], w, al ? a_lo[c] : a_hi[c]);
          else mfma_vv(acc[j], w, al ? a_lo[c] : a_hi[c]);
        }
      };
      constexpr std::integral_constant<int, 0> I0{};
      constexpr std::integral_constant<int, 1> I1{};
      mm(I0, 0, true, I0);
      __builtin_amdgcn_sched_barrier(0);
      read_frags(slot, 1, wh[1], wl[1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 1; j < 4; ++j) mm(I0, j, true, I0);
#pragma unroll
      for (int j = 0; j < 4; ++j) mm(I0, j, false, I1);
#pragma unroll
      for (int j = 0; j < 4; ++j) mm(I0, j, false, I0);
      __builtin_amdgcn_sched_barrier(0);
      next_stage_in();
      mm(I1, 0, true, I0);
      __builtin_amdgcn_sched_barrier(0);
      read_frags(nslot, 0, wh[0], wl[0]);
      __builtin_amdgcn_sched_barrier(0);
      mm(I1, 1, true, I0);
      mm(I1, 2, true, I0);
      issue_rel(t, ks + PDIST, 0);
      __builtin_amdgcn_sched_barrier(0);
      mm(I1, 3, true, I0);
      mm(I1, 0, false, I1);
      mm(I1, 1, false, I1);
      issue_rel(t, ks + PDIST, 1);
      __builtin_amdgcn_sched_barrier(0);
      mm(I1, 2, false, I1);
      mm(I1, 3, false, I1);
      mm(I1, 0, false, I0);
      issue_rel(t, ks + PDIST, 2);
      __builtin_amdgcn_sched_barrier(0);
      mm(I1, 1, false, I0);
      mm(I1, 2, false, I0);
      issue_rel(t, ks + PDIST, 3);
      mm(I1, 3, false, I0);
      __builtin_amdgcn_sched_barrier(0);
    });
    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));      // asm MFMA results -> vector reads
    if (probe && t == 0) p.clk[5] = __builtin_amdgcn_s_memtime();
    if (probe && t == 2) p.clk[9] = __builtin_amdgcn_s_memtime();

    // ================= GEGLU in the lane that owns the row -> hi / lo B-operand fragments of the 4 hidden 16-chunks ======================
    // chunk c' = 2 jj + (g >> 1) of the tile; the lane's 8 values of it are accumulator registers 8 (c' & 1) .. + 7 of blocks (2 jj, 2 jj + 1),
    // i.e. hidden features 16 c' + 8 (e >> 2) + 4 lh + (e & 3): pack layout 2 of the down projection's weight puts its k in that order
    bf16x8 hf_hi[4], hf_lo[4];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int hc = 0; hc < 2; ++hc) {
        f32x4 v0, v1;
        const int r0 = 8 * hc;
        {
          const f32x2 a = geglu_pair(f32x2{acc[2 * jj][r0], acc[2 * jj][r0 + 1]} * rsh, f32x2{acc[2 * jj + 1][r0], acc[2 * jj + 1][r0 + 1]} * rs);
          const f32x2 b = geglu_pair(f32x2{acc[2 * jj][r0 + 2], acc[2 * jj][r0 + 3]} * rsh, f32x2{acc[2 * jj + 1][r0 + 2], acc[2 * jj + 1][r0 + 3]} * rs);
          v0 = f32x4{a.x, a.y, b.x, b.y};
        }
        {
          const f32x2 a = geglu_pair(f32x2{acc[2 * jj][r0 + 4], acc[2 * jj][r0 + 5]} * rsh, f32x2{acc[2 * jj + 1][r0 + 4], acc[2 * jj + 1][r0 + 5]} * rs);
          const f32x2 b = geglu_pair(f32x2{acc[2 * jj][r0 + 6], acc[2 * jj][r0 + 7]} * rsh, f32x2{acc[2 * jj + 1][r0 + 6], acc[2 * jj + 1][r0 + 7]} * rs);
          v1 = f32x4{a.x, a.y, b.x, b.y};
        }
        u32x4 hi, lo;
        split8(v0, v1, hi, lo);
        hf_hi[2 * jj + hc] = __builtin_bit_cast(bf16x8, hi);
        hf_lo[2 * jj + hc] = __builtin_bit_cast(bf16x8, lo);
      }
    // vector-written fragments -> operands of asm MFMAs: no hazard padding from the compiler.  The pad names every fragment as an operand:
    // a bare `s_nop` is ordered against nothing, and the scheduler put the pack of chunk 0's first dword behind it (wrong hidden features
    // 0, 1, 4, 5 of every tile's first chunk on the first try)
    asm volatile("s_nop 7" : "+v"(hf_hi[0]), "+v"(hf_hi[1]), "+v"(hf_hi[2]), "+v"(hf_hi[3]), "+v"(hf_lo[0]), "+v"(hf_lo[1]), "+v"(hf_lo[2]), "+v"(hf_lo[3]));
    __builtin_amdgcn_sched_barrier(0);
    if (probe && t == 2) p.clk[10] = __builtin_amdgcn_s_memtime();

    // ================= down projection: k-steps 2 t, 2 t + 1 (64 hidden features) into the row's K output features ========================
    // stage v = 2 nt + kk: output n-tile nt (4 blocks of 32), hidden chunks 2 kk, 2 kk + 1
    static_for<NKD>([&](auto v_) {
      constexpr int v = decltype(v_)::value, ntile = v >> 1, kk = v & 1;
      const int slot = (slot0 + NKU + v) % NSTG, nslot = (slot0 + NKU + v + 1) % NSTG;
      auto dd = [&](auto b_, auto j_, bool w_lo, bool h_lo) {
        constexpr int b = decltype(b_)::value, j = decltype(j_)::value;
        const bf16x8& w = w_lo ? wl[b][j] : wh[b][j];
        const bf16x8& h = h_lo ? hf_lo[2 * kk + b] : hf_hi[2 * kk + b];
        mfma_acc_ag<AO + 16 * (4 * ntile + j)>(w, h);
      };
      constexpr std::integral_constant<int, 0> I0{};
      constexpr std::integral_constant<int, 1> I1{};
      constexpr std::integral_constant<int, 2> I2{};
      constexpr std::integral_constant<int, 3> I3{};
      dd(I0, I0, true, false);
      __builtin_amdgcn_sched_barrier(0);
      read_frags(slot, 1, wh[1], wl[1]);
      __builtin_amdgcn_sched_barrier(0);
      dd(I0, I1, true, false); dd(I0, I2, true, false); dd(I0, I3, true, false);
      dd(I0, I0, false, true); dd(I0, I1, false, true); dd(I0, I2, false, true); dd(I0, I3, false, true);
      dd(I0, I0, false, false); dd(I0, I1, false, false); dd(I0, I2, false, false); dd(I0, I3, false, false);
      __builtin_amdgcn_sched_barrier(0);
      next_stage_in();
      dd(I1, I0, true, false);
      __builtin_amdgcn_sched_barrier(0);
      read_frags(nslot, 0, wh[0], wl[0]);
      __builtin_amdgcn_sched_barrier(0);
      dd(I1, I1, true, false); dd(I1, I2, true, false);
      issue_rel(t, NKU + v + PDIST, 0);
      __builtin_amdgcn_sched_barrier(0);
      dd(I1, I3, true, false); dd(I1, I0, false, true); dd(I1, I1, false, true);
      issue_rel(t, NKU + v + PDIST, 1);
      __builtin_amdgcn_sched_barrier(0);
      dd(I1, I2, false, true); dd(I1, I3, false, true); dd(I1, I0, false, false);
      issue_rel(t, NKU + v + PDIST, 2);
      __builtin_amdgcn_sched_barrier(0);
      dd(I1, I1, false, false); dd(I1, I2, false, false);
      issue_rel(t, NKU + v + PDIST, 3);
      dd(I1, I3, false, false);
      __builtin_amdgcn_sched_barrier(0);
    });
    if (probe && t == 0) p.clk[6] = __builtin_amdgcn_s_memtime();
    if (probe && t == 2) p.clk[11] = __builtin_amdgcn_s_memtime();
  }
  if (probe) p.clk[12] = __builtin_amdgcn_s_memtime();
  // ---- + x, store: out accumulators from the AccVGPRs, the row's x read again (fp32), through the wave's store strip -------------------
  char* strip = smem + NSTG * STG + 4 * SCL + wid * 2048;
  float* st_row[2];
  const float* sk_row[2];
  bool st_ok[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int r = m0 + wid * 32 + 16 * it + (lane >> 2);
    st_ok[it] = r < p.M;
    st_row[it] = p.Y + (size_t)min(r, p.M - 1) * K + 4 * (lane & 3);
    sk_row[it] = p.X + (size_t)min(r, p.M - 1) * K + 4 * (lane & 3);
  }
  // every skip piece of the wave's rows requested up front (4 NOB loads in flight: one memory latency for the epilogue, not one per piece)
  f32x4 skip_all[NOB][2][2];
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
      for (int it = 0; it < 2; ++it)
        skip_all[ob][hb][it] = OUTP ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(sk_row[it] + 32 * ob + 16 * hb);
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 3" ::: "memory");       // tail LDS-DMA drained; last MFMA results readable
  static_for<NOB>([&](auto ob_) {
    constexpr int ob = decltype(ob_)::value;
    f32x4 blk[4];
    static_for<4>([&](auto g_) { blk[decltype(g_)::value] = areg_read4<AO + 16 * ob + 4 * decltype(g_)::value>(); });
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      const f32x4 (&skip)[2] = skip_all[ob][hb];
#pragma unroll
      for (int gg = 0; gg < 2; ++gg)
        *reinterpret_cast<f32x4*>(strip + l31 * 64 + (((2 * gg + lh) ^ ((l31 >> 2) & 1)) << 4)) = blk[2 * hb + gg];
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int r16 = 16 * it + (lane >> 2), c = lane & 3;
        const f32x4 o = *reinterpret_cast<const f32x4*>(strip + r16 * 64 + ((c ^ ((r16 >> 2) & 1)) << 4));
        if (st_ok[it]) st16(st_row[it] + 32 * ob + 16 * hb, f32x4(o + skip[it]));
      }
    }
  });
  if (probe) { p.clk[2] = __builtin_amdgcn_s_memtime(); p.clk[3] = __builtin_amdgcn_s_memrealtime(); p.clk[7] = (unsigned long long)total; }
  wg_stamp_end(wgs);
}


// ===========================================================================================================================================
// Width 128, TWO workgroups per CU.  The kernel above leaves the matrix pipe idle for ~45 % of a workgroup's life at level 0 (row prologue
// under load 12 k clocks, GEGLU 1.9 k per tile, epilogue 6.7 k, of 72 k: profiles/r03_ffn_x3_timeline.log) and nothing else is resident to
// fill it.  Two independent workgroups per CU need <= 128 ArchVGPRs + 128 AccVGPRs per wave and <= 80 KiB of LDS each:
//   * AccVGPRs a0..a63 = the row's activation fragments, a64..a127 = the 128 output accumulators (all of that half of the file);
//   * the d_ff range is walked in HALF tiles of 32 hidden features (value block + gate block: 32 accumulator registers instead of 64,
//     16 registers of hidden fragments instead of 32), straight out of the same packed images: a half tile's rows are one contiguous 4 KiB
//     run inside each [128 rows][32 k] image of pack layout 1, its down k-step one whole 16 KiB block of pack layout 2;
//   * ring of 4 stages of 16 KiB: per half tile 2 up stages (each two 32-k sub-stages [hi 4 KiB | lo 4 KiB] of the half tile's 64 rows)
//     and 1 down stage; 24 MFMAs per stage as above, one mid-stage barrier per stage, requests for stage s + 3 between the MFMAs of stage s.
// LDS 76 KiB per workgroup.  Measured (profiles/r03_ffn_x3_timeline.log): 157 -> 130 us per launch at level 0; inside the tile loop the two
// waves of a SIMD keep the matrix pipe 94 % busy (4 900 clocks per half tile and wave against 2 x 2 304 of MFMA).  The GEGLU stays packed
// fp32: beside ANOTHER wave's MFMAs v_pk_* costs nothing extra (a scalar build was level: 130.1 / 128.7 vs 130.2 / 131.0 us).
template <int IDX>
__device__ __forceinline__ void mfma_acc_ag_lo(const bf16x8 w, const bf16x8 h) {
  asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" :: "v"(w), "v"(h), "i"(IDX), "i"(IDX + 15) : KD_AGPR_LO128);
}
// output accumulators (named AccVGPRs) += W fragment x activation fragment held in AccVGPRs too
template <int IDX, int BIDX>
__device__ __forceinline__ void mfma_acc_aa_lo(const bf16x8 w) {
  asm volatile("v_mfma_f32_32x32x16_bf16 a[%c1:%c2], %0, a[%c3:%c4], a[%c1:%c2]" :: "v"(w), "i"(IDX), "i"(IDX + 15), "i"(BIDX), "i"(BIDX + 3) : KD_AGPR_LO128);
}
template <int IDX>
__device__ __forceinline__ void areg_zero16_lo() {
  static_for<16>([&](auto i_) { asm volatile("v_accvgpr_write_b32 a[%c0], 0" :: "i"(IDX + decltype(i_)::value) : KD_AGPR_LO128); });
}

// OUTP: the attention block's out projection runs first, in the same workgroup (x <- x + att Wo^T, image_transformer_v2.py:473-476 in front of
// :487-493): the wave's 32 ATTENTION rows become the B fragments (a0..a63), the output accumulators start from x (read straight into the C
// layout) and take 4 stages of Wo; what they then hold IS the new residual stream in the layout of an MFMA result -- which, with the up
// projection's weight packed in that k order (pack layout 3), is also the up projection's B operand: norm statistics, scale and hi / lo
// split happen in registers, the fragments replace the attention ones, and the block's own down projection keeps accumulating on top of
// the new x, so that neither the out projection's result nor the skip operand ever crosses HBM (per level-0 layer: 402 -> 201 MB).
template <bool OUTP>
__global__ __launch_bounds__(256, 2) void ffn_x3h_kernel(const FArgs3 p) {
  constexpr int NC = 8, K = 128, NOB = 4;
  constexpr int NSTG = 4, PDIST = NSTG - 1, PB = 4, UNIT = 3;   // stages per half tile: 2 up + 1 down
  constexpr int AO = 64;
  constexpr int CPR = K / 4, PIECES = 32 * CPR / 64, SCL = 1024, SUB = 8192, HALF = 4096;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const auto warm = code_warm_begin<16 * 1024>((int)blockIdx.x < p.warm && tid < 64);
  const int m0 = blockIdx.x * 128;
  const int T2 = 2 * p.n_tiles;                                 // half tiles
  const bool probe = p.clk && blockIdx.x == (gridDim.x * 5) / 8 && tid == 0;
  if (probe) { p.clk[0] = __builtin_amdgcn_s_memtime(); p.clk[1] = __builtin_amdgcn_s_memrealtime(); }
  const WgStamp wgs = wg_stamp_begin(p.clk);

  // stage (th, u): u < 2: sub-stages 2 u, 2 u + 1 of half tile th's up rows (piece j: sub-stage j >> 1, hi / lo image j & 1: this wave's
  // KiB of that 4 KiB run); u == 2: the down block of k-step th (this wave's 4 KiB of it).  Requests past the end repeat the last stage.
  auto issue_rel = [&](int th, auto off_, int j) {
    constexpr int off = decltype(off_)::value, u = off % UNIT;
    const int tt = min(th + off / UNIT, T2 - 1);                 // (past the end: the same kind of stage of the last half tile, never read)
    char* slot = smem + ((th * UNIT + off) % NSTG) * STG;
    const char* src;
    char* dst;
    if constexpr (u < 2) {
      src = p.Wu + ((size_t)(tt >> 1) * (NC / 2) + 2 * u + (j >> 1)) * STG + (j & 1) * IMG + (tt & 1) * HALF + wid * 1024 + lane * 16;
      dst = slot + (j >> 1) * SUB + (j & 1) * HALF + wid * 1024;
    } else {
      src = p.Wd + (size_t)tt * STG + wid * (PB * 1024) + j * 1024 + lane * 16;
      dst = slot + wid * (PB * 1024) + j * 1024;
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };

  // ---- this wave's 32 rows -> a0..a63 (hi / lo fragments of the normalised, scaled row; OUTP: of the attention row as it is) --------------
  const int row = m0 + wid * 32 + l31;
  const bool ok = row < p.M;
  const int rowc = ok ? row : p.M - 1;
  float rs = 1.f;
  f32x4 xres[OUTP ? NOB : 1][4];                       // OUTP: x of the lane's row in the C layout (features 32 ob + 8 g + 4 lh + 0..3)
  if constexpr (OUTP) {
    const float* xr = p.X + (size_t)rowc * K + 4 * lh;
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
      for (int g = 0; g < 4; ++g) xres[ob][g] = *reinterpret_cast<const f32x4*>(xr + 32 * ob + 8 * g);
  }
  {
    char* stage = smem + wid * STG;
    char* scl = smem + NSTG * STG + wid * SCL;
    const int r_first = min(m0 + wid * 32, p.M - 1), r_last = min(m0 + wid * 32 + 31, p.M - 1);
    const bool uni = p.scale_stride == 0 || r_first / p.rows_per_sample == r_last / p.rows_per_sample;
    if (uni) {
      const char* ssrc = reinterpret_cast<const char*>(p.scale + (size_t)(r_first / p.rows_per_sample) * p.scale_stride);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ssrc + (lane * 16) % (K * 4)),
                                       (__attribute__((address_space(3))) void*)scl, 16, 0, 0);
    }
    const float* sp = p.scale + (size_t)(rowc / p.rows_per_sample) * p.scale_stride + 8 * lh;
    const float* spl = reinterpret_cast<const float*>(scl) + 8 * lh;
    float ssq = 0.f;
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const int ci = i * 64 + lane, rr = ci / CPR, qs = ci % CPR;
      const int grow = min(m0 + wid * 32 + rr, p.M - 1);
      const char* src = reinterpret_cast<const char*>((OUTP ? p.Att : p.X) + (size_t)grow * K) + ((qs ^ (rr & 15)) << 4);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(stage + i * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const char* rowp = stage + l31 * (K * 4);
    static_for<NC / 4>([&](auto c4_) {
      constexpr int c0 = 4 * decltype(c4_)::value;
      f32x4 x0[4], x1[4], s0[4], s1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = 4 * (c0 + u) + 2 * lh;
        x0[u] = *reinterpret_cast<const f32x4*>(rowp + ((q ^ (l31 & 15)) << 4));
        x1[u] = *reinterpret_cast<const f32x4*>(rowp + (((q + 1) ^ (l31 & 15)) << 4));
        if constexpr (!OUTP) {
          if (uni) {
            s0[u] = *reinterpret_cast<const f32x4*>(spl + 16 * (c0 + u));
            s1[u] = *reinterpret_cast<const f32x4*>(spl + 16 * (c0 + u) + 4);
          } else {
            s0[u] = *reinterpret_cast<const f32x4*>(sp + 16 * (c0 + u));
            s1[u] = *reinterpret_cast<const f32x4*>(sp + 16 * (c0 + u) + 4);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      static_for<4>([&](auto u_) {
        constexpr int u = decltype(u_)::value;
        u32x4 hi, lo;
        if constexpr (OUTP) {
          split8(x0[u], x1[u], hi, lo);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) ssq = fmaf(x0[u][e], x0[u][e], fmaf(x1[u][e], x1[u][e], ssq));
          split8(x0[u] * s0[u], x1[u] * s1[u], hi, lo);
        }
        areg_write4_lo<8 * (c0 + u)>(hi);
        areg_write4_lo<8 * (c0 + u) + 4>(lo);
      });
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (!OUTP) {
      ssq += __shfl_xor(ssq, 32, 64);
      rs = rsqrtf(ssq / (float)K + p.eps);
    }
  }
  if constexpr (OUTP) {                                // the output accumulators start from x
    static_for<NOB>([&](auto ob_) {
      static_for<4>([&](auto g_) {
        constexpr int ob = decltype(ob_)::value, gq = decltype(g_)::value;
        areg_write4_lo<AO + 16 * ob + 4 * gq>(__builtin_bit_cast(u32x4, xres[ob][gq]));
      });
    });
  } else {
    static_for<NOB>([&](auto ob_) { areg_zero16_lo<AO + 16 * decltype(ob_)::value>(); });
  }
  code_warm_end(warm);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  KD_BARRIER();                                        // every wave has taken its rows out of the slot it borrowed
  // OUTP: the stream starts with the 4 stages of Wo ([128 out rows][32 k] hi | lo, plain layout); the block's own stages follow in the same
  // ring positions as without it (4 stages = once round the ring)
  auto issue_wo = [&](int q, int j) {
    const char* src = p.Wo + (size_t)q * STG + wid * (PB * 1024) + j * 1024 + lane * 16;
    char* dst = smem + (q % NSTG) * STG + wid * (PB * 1024) + j * 1024;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };
  if constexpr (OUTP) {
#pragma unroll
    for (int q = 0; q < PDIST; ++q)
#pragma unroll
      for (int j = 0; j < PB; ++j) issue_wo(q, j);
  } else {
    static_for<PDIST>([&](auto s_) {
#pragma unroll
      for (int j = 0; j < PB; ++j) issue_rel(0, s_, j);
    });
  }
  if (probe) p.clk[4] = __builtin_amdgcn_s_memtime();

  const int o0 = swz64(l31, lh), o1 = swz64(l31, 2 + lh);
  f32x16 acc[2];
  bf16x8 uh[2][2], ul[2][2];                           // up: [chunk parity][value / gate block]
  bf16x8 dh[2][4], dl[2][4];                           // down: [hidden chunk][output block]
  // chunk cc (0..3) of an up stage: sub-stage cc >> 1, 16-k chunk cc & 1 of it; rows 32 j + l31 of the half tile's 64
  auto read_up = [&](int slot, int cc, bf16x8 (&fh)[2], bf16x8 (&fl)[2]) {
    const char* st = smem + slot * STG + (cc >> 1) * SUB + ((cc & 1) ? o1 : o0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      fh[j] = *reinterpret_cast<const bf16x8*>(st + j * 32 * 64);
      fl[j] = *reinterpret_cast<const bf16x8*>(st + HALF + j * 32 * 64);
    }
  };
  auto read_dn = [&](int slot, int h, bf16x8 (&fh)[4], bf16x8 (&fl)[4]) {
    const char* st = smem + slot * STG + (h ? o1 : o0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      fh[j] = *reinterpret_cast<const bf16x8*>(st + j * 32 * 64);
      fl[j] = *reinterpret_cast<const bf16x8*>(st + IMG + j * 32 * 64);
    }
  };
  auto next_stage_in = [&]() {
    wait_vm(PB * (PDIST - 2));
    KD_BARRIER();
  };
  constexpr std::integral_constant<int, 0> I0{};
  constexpr std::integral_constant<int, 1> I1{};
  constexpr std::integral_constant<int, 2> I2{};
  constexpr std::integral_constant<int, 3> I3{};
  wait_vm(PB * (PDIST - 1));
  KD_BARRIER();
  if constexpr (OUTP) {
    // ================= out projection: x (in the output accumulators) += att Wo^T, 4 stages of 24 MFMAs ===================================
    read_dn(0, 0, dh[0], dl[0]);
    static_for<4>([&](auto q_) {
      constexpr int q = decltype(q_)::value;
      auto oo = [&](auto b_, auto j_, bool w_lo, auto al_) {
        constexpr int b = decltype(b_)::value, j = decltype(j_)::value, al = decltype(al_)::value, c = 2 * q + b;
        mfma_acc_aa_lo<AO + 16 * j, 8 * c + 4 * al>(w_lo ? dl[b][j] : dh[b][j]);
      };
      auto request = [&](int j) {                      // stage q + 3 of the stream: Wo's last stage, then the block's first three
        if constexpr (q == 0) issue_wo(3, j);
        else issue_rel(0, std::integral_constant<int, (q > 0 ? q - 1 : 0)>{}, j);
      };
      oo(I0, I0, true, I0);
      __builtin_amdgcn_sched_barrier(0);
      read_dn(q % NSTG, 1, dh[1], dl[1]);
      __builtin_amdgcn_sched_barrier(0);
      oo(I0, I1, true, I0); oo(I0, I2, true, I0); oo(I0, I3, true, I0);
      oo(I0, I0, false, I1); oo(I0, I1, false, I1); oo(I0, I2, false, I1); oo(I0, I3, false, I1);
      oo(I0, I0, false, I0); oo(I0, I1, false, I0); oo(I0, I2, false, I0); oo(I0, I3, false, I0);
      __builtin_amdgcn_sched_barrier(0);
      next_stage_in();
      oo(I1, I0, true, I0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (q < 3) read_dn((q + 1) % NSTG, 0, dh[0], dl[0]);
      else read_up(0, 0, uh[0], ul[0]);
      __builtin_amdgcn_sched_barrier(0);
      oo(I1, I1, true, I0); oo(I1, I2, true, I0);
      request(0);
      __builtin_amdgcn_sched_barrier(0);
      oo(I1, I3, true, I0); oo(I1, I0, false, I1); oo(I1, I1, false, I1);
      request(1);
      __builtin_amdgcn_sched_barrier(0);
      oo(I1, I2, false, I1); oo(I1, I3, false, I1); oo(I1, I0, false, I0);
      request(2);
      __builtin_amdgcn_sched_barrier(0);
      oo(I1, I1, false, I0); oo(I1, I2, false, I0);
      request(3);
      oo(I1, I3, false, I0);
      __builtin_amdgcn_sched_barrier(0);
    });
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");  // asm MFMA results -> v_accvgpr_read
    // ---- new x (C layout, still in the accumulators) -> AdaRMSNorm statistics, scale, hi / lo fragments in the k order of pack layout 3:
    // chunk 2 ob + hc of the row = registers 8 hc .. + 7 of output block ob -----------------------------------------------------------
    {
      const char* scl = smem + NSTG * STG + wid * SCL;
      const int r_first = min(m0 + wid * 32, p.M - 1), r_last = min(m0 + wid * 32 + 31, p.M - 1);
      const bool uni = p.scale_stride == 0 || r_first / p.rows_per_sample == r_last / p.rows_per_sample;
      const float* sg = uni ? nullptr : p.scale + (size_t)(rowc / p.rows_per_sample) * p.scale_stride + 4 * lh;
      const float* sl = reinterpret_cast<const float*>(scl) + 4 * lh;
      float ssq = 0.f;
      static_for<NOB>([&](auto ob_) {
        constexpr int ob = decltype(ob_)::value;
        f32x4 v[4], sc[4];
        static_for<4>([&](auto g_) { v[decltype(g_)::value] = areg_read4<AO + 16 * ob + 4 * decltype(g_)::value>(); });
#pragma unroll
        for (int g = 0; g < 4; ++g) sc[g] = uni ? *reinterpret_cast<const f32x4*>(sl + 32 * ob + 8 * g) : *reinterpret_cast<const f32x4*>(sg + 32 * ob + 8 * g);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int e = 0; e < 4; ++e) ssq = fmaf(v[g][e], v[g][e], ssq);
        static_for<2>([&](auto hc_) {
          constexpr int hc = decltype(hc_)::value;
          u32x4 hi, lo;
          split8(v[2 * hc] * sc[2 * hc], v[2 * hc + 1] * sc[2 * hc + 1], hi, lo);
          areg_write4_lo<8 * (2 * ob + hc)>(hi);
          areg_write4_lo<8 * (2 * ob + hc) + 4>(lo);
        });
      });
      ssq += __shfl_xor(ssq, 32, 64);
      rs = rsqrtf(ssq / (float)K + p.eps);
    }
  } else {
    read_up(0, 0, uh[0], ul[0]);
  }
  const float rsh = 0.5f * rs;
  for (int th = 0; th < T2; ++th) {
    const int slot0 = (th * UNIT) % NSTG;
    if (probe && th == 4) p.clk[8] = __builtin_amdgcn_s_memtime();
    // ================= up projection of the half tile: value / gate accumulators of its 32 hidden features (2 stages of 4 chunks) ==========
    static_for<2>([&](auto u_) {
      constexpr int u = decltype(u_)::value;
      const int slot = (slot0 + u) % NSTG, nslot = (slot0 + u + 1) % NSTG;
      // the 6 MFMAs of chunk cc: (w_lo x a_hi), (w_hi x a_lo), (w_hi x a_hi) for the value block and the gate block in alternation
      auto mm = [&](auto cc_, auto i_) {
        constexpr int cc = decltype(cc_)::value, i = decltype(i_)::value, j = i & 1, term = i >> 1, c = 4 * u + cc;
        const bf16x8& w = term == 0 ? ul[cc & 1][j] : uh[cc & 1][j];
        constexpr int al = term == 1;
        if constexpr (c == 0 && term == 0) mfma_ag0<8 * c + 4 * al>(acc[j], w);      // first MFMA of the chain: C = 0
        else mfma_ag<8 * c + 4 * al>(acc[j], w);
      };
      constexpr std::integral_constant<int, 4> I4{};
      constexpr std::integral_constant<int, 5> I5{};
      // chunk 0
      mm(I0, I0);
      __builtin_amdgcn_sched_barrier(0);
      read_up(slot, 1, uh[1], ul[1]);
      __builtin_amdgcn_sched_barrier(0);
      mm(I0, I1); mm(I0, I2); mm(I0, I3); mm(I0, I4); mm(I0, I5);
      // chunk 1
      mm(I1, I0);
      __builtin_amdgcn_sched_barrier(0);
      read_up(slot, 2, uh[0], ul[0]);
      __builtin_amdgcn_sched_barrier(0);
      mm(I1, I1); mm(I1, I2); mm(I1, I3); mm(I1, I4); mm(I1, I5);
      __builtin_amdgcn_sched_barrier(0);
      next_stage_in();
      // chunk 2
      mm(I2, I0);
      __builtin_amdgcn_sched_barrier(0);
      read_up(slot, 3, uh[1], ul[1]);
      __builtin_amdgcn_sched_barrier(0);
      mm(I2, I1); mm(I2, I2);
      issue_rel(th, std::integral_constant<int, u + PDIST>{}, 0);
      __builtin_amdgcn_sched_barrier(0);
      mm(I2, I3); mm(I2, I4); mm(I2, I5);
      issue_rel(th, std::integral_constant<int, u + PDIST>{}, 1);
      __builtin_amdgcn_sched_barrier(0);
      // chunk 3
      mm(I3, I0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (u == 0) read_up(nslot, 0, uh[0], ul[0]);
      else read_dn(nslot, 0, dh[0], dl[0]);
      __builtin_amdgcn_sched_barrier(0);
      mm(I3, I1); mm(I3, I2);
      issue_rel(th, std::integral_constant<int, u + PDIST>{}, 2);
      __builtin_amdgcn_sched_barrier(0);
      mm(I3, I3); mm(I3, I4);
      issue_rel(th, std::integral_constant<int, u + PDIST>{}, 3);
      mm(I3, I5);
      __builtin_amdgcn_sched_barrier(0);
    });
    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[0]), "+v"(acc[1]));      // asm MFMA results -> vector reads
    if (probe && th == 4) p.clk[9] = __builtin_amdgcn_s_memtime();

    // ================= GEGLU in the lane that owns the row -> hi / lo fragments of the half tile's 2 hidden 16-chunks =====================
    bf16x8 hf_hi[2], hf_lo[2];
#pragma unroll
    for (int hc = 0; hc < 2; ++hc) {
      f32x4 v0, v1;
      const int r0 = 8 * hc;
      {
        const f32x2 a = geglu_pair(f32x2{acc[0][r0], acc[0][r0 + 1]} * rsh, f32x2{acc[1][r0], acc[1][r0 + 1]} * rs);
        const f32x2 b = geglu_pair(f32x2{acc[0][r0 + 2], acc[0][r0 + 3]} * rsh, f32x2{acc[1][r0 + 2], acc[1][r0 + 3]} * rs);
        v0 = f32x4{a.x, a.y, b.x, b.y};
      }
      {
        const f32x2 a = geglu_pair(f32x2{acc[0][r0 + 4], acc[0][r0 + 5]} * rsh, f32x2{acc[1][r0 + 4], acc[1][r0 + 5]} * rs);
        const f32x2 b = geglu_pair(f32x2{acc[0][r0 + 6], acc[0][r0 + 7]} * rsh, f32x2{acc[1][r0 + 6], acc[1][r0 + 7]} * rs);
        v1 = f32x4{a.x, a.y, b.x, b.y};
      }
      u32x4 hi, lo;
      split8(v0, v1, hi, lo);
      hf_hi[hc] = __builtin_bit_cast(bf16x8, hi);
      hf_lo[hc] = __builtin_bit_cast(bf16x8, lo);
    }
    asm volatile("s_nop 7" : "+v"(hf_hi[0]), "+v"(hf_hi[1]), "+v"(hf_lo[0]), "+v"(hf_lo[1]));   // vector-written fragments -> asm MFMA operands
    __builtin_amdgcn_sched_barrier(0);
    if (probe && th == 4) p.clk[10] = __builtin_amdgcn_s_memtime();

    // ================= down projection: the half tile's 32 hidden features into the row's 128 output features (1 stage) ===================
    {
      const int slot = (slot0 + 2) % NSTG, nslot = (slot0 + 3) % NSTG;
      auto dd = [&](auto b_, auto j_, bool w_lo, bool h_lo) {
        constexpr int b = decltype(b_)::value, j = decltype(j_)::value;
        const bf16x8& w = w_lo ? dl[b][j] : dh[b][j];
        const bf16x8& h = h_lo ? hf_lo[b] : hf_hi[b];
        mfma_acc_ag_lo<AO + 16 * j>(w, h);
      };
      dd(I0, I0, true, false);
      __builtin_amdgcn_sched_barrier(0);
      read_dn(slot, 1, dh[1], dl[1]);
      __builtin_amdgcn_sched_barrier(0);
      dd(I0, I1, true, false); dd(I0, I2, true, false); dd(I0, I3, true, false);
      dd(I0, I0, false, true); dd(I0, I1, false, true); dd(I0, I2, false, true); dd(I0, I3, false, true);
      dd(I0, I0, false, false); dd(I0, I1, false, false); dd(I0, I2, false, false); dd(I0, I3, false, false);
      __builtin_amdgcn_sched_barrier(0);
      next_stage_in();
      dd(I1, I0, true, false);
      __builtin_amdgcn_sched_barrier(0);
      read_up(nslot, 0, uh[0], ul[0]);
      __builtin_amdgcn_sched_barrier(0);
      dd(I1, I1, true, false); dd(I1, I2, true, false);
      issue_rel(th, std::integral_constant<int, 2 + PDIST>{}, 0);
      __builtin_amdgcn_sched_barrier(0);
      dd(I1, I3, true, false); dd(I1, I0, false, true); dd(I1, I1, false, true);
      issue_rel(th, std::integral_constant<int, 2 + PDIST>{}, 1);
      __builtin_amdgcn_sched_barrier(0);
      dd(I1, I2, false, true); dd(I1, I3, false, true); dd(I1, I0, false, false);
      issue_rel(th, std::integral_constant<int, 2 + PDIST>{}, 2);
      __builtin_amdgcn_sched_barrier(0);
      dd(I1, I1, false, false); dd(I1, I2, false, false);
      issue_rel(th, std::integral_constant<int, 2 + PDIST>{}, 3);
      dd(I1, I3, false, false);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (probe && th == 4) p.clk[11] = __builtin_amdgcn_s_memtime();
  }
  if (probe) p.clk[12] = __builtin_amdgcn_s_memtime();

  // ---- + x, store (as above) -------------------------------------------------------------------------------------------------------------
  char* strip = smem + NSTG * STG + 4 * SCL + wid * 2048;
  float* st_row[2];
  const float* sk_row[2];
  bool st_ok[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int r = m0 + wid * 32 + 16 * it + (lane >> 2);
    st_ok[it] = r < p.M;
    st_row[it] = p.Y + (size_t)min(r, p.M - 1) * K + 4 * (lane & 3);
    sk_row[it] = p.X + (size_t)min(r, p.M - 1) * K + 4 * (lane & 3);
  }
  f32x4 skip_all[NOB][2][2];                           // (OUTP: the accumulators started from x: nothing to add)
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
      for (int it = 0; it < 2; ++it)
        skip_all[ob][hb][it] = OUTP ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(sk_row[it] + 32 * ob + 16 * hb);
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 3" ::: "memory");       // tail LDS-DMA drained; last MFMA results readable
  static_for<NOB>([&](auto ob_) {
    constexpr int ob = decltype(ob_)::value;
    f32x4 blk[4];
    static_for<4>([&](auto g_) { blk[decltype(g_)::value] = areg_read4<AO + 16 * ob + 4 * decltype(g_)::value>(); });
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      const f32x4 (&skip)[2] = skip_all[ob][hb];
#pragma unroll
      for (int gg = 0; gg < 2; ++gg)
        *reinterpret_cast<f32x4*>(strip + l31 * 64 + (((2 * gg + lh) ^ ((l31 >> 2) & 1)) << 4)) = blk[2 * hb + gg];
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int r16 = 16 * it + (lane >> 2), c = lane & 3;
        const f32x4 o = *reinterpret_cast<const f32x4*>(strip + r16 * 64 + ((c ^ ((r16 >> 2) & 1)) << 4));
        if (st_ok[it]) st16(st_row[it] + 32 * ob + 16 * hb, f32x4(o + skip[it]));
      }
    }
  });
  if (probe) { p.clk[2] = __builtin_amdgcn_s_memtime(); p.clk[3] = __builtin_amdgcn_s_memrealtime(); p.clk[7] = (unsigned long long)(T2 * UNIT); }
  wg_stamp_end(wgs);
}

extern unsigned long long* g_clk;      // gemm_x3.hip (kd_prof_clock_buffer)

template <bool OUTP>
static int launch_ffn_half(const FArgs3& a, const char* nm, double flops, double bytes, hipStream_t s) {
  constexpr int LDS = 4 * STG + 4 * 1024 + 4 * 2048;
  auto kern = ffn_x3h_kernel<OUTP>;
  static LdsAttr attr_set;
  attr_set.ensure(reinterpret_cast<const void*>(kern), LDS);
  LaunchScope prof(nm, flops, bytes, s);
  hipLaunchKernelGGL(kern, dim3((unsigned)((a.M + 127) / 128)), dim3(256), LDS, s, a);
  return check_launch("kd_ffn_f32");
}

template <int NC, bool OUTP = false>
static int launch_ffn(const FArgs3& a, const char* nm, double flops, double bytes, hipStream_t s) {
  auto kern = ffn_x3_kernel<NC, OUTP>;
  constexpr int K = NC * 16;
  constexpr int LDS = 8 * STG + 4 * (K * 4 < 1024 ? 1024 : K * 4) + 4 * 2048;
  static LdsAttr attr_set;
  attr_set.ensure(reinterpret_cast<const void*>(kern), LDS);
  LaunchScope prof(nm, flops, bytes, s);
  hipLaunchKernelGGL(kern, dim3((unsigned)((a.M + 127) / 128)), dim3(256), LDS, s, a);
  return check_launch("kd_ffn_f32");
}

}  // namespace x3
}  // namespace kd

using namespace kd;

extern "C" int kd_ffn_f32_supported(int M, int K, int d_ff) {
  if (!option("ffn_x3", 1)) return 0;
  if (!((K == 128 || K == 256) && d_ff > 0 && d_ff % 64 == 0)) return 0;
  // Where it is the FASTER form.  Width 128 (two workgroups per CU, ~55 us per workgroup): from 16 row panels on.  Width 256 runs ONE workgroup per CU
  // for ~120 us whatever the grid, so below a chip-filling grid the two-launch form (GEGLU projection with n-splits + residual projection, both of which
  // spread over the CUs) wins: at batch 4 of the headline config (32 panels) the fused block took 95 us where the pair takes ~35
  // (profiles/r04_small_batch.log).  Option "ffn_x3_min_panels_256": panels needed at width 256 (default: 7/8 of the CUs).
  if (K == 128) return M >= 2048;
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  const int need = option("ffn_x3_min_panels_256", cus - cus / 8);
  return (M + 127) / 128 >= need;
}

extern "C" int kd_ffn_f32(const KdFfn* dp, void* stream) {
  if (!dp) return fail(KD_EINVAL, "kd_ffn_f32: null descriptor");
  const KdFfn& d = *dp;
  if (!d.x || !d.out || !d.scale || !d.Wp_up || !d.Wp_down) return fail(KD_EINVAL, "kd_ffn_f32: null x / out / scale / Wp_up / Wp_down");
  if ((d.K != 128 && d.K != 256) || d.d_ff <= 0 || (d.d_ff & 63) || d.M <= 0) return fail(KD_EINVAL, "kd_ffn_f32: K must be 128 or 256, d_ff a multiple of 64 (K=%d d_ff=%d)", d.K, d.d_ff);
  if (d.rows_per_sample <= 0 || (d.scale_stride & 3)) return fail(KD_EINVAL, "kd_ffn_f32: rows_per_sample > 0, scale_stride %% 4 == 0");
  x3::FArgs3 a{};
  a.X = reinterpret_cast<const float*>(d.x); a.Y = reinterpret_cast<float*>(d.out);
  a.Wu = reinterpret_cast<const char*>(d.Wp_up); a.Wd = reinterpret_cast<const char*>(d.Wp_down);
  a.scale = d.scale; a.scale_stride = d.scale_stride; a.rows_per_sample = d.rows_per_sample; a.eps = d.eps;
  a.M = d.M; a.d_ff = d.d_ff; a.n_tiles = d.d_ff / 64;
  a.warm = option("code_warm", KD_CODE_WARM_DEFAULT);
  a.clk = x3::g_clk;
  const double flops = 2.0 * d.M * 3.0 * d.d_ff * d.K;
  const double bytes = 4.0 * (2.0 * d.M * d.K + 3.0 * d.d_ff * d.K);
  char nm[96] = "ffn_x3";
  if (prof_on()) snprintf(nm, sizeof(nm), "ffn_x3 M=%d K=%d d_ff=%d", d.M, d.K, d.d_ff);
  if (d.attn) {
    if (!d.Wp_out) return fail(KD_EINVAL, "kd_ffn_f32: attn without Wp_out");
    a.Att = reinterpret_cast<const float*>(d.attn); a.Wo = reinterpret_cast<const char*>(d.Wp_out);
    const double fl2 = flops + 2.0 * d.M * d.K * d.K, by2 = 4.0 * (3.0 * d.M * d.K + 3.0 * d.d_ff * d.K + (double)d.K * d.K);
    if (prof_on()) snprintf(nm, sizeof(nm), "ffn_x3+out M=%d K=%d d_ff=%d", d.M, d.K, d.d_ff);
    if (d.K == 128) return x3::launch_ffn_half<true>(a, nm, fl2, by2, (hipStream_t)stream);
    return x3::launch_ffn<16, true>(a, nm, fl2, by2, (hipStream_t)stream);
  }
  if (d.K == 128 && option("ffn_x3_half", 1)) return x3::launch_ffn_half<false>(a, nm, flops, bytes, (hipStream_t)stream);
  if (d.K == 128) return x3::launch_ffn<8>(a, nm, flops, bytes, (hipStream_t)stream);
  return x3::launch_ffn<16>(a, nm, flops, bytes, (hipStream_t)stream);
}

KD_TEXT_PAD(ffn_x3)      // last function of this code object: kd_common.h, code warm-up
