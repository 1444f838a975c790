// fp32-parity ("split3") projections at FEW ROWS (batch 1 - 4 of the headline config, the small configs): the latency form.
//
// The throughput kernels (gemm_x3.hip, gemm_x3r.hip) give a workgroup 128 rows and a serial chain per launch: a row prologue that
// normalises and splits 128 x K activations (~10 us at K = 512), then a barrier-per-stage K loop over 128 x 128 tiles (~0.5 us per 32 k:
// 25 us at K = 1536).  With 256 rows in a launch that chain is all there is -- 8 .. 24 workgroups on a 256-CU chip, 27 - 35 us per
// projection where the arithmetic is worth < 1 us (profiles/r04_small_batch.log).  This kernel cuts the chain instead of the work:
//   * a workgroup owns 32 rows x ONE HALF TILE of the packed weight (64 W rows = one q / k / v head vector, or 32 GEGLU outputs: the
//     contiguous 4 KiB run of each [128 rows][32 k] image, as gemm_x3h / ffn_x3h walk them), so M = 256, N = 1536 is 192 workgroups;
//   * its 8 waves split K: wave w takes the 32-k stages w, w + 8, ...; a stage's operands -- the lane's own 8 + 8 activations (and
//     scales) of its row, its W fragments -- are plain 16-byte loads straight into registers in the layout the MFMA wants (the packed
//     image's swizzle is just an address), two stages in flight per wave: no LDS ring, no barrier, no prologue;
//   * the AdaRMSNorm row factor is applied in the epilogue (as gemm_x3.hip does: the products run on x * scale), its sum of squares
//     collected by the waves as they stream the row;
//   * the 8 partial accumulators meet in LDS: every wave writes its 32 registers, wave q sums register group q of all eight IN WAVE
//     ORDER (bit-reproducible), wave 0 takes the sums back and runs the epilogue of gemm_x3.hip on them (cosine-sim + RoPE + split
//     store, GEGLU, residual, plain store).  The residual enters as wave 7's initial accumulators.
// Same arithmetic as the throughput kernels (3 bf16 MFMA terms per product, fp32 accumulate) in another summation order.
#include "x3_common.h"
#include <cstdio>

namespace kd {
namespace x3s {

using namespace x3;
using b16::u16;

constexpr int HALF = 4096, NW = 8;
constexpr int RED_BYTES = NW * 8 * 64 * 16;             // [wave][register group][lane] x 16 bytes: one row block's partial sums
constexpr int SCL_MAX_K = 2048, SCL_BYTES = SCL_MAX_K * 4;
template <int RB> constexpr int lds_bytes() { return RED_BYTES + RB * (NW * 32 * 4 + 8 * 64 * 16 + 2048) + SCL_BYTES; }

struct SArgs {
  const float* A; const char* Wp; float* C; const float* R;
  u16* Cl; int c_split;
  const float* scale; int scale_stride, rows_per_sample; float eps;
  int M, N, K, nk;
  int n_heads; const float* qk_scale; const float* pos; const float* freq; int qkv_packed;
  float out_add;
};

// SC: 0 no norm in front; 1 AdaRMSNorm, every lane loads its row's scale entries (rows of several samples in a workgroup); 2 AdaRMSNorm, all
// rows of the workgroup share one scale vector: parked in LDS once, read as broadcasts (a quarter less through the L1 than SC = 1)
template <int RB, int SC>
struct Stage {
  f32x4 x[RB][2][2];           // the lane's rows: k = 16 c + 8 lh .. + 7 of the stage
  f32x4 s[SC == 1 ? 2 : 1][2]; // SC = 1: the scale vector's entries there
  bf16x8 wh[2][2], wl[2][2];   // W fragments [16-k chunk c][32-row block j], hi / lo images
};

template <int EPI, int SC, int RB>
__global__ __launch_bounds__(512) void gemm_x3s_kernel(const SArgs p) {
  constexpr bool GEGLU = EPI == KD_EPI_GEGLU, NORM = SC != 0;
  constexpr int HCOL = GEGLU ? 32 : 64;                          // output columns of a half tile
  static_assert(RB == 1 || SC != 1, "two row blocks per wave: no registers left for per-lane scale entries");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* red = reinterpret_cast<f32x4*>(smem);                                     // [wave][register group q = 4 j + g][lane]
  float* ssqp = reinterpret_cast<float*>(smem + RED_BYTES);                        // [row block][wave][row]
  f32x4* red2 = reinterpret_cast<f32x4*>(smem + RED_BYTES + RB * NW * 32 * 4);     // [row block][q][lane]
  char* strips = smem + RED_BYTES + RB * (NW * 32 * 4 + 8 * 64 * 16);              // [row block] 2 KiB store strip of its epilogue wave
  const float* scl = reinterpret_cast<const float*>(strips + RB * 2048);           // SC = 2: the sample's scale vector
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ht = blockIdx.x, m0 = blockIdx.y * (32 * RB), n0 = ht * HCOL;
  const int nk = p.nk;
  const int n_my = wid < nk ? (nk - wid + NW - 1) / NW : 0;     // stages wid, wid + 8, ...
  int rowc[RB];
  bool ok[RB];
  const float* ap[RB];
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    const int row = m0 + 32 * r + l31;
    ok[r] = row < p.M;
    rowc[r] = ok[r] ? row : p.M - 1;
    ap[r] = p.A + (size_t)rowc[r] * p.K + 8 * lh;
  }
  const float* sp = SC == 1 ? p.scale + (size_t)(rowc[0] / p.rows_per_sample) * p.scale_stride + 8 * lh : nullptr;
  const char* wp = p.Wp + (size_t)(ht >> 1) * nk * STG + (ht & 1) * HALF;
  const int o0 = swz64(l31, lh), o1 = swz64(l31, 2 + lh);
  // (the 16-byte loads of a stage are issued in THIS order everywhere -- a sched_barrier behind each: the compiler's counted waits are per
  // register and merged over the paths into a block, so one order keeps them as tight as the program is)
#define KD_PIN() __builtin_amdgcn_sched_barrier(0)
  using Stg = Stage<RB, SC>;
  auto load = [&](Stg& b, int ks) {
    const char* w = wp + (size_t)ks * STG;
    KD_PIN();
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const float* a = ap[r] + ks * 32;
        b.x[r][c][0] = *reinterpret_cast<const f32x4*>(a + 16 * c); KD_PIN();
        b.x[r][c][1] = *reinterpret_cast<const f32x4*>(a + 16 * c + 4); KD_PIN();
      }
      if constexpr (SC == 1) {
        const float* sc = sp + ks * 32;
        b.s[c][0] = *reinterpret_cast<const f32x4*>(sc + 16 * c); KD_PIN();
        b.s[c][1] = *reinterpret_cast<const f32x4*>(sc + 16 * c + 4); KD_PIN();
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const char* f = w + (c ? o1 : o0) + j * 32 * 64;
        b.wl[c][j] = *reinterpret_cast<const bf16x8*>(f + IMG); KD_PIN();
        b.wh[c][j] = *reinterpret_cast<const bf16x8*>(f); KD_PIN();
      }
    }
  };

  // SC = 2: the scale vector of the workgroup's sample, requested AHEAD of the stage loads (its wait then leaves them in flight)
  f32x4 sv4 = {0.f, 0.f, 0.f, 0.f};
  if constexpr (SC == 2) {
    const float* sbase = p.scale + (size_t)(min(m0, p.M - 1) / p.rows_per_sample) * p.scale_stride;
    if (4 * tid < p.K) sv4 = *reinterpret_cast<const f32x4*>(sbase + 4 * tid);
    KD_PIN();
  }

  f32x16 acc[RB][2];
#pragma unroll
  for (int r = 0; r < RB; ++r)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[r][j][e] = 0.f;
  // KD_EPI_RESIDUAL: C = R + A W^T.  The residual is the last wave's initial partial sum, read straight into the C layout (lane (l31, lh),
  // block j, register 4 g + e <-> row l31, feature 32 j + 8 g + 4 lh + e): in flight behind that wave's stage loads, no epilogue adds
  if constexpr (EPI == KD_EPI_RESIDUAL) {
    if (wid == NW - 1) {
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const float* rp = p.R + (size_t)rowc[r] * p.N + n0 + 4 * lh;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(rp + 32 * j + 8 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[r][j][4 * g + e] = v[e];
          }
      }
    }
  }
  // wave r runs the epilogue of row block r: its few per-row / per-head operands are requested here, ahead of the stream
  float py = 0.f, px = 0.f, qsc = 1.f;
  f32x4 fv = {0.f, 0.f, 0.f, 0.f};
  int which = 2, head = 0;
  if constexpr (EPI == KD_EPI_QKV) {
    which = ht >= 2 * p.n_heads ? 2 : (ht >= p.n_heads ? 1 : 0);
    head = ht - which * p.n_heads;
    if (wid < RB && which < 2) {
      const int tok = min(m0 + 32 * wid + l31, p.M - 1) % p.rows_per_sample;
      py = p.pos[2 * tok];
      px = p.pos[2 * tok + 1];
      fv = *reinterpret_cast<const f32x4*>(p.freq + head * 8 + 4 * lh);
      qsc = sqrtf(p.qk_scale[head]);
    }
  }

  float ssq[RB];
#pragma unroll
  for (int r = 0; r < RB; ++r) ssq[r] = 0.f;
  auto compute = [&](const Stg& b, int ks) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      f32x4 s0, s1;
      if constexpr (SC == 1) { s0 = b.s[c][0]; s1 = b.s[c][1]; }
      if constexpr (SC == 2) {
        const float* sc = scl + ks * 32 + 16 * c + 8 * lh;
        s0 = *reinterpret_cast<const f32x4*>(sc);
        s1 = *reinterpret_cast<const f32x4*>(sc + 4);
      }
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        f32x4 v0 = b.x[r][c][0], v1 = b.x[r][c][1];
        if constexpr (NORM) {
#pragma unroll
          for (int e = 0; e < 4; ++e) ssq[r] = fmaf(v0[e], v0[e], fmaf(v1[e], v1[e], ssq[r]));
          v0 = v0 * s0;
          v1 = v1 * s1;
        }
        u32x4 hi, lo;
        split8(v0, v1, hi, lo);
        const bf16x8 a_hi = __builtin_bit_cast(bf16x8, hi), a_lo = __builtin_bit_cast(bf16x8, lo);
#pragma unroll
        for (int j = 0; j < 2; ++j) mfma_a(acc[r][j], b.wl[c][j], a_hi);
#pragma unroll
        for (int j = 0; j < 2; ++j) mfma_a(acc[r][j], b.wh[c][j], a_lo);
#pragma unroll
        for (int j = 0; j < 2; ++j) mfma_a(acc[r][j], b.wh[c][j], a_hi);
        if constexpr (RB > 1) KD_PIN();      // (one row block's split pieces at a time: at 64 rows the registers are all spoken for)
      }
    }
  };

  // ---- the wave's stages, two in flight.  Every path is straight-line up to the loop: a load behind a branch would make the compiler's
  // wait counting merge "issued" with "not issued" and wait for everything outstanding at the next use (seen: vmcnt(0) in the loop's first
  // compute()).  sched_barrier: left alone, the instruction scheduler moves both refills behind the second compute() and the next
  // iteration waits for all of them at its top -- a round trip to L2 per pair of stages with nothing in flight ---------------------------------
  auto park_scale = [&]() {       // (every wave, also one without stages: the barrier is the workgroup's)
    if constexpr (SC == 2) {
      if (4 * tid < p.K) *reinterpret_cast<f32x4*>(const_cast<float*>(scl) + 4 * tid) = sv4;
      __syncthreads();
    }
  };
  if (n_my >= 2) {
    Stg b0, b1;
    load(b0, wid);
    load(b1, wid + NW);
    __builtin_amdgcn_sched_barrier(0);
    park_scale();
    __builtin_amdgcn_sched_barrier(0);
    int i = 0;
    for (; i + 3 < n_my; i += 2) {
      __builtin_amdgcn_sched_barrier(0);
      compute(b0, wid + i * NW);
      __builtin_amdgcn_sched_barrier(0);
      load(b0, wid + (i + 2) * NW);
      __builtin_amdgcn_sched_barrier(0);
      compute(b1, wid + (i + 1) * NW);
      __builtin_amdgcn_sched_barrier(0);
      load(b1, wid + (i + 3) * NW);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (n_my - i == 3) {
      compute(b0, wid + i * NW);
      __builtin_amdgcn_sched_barrier(0);
      load(b0, wid + (i + 2) * NW);
      __builtin_amdgcn_sched_barrier(0);
      compute(b1, wid + (i + 1) * NW);
      __builtin_amdgcn_sched_barrier(0);
      compute(b0, wid + (i + 2) * NW);
    } else {                                   // two left
      compute(b0, wid + i * NW);
      __builtin_amdgcn_sched_barrier(0);
      compute(b1, wid + (i + 1) * NW);
    }
  } else if (n_my == 1) {
    Stg b0;
    load(b0, wid);
    __builtin_amdgcn_sched_barrier(0);
    park_scale();
    __builtin_amdgcn_sched_barrier(0);
    compute(b0, wid);
  } else {
    park_scale();
  }

  // ---- the 8 partial sums of a row block -> its epilogue wave (row block r: wave r), one row block after the other through `red` ------------
  if constexpr (NORM) {
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      ssq[r] += __shfl_xor(ssq[r], 32, 64);
      if (lh == 0) ssqp[(r * NW + wid) * 32 + l31] = ssq[r];
    }
  }
#pragma unroll
  for (int r = 0; r < RB; ++r) {
#pragma unroll
    for (int q = 0; q < 8; ++q)
      red[(wid * 8 + q) * 64 + lane] = f32x4{acc[r][q >> 2][4 * (q & 3)], acc[r][q >> 2][4 * (q & 3) + 1], acc[r][q >> 2][4 * (q & 3) + 2], acc[r][q >> 2][4 * (q & 3) + 3]};
    __syncthreads();
    f32x4 sum = red[wid * 64 + lane];                      // wave q = wid: register group q of wave 0, then 1 .. 7 on top, in that order
#pragma unroll
    for (int w = 1; w < NW; ++w) sum = sum + red[(w * 8 + wid) * 64 + lane];
    red2[(r * 8 + wid) * 64 + lane] = sum;
    __syncthreads();
  }
  if (wid >= RB) return;
  const int rb = wid;                                      // this wave's row block
  f32x16 out[2];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const f32x4 v = red2[(rb * 8 + q) * 64 + lane];
#pragma unroll
    for (int e = 0; e < 4; ++e) out[q >> 2][4 * (q & 3) + e] = v[e];
  }
  float rs = 1.f;
  if constexpr (NORM) {
    float t = ssqp[(rb * NW) * 32 + l31];
#pragma unroll
    for (int w = 1; w < NW; ++w) t += ssqp[(rb * NW + w) * 32 + l31];
    rs = rsqrtf(t / (float)p.K + p.eps);
  }
  const int mb = m0 + 32 * rb;
  const bool okb = mb + l31 < p.M;
  const int rowb = okb ? mb + l31 : p.M - 1;
  char* strip = strips + rb * 2048;

  // ---- epilogue (gemm_x3.hip's, on one half tile): the lane owns row l31, features n0 + 32 j + 8 g + 4 lh + (0..3) per register group --------
  float* st_row[2];
  bool st_ok[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int r = mb + 16 * it + (lane >> 2);
    st_ok[it] = r < p.M;
    st_row[it] = p.C + (size_t)min(r, p.M - 1) * p.N + 4 * (lane & 3);
  }
  // one 32-feature block -> memory through the wave-private strip ([32 rows][16 floats], two passes): four consecutive lanes then hold 64
  // contiguous bytes of ONE row (16 requests of 64 bytes per instruction instead of 64 of 16)
  auto store_block = [&](const f32x4 (&v)[4], int col) {
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
      for (int gg = 0; gg < 2; ++gg)
        *reinterpret_cast<f32x4*>(strip + l31 * 64 + (((2 * gg + lh) ^ ((l31 >> 2) & 1)) << 4)) = v[2 * hb + gg];
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int r16 = 16 * it + (lane >> 2), c = lane & 3;
        const f32x4 o = *reinterpret_cast<const f32x4*>(strip + r16 * 64 + ((c ^ ((r16 >> 2) & 1)) << 4));
        if (st_ok[it]) st16(st_row[it] + col + 16 * hb, o);
      }
    }
  };
  if constexpr (GEGLU) {
    const float rsh = 0.5f * rs;
    f32x4 blk[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x2 a = geglu_pair(f32x2{out[0][4 * g], out[0][4 * g + 1]} * rsh, f32x2{out[1][4 * g], out[1][4 * g + 1]} * rs);
      const f32x2 b = geglu_pair(f32x2{out[0][4 * g + 2], out[0][4 * g + 3]} * rsh, f32x2{out[1][4 * g + 2], out[1][4 * g + 3]} * rs);
      blk[g] = f32x4{a.x, a.y, b.x, b.y};
    }
    if (p.c_split) {                // the down projection's A operand as bf16 hi / lo planes (gemm_x3t.hip)
      float v[16], hi[16], lo[16];
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int q = 0; q < 4; ++q) v[4 * g + q] = blk[g][q];
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const unsigned h = pack_bf16(v[r], v[r + 1]);
        hi[r] = b16::bf_lo(h);
        hi[r + 1] = b16::bf_hi(h);
        lo[r] = v[r] - hi[r];
        lo[r + 1] = v[r + 1] - hi[r + 1];
      }
      const size_t off = (size_t)rowb * p.N + n0;
      b16::store_block_bf16(reinterpret_cast<u16*>(p.C) + off, hi, lh, okb);
      b16::store_block_bf16(p.Cl + off, lo, lh, okb);
    } else {
      store_block(blk, n0);
    }
  } else if constexpr (EPI == KD_EPI_QKV) {
    if (which < 2) {
      const float fr[4] = {fv[0], fv[1], fv[2], fv[3]};
      b16::qk_prep_blocks(out[0], out[1], rs, qsc, p.eps, py, px, fr);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) { out[0][r] *= rs; out[1][r] *= rs; }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      f32x4 blk[4];
      if (p.qkv_packed) {
#pragma unroll
        for (int g = 0; g < 4; ++g) blk[g] = pack_split4(f32x4{out[j][4 * g], out[j][4 * g + 1], out[j][4 * g + 2], out[j][4 * g + 3]});
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) blk[g] = f32x4{out[j][4 * g], out[j][4 * g + 1], out[j][4 * g + 2], out[j][4 * g + 3]};
      }
      store_block(blk, n0 + 32 * j);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      f32x4 blk[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) blk[g] = f32x4{out[j][4 * g], out[j][4 * g + 1], out[j][4 * g + 2], out[j][4 * g + 3]} * rs + p.out_add;
      store_block(blk, n0 + 32 * j);
    }
  }
}


template <int EPI, int SC, int RB>
static int launch(const SArgs& a, const char* nm, double flops, double bytes, hipStream_t s) {
  auto kern = gemm_x3s_kernel<EPI, SC, RB>;
  static LdsAttr attr_set;
  attr_set.ensure(reinterpret_cast<const void*>(kern), lds_bytes<RB>());
  const int hcol = EPI == KD_EPI_GEGLU ? 32 : 64;
  LaunchScope prof(nm, flops, bytes, s);
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.N / hcol), (unsigned)((a.M + 32 * RB - 1) / (32 * RB))), dim3(64 * NW), lds_bytes<RB>(), s, a);
  return check_launch("kd_gemm_f32(x3s)");
}
// (RB = 2, 64 rows per workgroup with every W fragment feeding two row blocks -- a third less through the L1 per row, half the workgroups --
// needs 64 accumulators next to 2 x 64 registers of stages in flight: at the 256 registers of two waves per SIMD hipcc spills 50 - 280 of
// them to scratch.  Not instantiated.)
template <int EPI>
static int launch_norm(const SArgs& a, int sc, const char* nm, double flops, double bytes, hipStream_t s) {
  return sc == 2 ? launch<EPI, 2, 1>(a, nm, flops, bytes, s) : launch<EPI, 1, 1>(a, nm, flops, bytes, s);
}

}  // namespace x3s

// Eligibility + dispatch (called by kd_gemm_f32 ahead of the throughput kernels).  Returns 1 if the descriptor was not taken.
// Taken: split3 projections on plain fp32 rows with at most `x3s_max_rows` rows (option; 0 turns the kernel off), K a multiple of 32, N a
// multiple of the half tile (64; GEGLU: 32 outputs): store (with or without the norm in front), norm -> qkv + cosine-sim + RoPE (positions
// form; any head count -- a workgroup holds one head vector), norm -> GEGLU (fp32 or hi / lo planes), residual projection.
int gemm_x3s_try(const GemmP& d, hipStream_t s, int* rc) {
  using namespace x3s;
  // Where it pays: per-kernel tables of the headline config at batch 1 / 2 / 4 / 8 (benchmarks/batch1_table.sh, profiles/r04_small_batch.log).
  // A launch of this kernel costs about 5 us + rounds x (1.6 + K / 130) us (+ 1 per round with a qkv / GEGLU epilogue): a workgroup's operands
  // are (64 + 32) x K x 4 bytes, re-read from L2 / the memory-side cache by every workgroup of its row / column, two stages in flight per
  // wave.  The throughput kernels cost about 22 us (K = 512) / 13.5 us (K <= 256) behind a norm while their grid is a few panels, and
  // 7 + tile rounds x K x 0.014 us without one.  So this form wins while its grid is one round of the chip (M = 1024, N = 512, K = 1536: 18.5
  // against gemm_x3r's 26.9 us; M = 256 qkv: 10.4 against 34.8) and at two rounds where K is short or the rows are few (M = 256 GEGLU of
  // 1536: 17.1 against 29.4), and loses beyond (M = 2048, N = 512, K = 1536 in two rounds: 34.0 against 29.1; M = 1024 GEGLU of 768 in three:
  // 17.4 against 13.9).  Option "x3s_max_wgs" replaces the estimate by a cap on the grid (tests, A/B runs).
  const int max_rows = option("x3s_max_rows", 4096);
  if (d.M > max_rows) return 1;
  if (d.precision != KD_PREC_SPLIT3 || d.a_mode != KD_A_PLAIN || !d.Wp || d.debug || d.a_split) return 1;
  if ((d.K & 31) || d.K < 64) return 1;
  const bool geglu = d.epi == KD_EPI_GEGLU;
  if (d.N % (geglu ? 32 : 64)) return 1;
  if (d.c_split && (!geglu || !d.C_lo)) return 1;
  if (d.out_add != 0.f && d.epi != KD_EPI_STORE) return 1;
  // variant (option x3s_scale_lds, off): the scale vector through LDS where a workgroup's 32 rows share it.  A quarter less through the L1,
  // but measured level with the per-lane loads (17.7 / 12.7 / 10.6 against 17.1 / 12.8 / 10.3 us): the kernel waits on latency, not on the L1
  const int rps = d.rows_per_sample > 0 ? d.rows_per_sample : d.M;
  const int sc = !d.norm ? 0 : (((d.scale_stride == 0 || rps % 32 == 0) && d.K <= SCL_MAX_K && option("x3s_scale_lds", 0)) ? 2 : 1);
  const int cus = cu_count();
  const long wgs = (long)((d.M + 31) / 32) * (d.N / (geglu ? 32 : 64));
  const int cap = option("x3s_max_wgs", -1);
  if (cap >= 0) {
    if (wgs > cap) return 1;
  } else {
    const bool heavy_epi = d.epi == KD_EPI_QKV || geglu;
    const double mine = 5.0 + (double)((wgs + cus - 1) / cus) * (1.6 + d.K / 130.0 + (heavy_epi ? 1.0 : 0.0));
    const long tiles = (long)((d.M + 127) / 128) * ((d.N + 127) / 128);
    const double other = d.norm ? (d.K >= 512 ? 22.0 : 13.5) : 7.0 + (double)((tiles + cus - 1) / cus) * d.K * 0.0141;
    if (mine * 1.15 >= other) return 1;
  }
  SArgs a{};
  a.A = d.A; a.Wp = reinterpret_cast<const char*>(d.Wp); a.C = d.C; a.R = d.R;
  a.Cl = reinterpret_cast<u16*>(d.C_lo); a.c_split = d.c_split;
  a.scale = d.norm ? d.scale : nullptr; a.scale_stride = d.scale_stride; a.rows_per_sample = rps; a.eps = d.eps;
  a.M = d.M; a.N = d.N; a.K = d.K; a.nk = d.K / 32;
  a.n_heads = d.n_heads; a.qk_scale = d.qk_scale; a.pos = d.rope_pos; a.freq = d.rope_freq; a.qkv_packed = d.qkv_packed;
  a.out_add = d.out_add;
  const double n_eff = geglu ? 2.0 * d.N : (double)d.N;
  const double flops = 2.0 * d.M * n_eff * d.K;
  const double bytes = 4.0 * ((double)d.M * d.K + n_eff * d.K + (double)d.M * d.N * (d.epi == KD_EPI_RESIDUAL ? 2 : 1));
  char nm[96] = "gemm_x3s";
  if (prof_on()) snprintf(nm, sizeof(nm), "gemm_x3s<n%d,e%d> M=%d N=%d K=%d", sc, d.epi, d.M, d.N, d.K);
  if (option("x3s_trace", 0)) {                 // debugging aid (kd_set_option): one line per launch on stderr, the stream drained in front of it
    fprintf(stderr, "x3s: epi=%d norm=%d M=%d N=%d K=%d rps=%d stride=%d heads=%d packed=%d c_split=%d A=%p Wp=%p C=%p R=%p scale=%p pos=%p freq=%p\n", d.epi, d.norm, d.M, d.N, d.K,
            a.rows_per_sample, a.scale_stride, a.n_heads, a.qkv_packed, a.c_split, (const void*)a.A, (const void*)a.Wp, (void*)a.C, (const void*)a.R, (const void*)a.scale, (const void*)a.pos, (const void*)a.freq);
    (void)hipStreamSynchronize(s);
  }
  if (d.epi == KD_EPI_STORE && d.norm) *rc = launch_norm<KD_EPI_STORE>(a, sc, nm, flops, bytes, s);
  else if (d.epi == KD_EPI_STORE) *rc = launch<KD_EPI_STORE, 0, 1>(a, nm, flops, bytes, s);
  else if (d.epi == KD_EPI_RESIDUAL && !d.norm && d.R) *rc = launch<KD_EPI_RESIDUAL, 0, 1>(a, nm, flops, bytes, s);
  else if (d.epi == KD_EPI_QKV && d.norm && d.rope_pos && d.rope_freq) *rc = launch_norm<KD_EPI_QKV>(a, sc, nm, flops, bytes, s);
  else if (geglu && d.norm) *rc = launch_norm<KD_EPI_GEGLU>(a, sc, nm, flops, bytes, s);
  else return 1;
  return 0;
}

}  // namespace kd

// Not a code-warm-up user, but its code object ends like those of the kernels that are (kd_common.h): 36 KiB of s_nop behind the last kernel,
// so that an instruction fetch running ahead of a wave's last instructions stays inside the loaded image whatever the loader put behind it.
KD_TEXT_PAD(gemm_x3s)
