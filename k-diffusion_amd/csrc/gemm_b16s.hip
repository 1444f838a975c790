// bf16-mode projections at FEW ROWS (batch 1 - 2 of the headline config): the latency form, the bf16 sibling of gemm_x3s.hip.
//
// At 256 rows the bf16 throughput kernels have a handful of workgroups and a serial chain each (the generic kernel took 23 - 24 us for the
// level-2 qkv / GEGLU projections at batch 1, the tiled kernel 15 us for the K = 1536 down projection: profiles/r04_small_batch.log).  Here
// a workgroup owns 32 rows x ONE HALF TILE of the packed weight (64 W rows of a [128 rows][64 k] block: one q / k / v head vector, or 32 GEGLU
// outputs), its 8 waves split K (wave w: the 64-k blocks w, w + 8, ...), a block's operands -- the lane's own 4 x 8 bf16 activations, their
// fp32 scale entries, its 8 W fragments -- are plain 16-byte loads straight into registers in MFMA layout (the packed image's swizzle is only
// an address), two blocks in flight per wave; the AdaRMSNorm row factor is applied in the epilogue (as gemm_bf16.hip's A-stationary kernel
// does: the products run on bf16(x * scale)), its sum of squares collected while streaming; the 8 partial accumulators meet in LDS, wave q sums
// register group q of all eight in wave order (bit-reproducible), wave 0 runs the epilogue; the residual enters as wave 7's initial
// accumulators.  One bf16 MFMA per product, fp32 accumulation, one rounding at the store: the arithmetic of the throughput kernels in another
// summation order.
#include "bf16_common.h"
#include <atomic>

namespace kd {
namespace b16s {

using namespace b16;

constexpr int HALFB = 8192, NW = 8;                  // bytes of 64 W rows of a block; waves per workgroup
constexpr int RED_BYTES = NW * 8 * 64 * 16, SSQ_BYTES = NW * 32 * 4, RED2_BYTES = 8 * 64 * 16;
constexpr int LDS_BYTES = RED_BYTES + SSQ_BYTES + RED2_BYTES;

struct SArgs {
  const u16* A; const char* Wp; u16* C; const u16* R;
  const float* scale; int scale_stride, rows_per_sample; float eps;
  int M, N, K, nk;
  int n_heads; const float* qk_scale; const float* pos; const float* freq;
};

template <bool NORM>
struct Block {
  u32x4 x[4];                  // the lane's row: k = 16 t + 8 lh .. + 7 of the 64-k block, bf16 pairs
  f32x4 s[NORM ? 4 : 1][2];    // the scale vector's entries there
  bf16x8 w[4][2];              // W fragments [16-k chunk t][32-row block j]
};

#define KD_PIN() __builtin_amdgcn_sched_barrier(0)

template <int EPI, bool NORM>
__global__ __launch_bounds__(512) void gemm_b16s_kernel(const SArgs p) {
  constexpr bool GEGLU = EPI == KD_EPI_GEGLU;
  constexpr int HCOL = GEGLU ? 32 : 64;                          // output columns of a half tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ht = blockIdx.x, m0 = blockIdx.y * 32, n0 = ht * HCOL;
  const int row = m0 + l31;
  const bool ok = row < p.M;
  const int rowc = ok ? row : p.M - 1;
  const int nk = p.nk;
  const int n_my = wid < nk ? (nk - wid + NW - 1) / NW : 0;     // blocks wid, wid + 8, ...

  const u16* ap = p.A + (size_t)rowc * p.K + 8 * lh;
  const float* sp = NORM ? p.scale + (size_t)(rowc / p.rows_per_sample) * p.scale_stride + 8 * lh : nullptr;
  const char* wp = p.Wp + (size_t)(ht >> 1) * nk * WBLK + (ht & 1) * HALFB;
  int off[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) off[t] = swz128(l31, 2 * t + lh);
  // (the 16-byte loads of a block are issued in THIS order everywhere -- a sched_barrier behind each: the compiler's counted waits are per
  // register and merged over the paths into a basic block, so one order keeps them as tight as the program is; see gemm_x3s.hip)
  using Blk = Block<NORM>;
  auto load = [&](Blk& b, int ks) {
    const u16* a = ap + ks * 64;
    const char* w = wp + (size_t)ks * WBLK;
    KD_PIN();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      b.x[t] = *reinterpret_cast<const u32x4*>(a + 16 * t); KD_PIN();
      if constexpr (NORM) {
        const float* sc = sp + ks * 64 + 16 * t;
        b.s[t][0] = *reinterpret_cast<const f32x4*>(sc); KD_PIN();
        b.s[t][1] = *reinterpret_cast<const f32x4*>(sc + 4); KD_PIN();
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        b.w[t][j] = *reinterpret_cast<const bf16x8*>(w + j * 32 * 128 + off[t]); KD_PIN();
      }
    }
  };

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  // KD_EPI_RESIDUAL: C = R + A W^T.  The residual (bf16) is the last wave's initial partial sum, brought into the C layout as fp32
  if constexpr (EPI == KD_EPI_RESIDUAL) {
    if (wid == NW - 1) {
      const u16* rrow = p.R + (size_t)rowc * p.N + n0;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float r[16];
        load_block_bf16(rrow + 32 * j, r, lh);
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = r[e];
      }
    }
  }
  // wave 0 runs the epilogue: its few per-row / per-head operands are requested here, ahead of the stream
  float py = 0.f, px = 0.f, qsc = 1.f;
  f32x4 fv = {0.f, 0.f, 0.f, 0.f};
  int which = 2, head = 0;
  if constexpr (EPI == KD_EPI_QKV) {
    which = ht >= 2 * p.n_heads ? 2 : (ht >= p.n_heads ? 1 : 0);
    head = ht - which * p.n_heads;
    if (wid == 0 && which < 2) {
      const int tok = rowc % p.rows_per_sample;
      py = p.pos[2 * tok];
      px = p.pos[2 * tok + 1];
      fv = *reinterpret_cast<const f32x4*>(p.freq + head * 8 + 4 * lh);
      qsc = sqrtf(p.qk_scale[head]);
    }
  }

  float ssq = 0.f;
  auto compute = [&](const Blk& b) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      bf16x8 a;
      if constexpr (NORM) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { x[2 * e] = bf_lo(b.x[t][e]); x[2 * e + 1] = bf_hi(b.x[t][e]); }
#pragma unroll
        for (int e = 0; e < 8; ++e) ssq = fmaf(x[e], x[e], ssq);
        const u32x4 o = {pack_bf16(x[0] * b.s[t][0][0], x[1] * b.s[t][0][1]), pack_bf16(x[2] * b.s[t][0][2], x[3] * b.s[t][0][3]),
                         pack_bf16(x[4] * b.s[t][1][0], x[5] * b.s[t][1][1]), pack_bf16(x[6] * b.s[t][1][2], x[7] * b.s[t][1][3])};
        a = __builtin_bit_cast(bf16x8, o);
      } else {
        a = __builtin_bit_cast(bf16x8, b.x[t]);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b.w[t][j], a, acc[j], 0, 0, 0);
    }
  };

  // ---- the wave's blocks, two in flight; every path straight-line up to the loop, issue order pinned (gemm_x3s.hip has the why) -------------
  if (n_my >= 2) {
    Blk b0, b1;
    load(b0, wid);
    load(b1, wid + NW);
    KD_PIN();
    int i = 0;
    for (; i + 3 < n_my; i += 2) {
      KD_PIN();
      compute(b0);
      KD_PIN();
      load(b0, wid + (i + 2) * NW);
      KD_PIN();
      compute(b1);
      KD_PIN();
      load(b1, wid + (i + 3) * NW);
      KD_PIN();
    }
    if (n_my - i == 3) {
      compute(b0);
      KD_PIN();
      load(b0, wid + (i + 2) * NW);
      KD_PIN();
      compute(b1);
      KD_PIN();
      compute(b0);
    } else {                                   // two left
      compute(b0);
      KD_PIN();
      compute(b1);
    }
  } else if (n_my == 1) {
    Blk b0;
    load(b0, wid);
    compute(b0);
  }

  // ---- the 8 partial sums -> wave 0 -----------------------------------------------------------------------------------------------------------
  f32x4* red = reinterpret_cast<f32x4*>(smem);                             // [wave][register group q = 4 j + g][lane]
  float* ssqp = reinterpret_cast<float*>(smem + RED_BYTES);                // [wave][row]
  f32x4* red2 = reinterpret_cast<f32x4*>(smem + RED_BYTES + SSQ_BYTES);    // [q][lane]
#pragma unroll
  for (int q = 0; q < 8; ++q) red[(wid * 8 + q) * 64 + lane] = f32x4{acc[q >> 2][4 * (q & 3)], acc[q >> 2][4 * (q & 3) + 1], acc[q >> 2][4 * (q & 3) + 2], acc[q >> 2][4 * (q & 3) + 3]};
  if constexpr (NORM) {
    ssq += __shfl_xor(ssq, 32, 64);
    if (lh == 0) ssqp[wid * 32 + l31] = ssq;
  }
  __syncthreads();
  {
    f32x4 sum = red[wid * 64 + lane];                      // wave q = wid: register group q of wave 0, then 1 .. 7 on top, in that order
#pragma unroll
    for (int w = 1; w < NW; ++w) sum = sum + red[(w * 8 + wid) * 64 + lane];
    red2[wid * 64 + lane] = sum;
  }
  __syncthreads();
  if (wid != 0) return;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const f32x4 v = red2[q * 64 + lane];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[q >> 2][4 * (q & 3) + e] = v[e];
  }
  float rs = 1.f;
  if constexpr (NORM) {
    float t = ssqp[l31];
#pragma unroll
    for (int w = 1; w < NW; ++w) t += ssqp[w * 32 + l31];
    rs = rsqrtf(t / (float)p.K + p.eps);
  }

  // ---- epilogue (gemm_bf16.hip's A-stationary one, on one half tile): the lane owns row l31; one rounding to bf16 at the store ----------------
  u16* crow = p.C + (size_t)rowc * p.N + n0;
  if constexpr (GEGLU) {
    float v[16];
    const float rsh = 0.5f * rs;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const f32x2 o = geglu_pair(f32x2{acc[0][r], acc[0][r + 1]} * rsh, f32x2{acc[1][r], acc[1][r + 1]} * rs);
      v[r] = o.x;
      v[r + 1] = o.y;
    }
    store_block_bf16(crow, v, lh, ok);
  } else if constexpr (EPI == KD_EPI_QKV) {
    if (which < 2) {
      const float fr[4] = {fv[0], fv[1], fv[2], fv[3]};
      qk_prep_blocks(acc[0], acc[1], rs, qsc, p.eps, py, px, fr);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[0][r] *= rs; acc[1][r] *= rs; }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = acc[j][r];
      store_block_bf16(crow + 32 * j, v, lh, ok);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = acc[j][r] * rs;
      store_block_bf16(crow + 32 * j, v, lh, ok);
    }
  }
}


template <int EPI, bool NORM>
static int launch(const SArgs& a, const char* nm, double flops, double bytes, hipStream_t s) {
  auto kern = gemm_b16s_kernel<EPI, NORM>;
  static LdsAttr attr_set;
  attr_set.ensure(reinterpret_cast<const void*>(kern), LDS_BYTES);
  const int hcol = EPI == KD_EPI_GEGLU ? 32 : 64;
  LaunchScope prof(nm, flops, bytes, s);
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.N / hcol), (unsigned)((a.M + 31) / 32)), dim3(64 * NW), LDS_BYTES, s, a);
  return check_launch("kd_gemm_bf16(b16s)");
}

}  // namespace b16s

namespace b16 {

// Eligibility + dispatch (called by kd_gemm_bf16 ahead of the throughput kernels).  Returns 1 if the descriptor was not taken.
// Taken: plain bf16 rows, K a multiple of 64, N a multiple of the half tile (64; GEGLU: 32 outputs): norm -> qkv / GEGLU / store, store,
// residual projection -- where the grid is at most one round of the chip, or two at a few hundred rows behind a norm (there the throughput
// side is the generic kernel at 23 - 24 us; elsewhere it is already at 9 - 15 us and this form only wins in one round).
// Options: "b16s_max_rows" (4096; 0 = off), "b16s_max_wgs" (-1; >= 0: a cap on the grid instead of the rule).
int gemm_b16s_try(const KdGemm& d, hipStream_t s, int* rc) {
  using namespace b16s;
  if (d.M > option("b16s_max_rows", 4096)) return 1;
  if (d.precision != KD_PREC_BF16 || d.a_mode != KD_A_PLAIN || !d.Wp || d.a_split || d.c_split) return 1;
  if ((d.K & 63) || d.out_add != 0.f) return 1;
  const bool geglu = d.epi == KD_EPI_GEGLU;
  if (d.N % (geglu ? 32 : 64)) return 1;
  const int cus = cu_count();
  const long wgs = (long)((d.M + 31) / 32) * (d.N / (geglu ? 32 : 64));
  const int cap = option("b16s_max_wgs", -1);
  if (wgs > (cap >= 0 ? cap : ((d.norm && d.M <= 512) ? 2 * cus : cus))) return 1;
  SArgs a{};
  a.A = reinterpret_cast<const u16*>(d.A); a.Wp = reinterpret_cast<const char*>(d.Wp); a.C = reinterpret_cast<u16*>(d.C);
  a.R = reinterpret_cast<const u16*>(d.R);
  a.scale = d.norm ? d.scale : nullptr; a.scale_stride = d.scale_stride; a.rows_per_sample = d.rows_per_sample > 0 ? d.rows_per_sample : d.M; a.eps = d.eps;
  a.M = d.M; a.N = d.N; a.K = d.K; a.nk = d.K / 64;
  a.n_heads = d.n_heads; a.qk_scale = d.qk_scale; a.pos = d.rope_pos; a.freq = d.rope_freq;
  const double n_eff = geglu ? 2.0 * d.N : (double)d.N;
  const double flops = 2.0 * d.M * n_eff * d.K;
  const double bytes = 2.0 * ((double)d.M * d.K + n_eff * d.K + (double)d.M * d.N * (d.epi == KD_EPI_RESIDUAL ? 2 : 1));
  char nm[96] = "gemm_bf16_few_rows";
  if (prof_on()) snprintf(nm, sizeof(nm), "gemm_bf16_few_rows<n%d,e%d> M=%d N=%d K=%d", d.norm ? 1 : 0, d.epi, d.M, d.N, d.K);
  if (d.epi == KD_EPI_STORE) *rc = d.norm ? launch<KD_EPI_STORE, true>(a, nm, flops, bytes, s) : launch<KD_EPI_STORE, false>(a, nm, flops, bytes, s);
  else if (d.epi == KD_EPI_RESIDUAL && !d.norm && d.R) *rc = launch<KD_EPI_RESIDUAL, false>(a, nm, flops, bytes, s);
  else if (d.epi == KD_EPI_QKV && d.norm && d.rope_pos && d.rope_freq) *rc = launch<KD_EPI_QKV, true>(a, nm, flops, bytes, s);
  else if (geglu && d.norm) *rc = launch<KD_EPI_GEGLU, true>(a, nm, flops, bytes, s);
  else return 1;
  return 0;
}

}  // namespace b16
}  // namespace kd

// Not a code-warm-up user, but its code object ends like those of the kernels that are (kd_common.h): 36 KiB of s_nop behind the last kernel,
// so that an instruction fetch running ahead of a wave's last instructions stays inside the loaded image whatever the loader put behind it.
KD_TEXT_PAD(gemm_b16s)
