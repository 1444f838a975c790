// "A-stationary" split-bf16x3 GEMM for the wide projections of the HDiT levels with K <= 256
// (AdaRMSNorm -> qkv, AdaRMSNorm -> up-projection + GEGLU):  C = epilogue( norm(A) @ W^T ).
//
// Why a second GEMM kernel.  In the tiled kernel (gemm.hip) every 128x128 output tile re-reads, re-normalises and
// re-splits its A panel: with N = 3..6 x the tile width that is 3..6 x the VALU work, LDS writes and L2->LDS traffic
// of A, and at K = 128..256 the kernel is VALU/LDS bound at ~20-30 % MFMA utilisation (profiles/r01_gemm_pmc.md).
// Here a workgroup owns a 128-row panel for ALL of N:
//   * each of its 4 wave64s keeps its 32 rows of the normalised, scaled, bf16-split A panel IN REGISTERS as MFMA
//     A-operand fragments (K/16 chunks x (hi + lo) x 4 VGPRs: 64 VGPRs at K = 128, 128 at K = 256): A is read
//     from HBM exactly once and converted once;
//   * W streams through a 4-stage LDS ring as the packed, pre-swizzled bf16 image of kd_pack_weight_bf16x3
//     (16 KiB per (n-tile, 32-wide K-step)), moved by global_load_lds (no VGPRs, no ds_write pass), requested
//     three stages ahead with counted s_waitcnt vmcnt so the ring never drains;
//   * per n-tile the 32x128 accumulator block of a wave goes through the same wave-private transposing LDS strips
//     as in gemm.hip (float4 row-segment stores; q/k preparation or GEGLU in the epilogue) while the next n-tile's
//     W stages are already in flight.
// vmcnt bookkeeping: on gfx9-class hardware stores share the vector-memory counter with loads, and a counted wait
// behind the epilogue's stores would drain them.  The stages needed right after an epilogue are therefore confirmed
// BEFORE it (they were requested 2-3 stages earlier), and the first wait that can see the stores comes two stages
// later, when they have long been written.
#include "kd_common.h"
#include <cstdlib>

namespace kd {

namespace astat {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr int BM = 128;                 // rows per workgroup of the 4-wave form (4 waves x 32); the 8-wave form owns 2 * BM
constexpr int BK = 32;                  // K per W stage
constexpr int IMG = 128 * BK * 2;       // one bf16 image of a stage: [128 W rows][32 k] = 8 KiB
constexpr int STAGE = 2 * IMG;          // hi + lo = 16 KiB  (== WP_BLOCK of gemm.hip)
constexpr int NSTG = 4;                 // ring depth
constexpr int LDS_RING = NSTG * STAGE;  // 64 KiB
constexpr int lds_bytes(int nwv) { return LDS_RING + nwv * 8 * 64 * 4 + nwv * 32 * 4 + 64; }   // + epilogue strips + row rsqrt table + sqrt(qk scale) per head (<= 16)

__device__ __forceinline__ int swz(int row, int c) { return row * 64 + ((c ^ ((row >> 2) & 3)) << 4); }
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
  bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void split4(const f32x4 v, u32x2& hi, u32x2& lo) {
  hi[0] = pack_bf16(v[0], v[1]);
  hi[1] = pack_bf16(v[2], v[3]);
  lo[0] = pack_bf16(v[0] - __uint_as_float(hi[0] << 16), v[1] - __uint_as_float(hi[0] & 0xFFFF0000u));
  lo[1] = pack_bf16(v[2] - __uint_as_float(hi[1] << 16), v[3] - __uint_as_float(hi[1] & 0xFFFF0000u));
}

#define KD_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define KD_BARRIER() asm volatile("s_barrier" ::: "memory")

// NC = K / 16 (8, 16 or 32); EPI: KD_EPI_STORE / KD_EPI_QKV / KD_EPI_GEGLU.  gridDim.y splits the n-tiles of a panel over
// several workgroups when M alone gives too few panels to fill the chip (each re-normalises the A panel: cheap next to
// its share of W).  K = 512 keeps 256 VGPRs of A fragments per lane: one workgroup per CU, unified VGPR/AGPR file.
// NWV = waves per workgroup (4 or 8): an 8-wave workgroup owns a 256-row panel fed by ONE W ring -- half the L2 -> CU
// bytes per flop of two co-resident 4-wave workgroups.  The vector-memory path delivers ~10-13 B/clk/CU even on L2 hits
// (every tiled / streamed kernel here tops out there), so bytes per flop into the CU is what the main loop pays for.
template <int NC, int EPI, int NWV>
__global__ __launch_bounds__(64 * NWV, (NC <= 8 && NWV == 4) ? 2 : 1) void gemm_astat_kernel(const GemmP p) {
  constexpr int BMW = NWV * 32;                           // rows of this workgroup's panel
  constexpr int PCS = 4 * 4 / NWV;                        // 1 KiB pieces of a W stage moved by each wave (4 or 2)
  constexpr int K = NC * 16, NK = NC / 2;                 // NK: W stages per n-tile (multiple of NSTG)
  constexpr bool GEGLU = EPI == KD_EPI_GEGLU;
  constexpr int NCOL = GEGLU ? 64 : 128;                  // output columns per n-tile
  static_assert(NK % NSTG == 0, "ring slot of a stage must be a compile-time constant");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  (void)code_warm_begin<32768>((int)blockIdx.x < p.warm && blockIdx.y == 0 && threadIdx.x < 64);   // kd_common.h: this kernel's code -> L2
  char* ring = smem;
  float* strips = reinterpret_cast<float*>(smem + LDS_RING);
  float* rs_tab = strips + NWV * 8 * 64;
  float* sq_tab = rs_tab + BMW;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int M = p.M, N = p.N;
  const int n_tiles_all = N / NCOL;
  const int nt_begin = (int)((long)n_tiles_all * blockIdx.y / gridDim.y), nt_end = (int)((long)n_tiles_all * (blockIdx.y + 1) / gridDim.y);
  const int n_tiles = nt_end - nt_begin;                  // n-tiles of this workgroup
  const int total = n_tiles * NK;                         // its W stages
  const int m0 = blockIdx.x * BMW;

  // ---- W stage s -> ring slot s % NSTG (this wave's quarter: 4 x 1 KiB, lands at slot + wid*4 KiB + lane*16) ----------
  const char* wp = reinterpret_cast<const char*>(p.Wp) + (size_t)nt_begin * NK * STAGE;
  auto issue = [&](int s) {
    const char* src = wp + (size_t)s * STAGE + wid * (PCS * 1024) + lane * 16;
    char* dst = ring + (s % NSTG) * STAGE + wid * (PCS * 1024);
#pragma unroll
    for (int j = 0; j < PCS; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + j * 1024),
                                       (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, 0, 0);
  };
  issue(0);
  issue(1);
  issue(2);

  // ---- this wave's 32 rows of A -> normalise, scale, split -> MFMA A fragments in registers ---------------------------
  // lane (row l31, half lh) holds k = 16c + 8lh .. +8 of chunk c
  bf16x8 ah[NC], al[NC];
  {
    const int row = m0 + wid * 32 + l31;
    const bool ok = row < M;
    const float* ap = p.A + (long)(ok ? row : M - 1) * K + 8 * lh;
    const float* sp = p.scale + (long)(m0 / p.rows_per_sample) * p.scale_stride + 8 * lh;
    float ssq = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      f32x4 x0 = *reinterpret_cast<const f32x4*>(ap + 16 * c), x1 = *reinterpret_cast<const f32x4*>(ap + 16 * c + 4);
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(sp + 16 * c), s1 = *reinterpret_cast<const f32x4*>(sp + 16 * c + 4);
      if (!ok) { x0 = f32x4{0.f, 0.f, 0.f, 0.f}; x1 = x0; }
      ssq += x0[0] * x0[0] + x0[1] * x0[1] + x0[2] * x0[2] + x0[3] * x0[3] + x1[0] * x1[0] + x1[1] * x1[1] + x1[2] * x1[2] + x1[3] * x1[3];
      u32x2 h0, l0, h1, l1;
      split4(x0 * s0, h0, l0);
      split4(x1 * s1, h1, l1);
      ah[c] = __builtin_bit_cast(bf16x8, u32x4{h0[0], h0[1], h1[0], h1[1]});
      al[c] = __builtin_bit_cast(bf16x8, u32x4{l0[0], l0[1], l1[0], l1[1]});
    }
    ssq += __shfl_xor(ssq, 32, 64);
    if (lh == 0) rs_tab[wid * 32 + l31] = rsqrtf(ssq / (float)K + p.eps);
    if (EPI == KD_EPI_QKV && tid < p.n_heads) sq_tab[tid] = sqrtf(p.qk_scale[tid]);
  }
  const bool full_panel = m0 + BMW <= M && !(p.debug & 32);   // (debug bit 32: keep the conservative wait, for A/B runs)    // every row of the panel exists: every epilogue store is issued
  const int tok0 = EPI == KD_EPI_QKV ? (m0 + wid * 32) % p.rows_per_sample : 0;   // token of this wave's first row (panels never straddle samples)
  __syncthreads();       // rs_tab visible (also drains this wave's first W stages: they are needed next anyway)
  float rsv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) rsv[r] = rs_tab[wid * 32 + mfma32_row(r, lane)];

  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  float* strip = strips + wid * (8 * 64);

  // ---- epilogue of n-tile nt: acc (32 rows x 128 W-rows) -> strips -> global ------------------------------------------------
  auto epilogue = [&](int nt) {
    const int n0 = (nt_begin + nt) * NCOL;
    constexpr int NPASS = GEGLU ? 4 : 8;                       // (8-row group g, 64-column half) passes
    // qkv: the 64 columns of a pass are ONE (q|k|v, head) vector of a row.  The RoPE table chunks of pass i+1 are
    // requested while pass i goes through its strip (one pass ahead: 16 VGPRs), so that their latency -- and the
    // in-order wait behind the W stages still in flight -- is paid once per n-tile instead of once per pass.
    auto qk_of = [&](int half, int& which, int& head) {
      const int vec = (n0 + half * 64) >> 6;
      which = vec / p.n_heads;
      head = vec - which * p.n_heads;
    };
    auto load_tab = [&](int pass, f32x4 (&cs)[2], f32x4 (&sn)[2]) {
      const int g = GEGLU ? pass : pass >> 1, half = GEGLU ? 0 : pass & 1;
      int which, head;
      qk_of(half, which, head);
      if (which < 2) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int row8 = (lane + 64 * t) >> 4;
          const int tok = min(tok0 + 8 * g + row8, p.rows_per_sample - 1);
          const long tr = ((long)tok * p.n_heads + head) * KD_ROT + 4 * (lane & 3);   // lanes >= 8 of a row group re-read chunk c & 3, unused
          cs[t] = *reinterpret_cast<const f32x4*>(p.rope_cos + tr);
          sn[t] = *reinterpret_cast<const f32x4*>(p.rope_sin + tr);
        }
      }
    };
    f32x4 csA[2], snA[2], csB[2], snB[2];
    if (EPI == KD_EPI_QKV) load_tab(0, csA, snA);
    auto one_pass = [&](int pass, f32x4 (&cs)[2], f32x4 (&sn)[2], f32x4 (&cs_next)[2], f32x4 (&sn_next)[2]) {
      const int g = GEGLU ? pass : pass >> 1, half = GEGLU ? 0 : pass & 1;
      const int row_base = m0 + wid * 32 + 8 * g;
      if (EPI == KD_EPI_QKV && pass + 1 < NPASS) load_tab(pass + 1, cs_next, sn_next);
      // registers of rows 8g..8g+7 -> strip[8][64]
      if (GEGLU) {
#pragma unroll
        for (int q = 0; q < 4; q += 2) {                  // accumulator rows r, r+1 are adjacent registers: packed fp32 math
          const int r = 4 * g + q, row8 = q + 4 * lh;
          const f32x2 rs2 = {rsv[r], rsv[r + 1]};
          const f32x2 rsh = rs2 * 0.5f;
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const f32x2 o = geglu_pair(f32x2{acc[2 * hh][r], acc[2 * hh][r + 1]} * rsh, f32x2{acc[2 * hh + 1][r], acc[2 * hh + 1][r + 1]} * rs2);
            strip[row8 * 64 + 32 * hh + l31] = o.x;
            strip[(row8 + 1) * 64 + 32 * hh + l31] = o.y;
          }
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = 4 * g + q, row8 = q + 4 * lh;
          strip[row8 * 64 + l31] = acc[2 * half][r] * rsv[r];
          strip[row8 * 64 + 32 + l31] = acc[2 * half + 1][r] * rsv[r];
        }
      }
      int which = 2, head = 0;
      if (EPI == KD_EPI_QKV) qk_of(half, which, head);
      // strip -> global: lane owns (row8, 4 columns), 2 items per lane
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int idx = lane + 64 * t, row8 = idx >> 4, c4 = (idx & 15) * 4;
        const int gm = row_base + row8, gn = n0 + half * 64 + c4;
        f32x4 v = *reinterpret_cast<const f32x4*>(strip + row8 * 64 + c4);
        if (EPI == KD_EPI_QKV && which < 2) v = prep_row16_regs(v, lane & 15, sq_tab[head], cs[t], sn[t], p.eps);
        if (EPI == KD_EPI_STORE) v = v + p.out_add;
        if (EPI == KD_EPI_QKV && p.qkv_packed) {                  // operand format of the split attention cores (kdiff_hip.h)
          u32x2 hi, lo;
          split4(v, hi, lo);
          v = f32x4{__uint_as_float(hi[0]), __uint_as_float(hi[1]), __uint_as_float(lo[0]), __uint_as_float(lo[1])};
        }
        if (gm < M) *reinterpret_cast<f32x4*>(p.C + (long)gm * N + gn) = v;
      }
    };
#pragma unroll
    for (int pass = 0; pass < NPASS; pass += 2) {
      one_pass(pass, csA, snA, csB, snB);
      one_pass(pass + 1, csB, snB, csA, snA);
    }
  };

  // ---- main loop over n-tiles; the K-steps of a tile are unrolled so that A fragments have static register names ----------
  for (int nt = 0; nt < n_tiles; ++nt) {
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
      const int s = nt * NK + ks;
      // Stage s has landed?  Stages ks = 0, 1 of every tile but the first were confirmed before the previous epilogue.
      if (nt > 0 && ks == 2 && full_panel && NWV == 4) {
        // First counted wait after an epilogue.  vmcnt retires in issue order on gfx9-class hardware, loads and stores
        // alike, and this wave's queue now reads [stage s][NST epilogue stores][stage s+1][stage s+2]: allowing the
        // stores to stay outstanding (+NST) waits for stage s only.  Counting loads alone here made every n-tile wait for
        // the acknowledgement of the previous tile's stores -- microseconds under HBM write pressure.  (Only when no row
        // of the panel is masked: a masked row's store may not be issued at all.)
        constexpr int NST = GEGLU ? 8 : 16;
        if (s + 2 < total) { if (NST == 8) KD_WAIT_VM(16); else KD_WAIT_VM(24); }
        else if (s + 1 < total) { if (NST == 8) KD_WAIT_VM(12); else KD_WAIT_VM(20); }
        else { if (NST == 8) KD_WAIT_VM(8); else KD_WAIT_VM(16); }
      } else if (nt == 0 || ks >= 2) {
        if (NWV == 4) { if (s + 2 < total) KD_WAIT_VM(8); else if (s + 1 < total) KD_WAIT_VM(4); else KD_WAIT_VM(0); }
        else { if (s + 2 < total) KD_WAIT_VM(4); else if (s + 1 < total) KD_WAIT_VM(2); else KD_WAIT_VM(0); }
      }
      KD_BARRIER();                      // every wave's quarter of stage s is in; everyone is done reading slot (s-1) % NSTG
      if (s + 3 < total) issue(s + 3);   // refill the slot freed by stage s-1
      const char* st = ring + (ks % NSTG) * STAGE;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = 2 * ks + h;
        const int o = swz(l31, 2 * h + lh);
        bf16x8 bh[4], bl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bh[j] = *reinterpret_cast<const bf16x8*>(st + j * 32 * 64 + o);
          bl[j] = *reinterpret_cast<const bf16x8*>(st + IMG + j * 32 * 64 + o);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[c], bh[j], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[c], bl[j], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[c], bh[j], acc[j], 0, 0, 0);
      }
    }
    // stages (nt+1, ks = 0, 1) were requested >= 2 stages ago: confirm them now, before the stores of this epilogue
    // enter the vector-memory queue (outstanding after the last issue: stages s+1, s+2, s+3 -> leave only s+3)
    if (nt + 1 < n_tiles) { if (NWV == 4) KD_WAIT_VM(4); else KD_WAIT_VM(2); }
    epilogue(nt);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  }
}

template <int NC, int EPI, int NWV>
static int launch(const GemmP& d, hipStream_t s) {
  auto kern = gemm_astat_kernel<NC, EPI, NWV>;
  constexpr int LDS_BYTES = lds_bytes(NWV), BMW = NWV * 32;
  static LdsAttr attr_set;
  attr_set.ensure(reinterpret_cast<const void*>(kern), LDS_BYTES);
  const double n_eff = (EPI == KD_EPI_GEGLU) ? 2.0 * d.N : (double)d.N;
  char nm[96] = "gemm_astat";
  if (prof_on()) snprintf(nm, sizeof(nm), "gemm_astat<e%d> M=%d N=%d K=%d", EPI, d.M, d.N, d.K);
  LaunchScope prof(nm, 2.0 * d.M * n_eff * d.K, 4.0 * ((double)d.M * d.K + n_eff * d.K + (double)d.M * d.N), s);
  // panels x n-splits: aim at >= 256 workgroups (one per CU) while every split keeps >= 2 n-tiles
  const int panels = (d.M + BMW - 1) / BMW, n_tiles = d.N / (EPI == KD_EPI_GEGLU ? 64 : 128);
  int splits = 1;
  while (panels * splits < 256 && n_tiles / (splits * 2) >= 2) splits *= 2;
  GemmP e = d;
  if (option("astat_storewait", 0)) e.debug |= 32;
  hipLaunchKernelGGL(kern, dim3((unsigned)panels, (unsigned)splits), dim3(64 * NWV), LDS_BYTES, s, e);
  return check_launch("kd_gemm_f32(astat)");
}

}  // namespace astat

// Eligibility + dispatch (called by kd_gemm_f32).  Returns 1 if the descriptor was not taken.
int gemm_astat_try(const GemmP& d, hipStream_t s, int* rc) {
  using namespace astat;
  if (d.precision != KD_PREC_SPLIT3 || d.a_mode != KD_A_PLAIN || !d.norm || !d.Wp || (d.debug & ~32)) return 1;
  if (d.epi != KD_EPI_STORE && d.epi != KD_EPI_QKV && d.epi != KD_EPI_GEGLU) return 1;
  const int max_k = option("astat_max_k", 512);    // A/B switch for benchmarks/
  if ((d.K != 128 && d.K != 256 && d.K != 512) || d.K > max_k) return 1;
  const int ncol = d.epi == KD_EPI_GEGLU ? 64 : 128;
  if (d.N % ncol || d.N / ncol < 2) return 1;                                   // one n-tile: nothing to amortise
  if (!(d.scale_stride == 0 || d.rows_per_sample % BM == 0)) return 1;          // one scale vector per panel
  if (d.epi == KD_EPI_QKV && (d.rows_per_sample % BM || d.n_heads > 16)) return 1;
  if (d.M < 4 * BM) return 1;
  // 256-row panels (8 waves, one W ring; KDIFF_ASTAT_WAVES=8) where the A fragments leave room for two waves per SIMD
  // (K = 128) and the panel grid still fills the chip.  Measured neutral against two co-resident 4-wave workgroups
  // (level-0 qkv 59 -> 65 us, GEGLU 117 -> 114 us, end to end -0.4 %: profiles/r01_astat_ablation.md), so the 4-wave
  // form stays the default: halving the W bytes into the CU is not what the main loop waits for.
  const int max_waves = option("astat_waves", 4);
  const bool wide = max_waves >= 8 && d.K == 128 && d.M >= 256 * 2 * BM && (d.scale_stride == 0 || d.rows_per_sample % (2 * BM) == 0) &&
                    (d.epi != KD_EPI_QKV || d.rows_per_sample % (2 * BM) == 0);
#define KD_AS(NCV, EP) if (d.K == NCV * 16 && d.epi == EP) { *rc = launch<NCV, EP, 4>(d, s); return 0; }
  if (wide) {
    if (d.epi == KD_EPI_STORE) { *rc = launch<8, KD_EPI_STORE, 8>(d, s); return 0; }
    if (d.epi == KD_EPI_QKV) { *rc = launch<8, KD_EPI_QKV, 8>(d, s); return 0; }
    if (d.epi == KD_EPI_GEGLU) { *rc = launch<8, KD_EPI_GEGLU, 8>(d, s); return 0; }
  }
  KD_AS(8, KD_EPI_STORE) KD_AS(8, KD_EPI_QKV) KD_AS(8, KD_EPI_GEGLU)
  KD_AS(16, KD_EPI_STORE) KD_AS(16, KD_EPI_QKV) KD_AS(16, KD_EPI_GEGLU)
  KD_AS(32, KD_EPI_STORE) KD_AS(32, KD_EPI_QKV) KD_AS(32, KD_EPI_GEGLU)
#undef KD_AS
  return 1;
}

}  // namespace kd

KD_TEXT_PAD(gemm_astat)      // last function of this code object: kd_common.h, code warm-up
