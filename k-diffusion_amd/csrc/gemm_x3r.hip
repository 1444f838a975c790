// fp32-parity ("split3") projections WITHOUT a norm in front -- the residual projections behind the attention cores and the FF blocks'
// down projections where the block is not fused, the 2 x 2 token merges -- in the round-3 shape:  C = [R +] gather(A) W^T, fp32 A in HBM.
//
// Replaces gemm.hip's round-1 tile kernel for these shapes (both operands through registers, one barrier per 32-k step between the
// ds_write pass and the MFMAs: 61 us for the level-2 down projection against a 19 us matrix floor).  What changed:
//   * 128 x 128 tile, swapped product (a lane owns ONE activation row: wave w the tile rows 32 w .. + 31, all 128 columns), so a wave only
//     ever reads ITS OWN rows of the A sub-tile: the wave stages them itself (global_load_lds, whole 128-byte runs of 8 rows per
//     instruction, 16-byte chunks XOR-swizzled on the source side) and needs no barrier for them;
//   * the fp32 row chunks are split into hi / lo bf16 fragments in registers right before their MFMAs (22 vector instructions per 12
//     MFMAs: inside the issue shadow, profiles/r03_issue_model.md) -- nothing is written back to LDS;
//   * the packed weight (kd_pack_weight_bf16x3) streams through the same 4-stage ring as in gemm_x3.hip (lane-linear 16 KiB stages);
//   * the residual is the accumulators' start value (read straight into the C layout before the ring starts), stores go through the
//     wave's LDS strip as 64-byte row runs;
//   * the merge gather (image_transformer_v2.py:586-595: a coarse token's row is its four fine tokens' rows side by side) is address
//     arithmetic of the staging requests: a 32-k stage lies inside ONE fine token's row.
#include "x3_common.h"

namespace kd {
namespace x3 { extern unsigned long long* g_clk; }      // gemm_x3.hip (kd_prof_clock_buffer)
namespace x3r {

using namespace x3;

constexpr int ASTG = 16384;             // A sub-tile of a stage: [128 rows][32 k] fp32, rows of 128 bytes
constexpr int STAGE = ASTG + STG;       // + the packed W stage [hi | lo][128 rows][32 k] bf16
constexpr int NSTG = 4, PDIST = NSTG - 1;

struct RArgs {
  const float* A; const char* Wp; float* C; const float* R;
  int M, N, K, nk;                      // nk = K / 32
  int gh, gw;                           // merge: the COARSE token grid (A is [B, 2 gh, 2 gw, K / 4])
  int warm;
  unsigned long long* clk;             // kd_prof_clock_buffer: stamps of one workgroup's stage 8
};

template <int AMODE, int EPI>
__global__ __launch_bounds__(256, 1) void gemm_x3r_kernel(const RArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const auto warm = code_warm_begin<12 * 1024>((int)blockIdx.x < p.warm && tid < 64);
  const int n_tiles = p.N / 128;
  int tile;
  {   // XCD-aware order: the n-tiles that re-read one row panel run back to back on ONE L2
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  tile = __builtin_amdgcn_readfirstlane(tile);
  const int nt = tile % n_tiles, mt = tile / n_tiles;
  const int m0 = mt * 128, n0 = nt * 128;
  const int nk = p.nk;

  f32x16 acc[4];
  // ---- staging requests of one stage: 4 pieces of this wave's OWN 32 rows of A (piece i: rows 32 w + 8 i .. + 7, lane = 8 * row + slot),
  // 4 pieces of the shared W stage ---------------------------------------------------------------------------------------------------------
  const char* arow[4];                   // &A[row of piece i][chunk this lane fetches], k = 0 (merge: the row's first fine token)
  const int Cin = p.K >> 2;              // merge: features of a fine token
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rt = wid * 32 + 8 * i + (lane >> 3);                 // tile row
    const int q = (lane & 7) ^ ((rt >> 1) & 7);                    // source chunk for LDS slot lane & 7
    const int gm = min(m0 + rt, p.M - 1);
    if constexpr (AMODE == KD_A_MERGE2x2) {
      const int per = p.gh * p.gw, b = gm / per, rr = gm - b * per, ci = rr / p.gw, cj = rr - ci * p.gw;
      arow[i] = reinterpret_cast<const char*>(p.A + (((size_t)b * (2 * p.gh) + 2 * ci) * (2 * p.gw) + 2 * cj) * Cin) + q * 16;
    } else {
      arow[i] = reinterpret_cast<const char*>(p.A + (size_t)gm * p.K) + q * 16;
    }
  }
  const char* wp = p.Wp + (size_t)nt * nk * STG + wid * 4096 + lane * 16;
  auto issue = [&](int s_) {
    const int s = min(s_, nk - 1);                                 // past the end: the last stage again (uniform request count, never read)
    char* slot = smem + (s_ % NSTG) * STAGE;
    size_t aoff;
    if constexpr (AMODE == KD_A_MERGE2x2) {
      const int k0 = s * 32, quad = k0 / Cin, e = k0 - quad * Cin;           // a stage lies inside one fine token (Cin % 32 == 0)
      aoff = ((size_t)(quad >> 1) * (2 * p.gw) + (quad & 1)) * Cin * 4 + (size_t)e * 4;
    } else {
      aoff = (size_t)s * 128;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(arow[i] + aoff),
                                       (__attribute__((address_space(3))) void*)(slot + wid * 4096 + i * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wp + (size_t)s * STG + i * 1024),
                                       (__attribute__((address_space(3))) void*)(slot + ASTG + wid * 4096 + i * 1024), 16, 0, 0);
  };
#pragma unroll
  for (int s = 0; s < PDIST; ++s) issue(s);
  // KD_EPI_RESIDUAL: the accumulators start from R (C layout: lane (l31, lh), block j, register 4 g + e <-> row l31, column 32 j + 8 g + 4 lh + e)
  {
    const int rrow = min(m0 + wid * 32 + l31, p.M - 1);
    if constexpr (EPI == KD_EPI_RESIDUAL) {
      const float* rp = p.R + (size_t)rrow * p.N + n0 + 4 * lh;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(rp + 32 * j + 8 * g);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[j][4 * g + e] = v[e];
        }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    }
  }

  // (consumed HERE as far as the compiler knows: its wait for the residual loads lands in front of the loop -- together with the first
  // stages, which are needed at once anyway -- and not as a conservative vmcnt inside it)
  asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
  code_warm_end(warm);

  // fragment addresses inside a stage: this lane's row of the A sub-tile (chunk pair 4 c + 2 lh, + 1 of chunk c), W rows 32 j + l31
  const int rt = wid * 32 + l31;
  const int a0 = rt * 128 + (((2 * lh) ^ ((rt >> 1) & 7)) << 4);   // c = 1: ^ 64; second half of the pair: ^ 16
  const int o0 = swz64(l31, lh), o1 = swz64(l31, 2 + lh);
  // The K loop is software-pipelined by 16-k chunks, as in gemm_x3.hip: while the 12 MFMAs of a chunk run, the 10 fragment reads of the
  // NEXT chunk are in flight (also across the stage boundary, whose wait + barrier sits in the MIDDLE of a stage) and its fp32 row pieces
  // are split into hi / lo; the 8 staging requests of stage s + 3 are issued one by one between the MFMAs of the stage's second chunk.
  f32x4 xr[2][2];                        // raw row pieces of a chunk, [buffer][half]
  bf16x8 wh[2][4], wl[2][4], ah[2], al[2];
  auto read_chunk = [&](int slot, int c, int buf) {
    const char* st = smem + slot * STAGE;
    xr[buf][0] = *reinterpret_cast<const f32x4*>(st + (a0 ^ (c << 6)));
    xr[buf][1] = *reinterpret_cast<const f32x4*>(st + (a0 ^ (c << 6) ^ 16));
    const char* wst = st + ASTG + (c ? o1 : o0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      wh[buf][j] = *reinterpret_cast<const bf16x8*>(wst + j * 32 * 64);
      wl[buf][j] = *reinterpret_cast<const bf16x8*>(wst + IMG + j * 32 * 64);
    }
  };
  auto split_chunk = [&](int buf) {
    u32x4 hi, lo;
    split8(xr[buf][0], xr[buf][1], hi, lo);
    ah[buf] = __builtin_bit_cast(bf16x8, hi);
    al[buf] = __builtin_bit_cast(bf16x8, lo);
  };
  auto mm = [&](int buf, int j, int term) {
    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(term == 0 ? wl[buf][j] : wh[buf][j], term == 1 ? al[buf] : ah[buf], acc[j], 0, 0, 0);
  };
  auto issue_one = [&](int s_, int i) {                            // piece i (0..3: A, 4..7: W) of stage s_'s requests
    const int s = min(s_, nk - 1);
    char* slot = smem + (s_ % NSTG) * STAGE;
    if (i < 4) {
      size_t aoff;
      if constexpr (AMODE == KD_A_MERGE2x2) {
        const int k0 = s * 32, quad = k0 / Cin, e = k0 - quad * Cin;
        aoff = ((size_t)(quad >> 1) * (2 * p.gw) + (quad & 1)) * Cin * 4 + (size_t)e * 4;
      } else {
        aoff = (size_t)s * 128;
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(arow[i] + aoff),
                                       (__attribute__((address_space(3))) void*)(slot + wid * 4096 + i * 1024), 16, 0, 0);
    } else {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wp + (size_t)s * STG + (i - 4) * 1024),
                                       (__attribute__((address_space(3))) void*)(slot + ASTG + wid * 4096 + (i - 4) * 1024), 16, 0, 0);
    }
  };
  wait_vm(8 * (PDIST - 1));
  KD_BARRIER();
  read_chunk(0, 0, 0);
  split_chunk(0);
  const bool probe = p.clk && blockIdx.x == (gridDim.x * 5) / 8 && tid == 0;
  if (probe) { p.clk[0] = __builtin_amdgcn_s_memtime(); p.clk[1] = __builtin_amdgcn_s_memrealtime(); }
  for (int s = 0; s < nk; ++s) {
    const int slot = s % NSTG, nslot = (s + 1) % NSTG;
    if (probe && s == 8) p.clk[8] = __builtin_amdgcn_s_memtime();
    // ---- chunk 0 of stage s (buffer 0); chunk 1's fragments are requested behind its first MFMA -----------------------------------------
    mm(0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    read_chunk(slot, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    mm(0, 1, 0); mm(0, 2, 0); mm(0, 3, 0);
    mm(0, 0, 1); mm(0, 1, 1); mm(0, 2, 1); mm(0, 3, 1);
    __builtin_amdgcn_sched_barrier(0);
    split_chunk(1);
    __builtin_amdgcn_sched_barrier(0);
    mm(0, 0, 2); mm(0, 1, 2); mm(0, 2, 2); mm(0, 3, 2);
    __builtin_amdgcn_sched_barrier(0);
    // ---- stage s + 1 in for every wave, everyone past stage s - 1 -------------------------------------------------------------------------
    if (probe && s == 8) p.clk[9] = __builtin_amdgcn_s_memtime();
    wait_vm(8 * (PDIST - 2));
    if (probe && s == 8) p.clk[10] = __builtin_amdgcn_s_memtime();
    KD_BARRIER();
    if (probe && s == 8) p.clk[11] = __builtin_amdgcn_s_memtime();
    // ---- chunk 1 of stage s (buffer 1); the next stage's first chunk behind its first MFMA; the requests of stage s + 3 in between -------
    mm(1, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    read_chunk(nslot, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    mm(1, 1, 0);
    issue_one(s + PDIST, 0);
    __builtin_amdgcn_sched_barrier(0);
    mm(1, 2, 0);
    issue_one(s + PDIST, 1);
    __builtin_amdgcn_sched_barrier(0);
    mm(1, 3, 0); mm(1, 0, 1);
    issue_one(s + PDIST, 2);
    __builtin_amdgcn_sched_barrier(0);
    mm(1, 1, 1);
    issue_one(s + PDIST, 3);
    __builtin_amdgcn_sched_barrier(0);
    mm(1, 2, 1); mm(1, 3, 1);
    issue_one(s + PDIST, 4);
    __builtin_amdgcn_sched_barrier(0);
    split_chunk(0);
    __builtin_amdgcn_sched_barrier(0);
    mm(1, 0, 2);
    issue_one(s + PDIST, 5);
    __builtin_amdgcn_sched_barrier(0);
    mm(1, 1, 2);
    issue_one(s + PDIST, 6);
    __builtin_amdgcn_sched_barrier(0);
    mm(1, 2, 2);
    issue_one(s + PDIST, 7);
    mm(1, 3, 2);
    __builtin_amdgcn_sched_barrier(0);
    if (probe && s == 8) p.clk[12] = __builtin_amdgcn_s_memtime();
  }
  if (probe) { p.clk[2] = __builtin_amdgcn_s_memtime(); p.clk[3] = __builtin_amdgcn_s_memrealtime(); p.clk[7] = (unsigned long long)nk; }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the clamped tail requests still target this workgroup's LDS

  // ---- store through the wave's strip: 16 rows x 64 bytes per instruction ----------------------------------------------------------------
  char* strip = smem + NSTG * STAGE + wid * 2048;
  float* st_row[2];
  bool st_ok[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int r = m0 + wid * 32 + 16 * it + (lane >> 2);
    st_ok[it] = r < p.M;
    st_row[it] = p.C + (size_t)min(r, p.M - 1) * p.N + n0 + 4 * (lane & 3);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
      for (int gg = 0; gg < 2; ++gg) {
        const int g = 2 * hb + gg;
        *reinterpret_cast<f32x4*>(strip + l31 * 64 + (((2 * gg + lh) ^ ((l31 >> 2) & 1)) << 4)) =
            f32x4{acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
      }
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int r16 = 16 * it + (lane >> 2), cc = lane & 3;
        const f32x4 o = *reinterpret_cast<const f32x4*>(strip + r16 * 64 + ((cc ^ ((r16 >> 2) & 1)) << 4));
        if (st_ok[it]) *reinterpret_cast<f32x4*>(st_row[it] + 32 * j + 16 * hb) = o;
      }
    }
}

template <int AMODE, int EPI>
static int launch(const RArgs& a, const char* nm, double flops, double bytes, hipStream_t s) {
  auto kern = gemm_x3r_kernel<AMODE, EPI>;
  constexpr int LDS = NSTG * STAGE + 4 * 2048;
  static LdsAttr attr_set;
  attr_set.ensure(reinterpret_cast<const void*>(kern), LDS);
  const long tiles = (long)((a.M + 127) / 128) * (a.N / 128);
  LaunchScope prof(nm, flops, bytes, s);
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), LDS, s, a);
  return check_launch("kd_gemm_f32(x3 residual / merge)");
}

}  // namespace x3r

// Eligibility + dispatch (called by kd_gemm_f32 ahead of the round-1 tile kernel).  Returns 1 if the descriptor was not taken.
int gemm_x3r_try(const GemmP& d, hipStream_t s, int* rc) {
  using namespace x3r;
  const int mode = option("x3r", 1);
  if (!mode) return 1;
  if (d.precision != KD_PREC_SPLIT3 || d.norm || !d.Wp || d.debug || d.a_split || d.c_split) return 1;
  if (d.a_mode != KD_A_PLAIN && d.a_mode != KD_A_MERGE2x2) return 1;
  if (d.epi != KD_EPI_STORE && d.epi != KD_EPI_RESIDUAL) return 1;
  if ((d.K & 31) || (d.N & 127) || d.M < 512 || d.out_add != 0.f) return 1;
  // One workgroup per CU (136 KiB of LDS): taken where the tiles fit ONE round of the chip -- the level-2 projections and the merge
  // into level 2 (M = 8192 at batch 32: 47-52 vs 59-62 us, 22-24 vs 28, 33-37 vs 50-52 us).  With more tiles than CUs the round-1 kernel's
  // two resident workgroups per CU win (level-1 out projection 31 vs 27 us, level-0 merge 45 vs 38 us: benchmarks/x3_bench.py).
  // Option "x3r" = 2 takes every eligible shape (A/B runs).  (A TokenSplit + lerp epilogue on this kernel -- scatter as row bases of the
  // store runs, skip requested ahead of the K loop -- was measured slower than the round-1 kernel's, 57.9 vs 53.8 and 44.6 vs 41.4 us, and
  // was not kept.)
  {
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    static int cached = 0;
    if (!cached) { if (hipDeviceGetAttribute(&cached, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cached <= 0) cached = 256; }
    cus = cached;
    const long tiles = (long)((d.M + 127) / 128) * (d.N / 128);
    if (mode != 2 && tiles > cus) return 1;
  }
  if (d.a_mode == KD_A_MERGE2x2 && ((d.K >> 2) & 31)) return 1;
  RArgs a{};
  a.A = d.A; a.Wp = reinterpret_cast<const char*>(d.Wp); a.C = d.C; a.R = d.R;
  a.M = d.M; a.N = d.N; a.K = d.K; a.nk = d.K / 32; a.gh = d.gh; a.gw = d.gw;
  a.warm = d.warm;
  a.clk = x3::g_clk;
  const double flops = 2.0 * d.M * (double)d.N * d.K;
  const double bytes = 4.0 * ((double)d.M * d.K + (double)d.N * d.K + (double)d.M * d.N * (d.epi == KD_EPI_RESIDUAL ? 2 : 1));
  char nm[96] = "gemm_x3r";
  if (prof_on()) snprintf(nm, sizeof(nm), "gemm_x3r<a%d,e%d> M=%d N=%d K=%d", d.a_mode, d.epi, d.M, d.N, d.K);
  if (d.a_mode == KD_A_PLAIN && d.epi == KD_EPI_RESIDUAL) *rc = launch<KD_A_PLAIN, KD_EPI_RESIDUAL>(a, nm, flops, bytes, s);
  else if (d.a_mode == KD_A_PLAIN) *rc = launch<KD_A_PLAIN, KD_EPI_STORE>(a, nm, flops, bytes, s);
  else if (d.epi == KD_EPI_STORE) *rc = launch<KD_A_MERGE2x2, KD_EPI_STORE>(a, nm, flops, bytes, s);
  else return 1;
  return 0;
}

}  // namespace kd

KD_TEXT_PAD(gemm_x3r)      // last function of this code object: kd_common.h, code warm-up
