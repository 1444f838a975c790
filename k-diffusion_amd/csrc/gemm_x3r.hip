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
//
// Round 4: LOADER WAVES (template flag LW, option "x3r_lw", default on).  With the staging requests inside the K loop a wave spent more
// time ISSUING its 8 LDS-DMA instructions per stage than on the stage's 24 MFMAs (1 640 clocks per stage against 768 of MFMA: a request
// costs the issuing wave 60 - 180 clocks while the address path takes it, and nothing else of that wave issues meanwhile -- its MFMAs
// included).  The kernel uses 200 of a SIMD's 512 registers per lane, so the workgroup can carry a SECOND wave per SIMD that does nothing
// but staging: waves 4..7 issue the requests of row group (wave - 4) and their quarter of the W stage, count them in (vmcnt) and meet the
// compute waves at the ring's one barrier per stage; waves 0..3 keep only ds_read / split / MFMA in their loop.  The matrix pipe no longer
// waits for the address path (MI355X_MICROARCH.md, "Two waves per SIMD": the pairing that nets is matrix beside memory).
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
  int gh, gw;                           // merge / split: the COARSE token grid (merge: A is [B, 2 gh, 2 gw, K / 4]; split: C and R are [B, 2 gh, 2 gw, N / 4])
  const float* fac;                     // KD_EPI_SPLIT_LERP: the lerp weight (device scalar)
  int warm;
  unsigned long long* clk;             // kd_prof_clock_buffer: stamps of one workgroup's stage 8
};

template <int AMODE, int EPI, bool LW, bool PROBE>
__global__ __launch_bounds__(LW ? 512 : 256, LW ? 2 : 1) void gemm_x3r_kernel(const RArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wid_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = LW && wid_all >= 4;                          // (wave-uniform)
  const int wid = LW ? (wid_all & 3) : wid_all;                    // row group: rows 32 wid .. + 31 of the tile (compute AND staging role)
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
  const bool probe = PROBE && p.clk && blockIdx.x == (gridDim.x * 5) / 8 && tid == 0;
  if (probe) p.clk[13] = __builtin_amdgcn_s_memtime();             // kernel entry
  // extended time line (benchmarks/x3r_bench.py sets clk[15] to the magic value and hands over 32 + 3 * grid entries): entry / exit of EVERY
  // workgroup's first compute wave and exit of its first loader wave, in 100 MHz ticks
  const bool wide = PROBE && p.clk && p.clk[15] == 0x4b44ull && lane == 0;
  if (wide && wid_all == 0) p.clk[32 + 3 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();

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
  if constexpr (LW) {
    if (loader) {
      // ---- loader wave: the staging requests of row group `wid` and its quarter of every W stage, nothing else ------------------------------
#pragma unroll
      for (int s = 0; s < PDIST; ++s) issue(s);
      wait_vm(8 * (PDIST - 1));
      KD_BARRIER();                                                // stage 0 in (the compute waves' first barrier)
      for (int s = 0; s < nk; ++s) {
        // stage s + 1 in before the barrier in the middle of stage s; behind that barrier every compute wave is past stage s - 1,
        // whose slot takes stage s + PDIST
        wait_vm(8 * (PDIST - 2));
        KD_BARRIER();
        issue(s + PDIST);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the clamped tail requests still target this workgroup's LDS
      if (wide && wid_all == 4) p.clk[32 + 3 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
      return;
    }
  } else {
#pragma unroll
    for (int s = 0; s < PDIST; ++s) issue(s);
  }
  // KD_EPI_SPLIT_LERP (TokenSplit, image_transformer_v2.py:610-621): output column n = quadrant * cout + e of coarse token gm is feature e of the
  // fine token (2 h + (quadrant >> 1), 2 w + (quadrant & 1)); an n-tile of 128 columns lies inside ONE quadrant (cout % 128 == 0), so a tile row
  // is 128 contiguous floats of one fine token's row, for the result and for the skip operand alike.  Offset (floats) of tile row `gm`:
  auto out_row = [&](int gm) -> size_t {
    if constexpr (EPI == KD_EPI_SPLIT_LERP) {
      const int cout = p.N >> 2, qd = n0 / cout, e0 = n0 - qd * cout;
      const int hw = p.gh * p.gw, b = gm / hw, rr = gm - b * hw, h = rr / p.gw, w = rr - h * p.gw;
      return (((size_t)b * (2 * p.gh) + 2 * h + (qd >> 1)) * (2 * p.gw) + 2 * w + (qd & 1)) * cout + e0;
    } else {
      return (size_t)gm * p.N + n0;
    }
  };
  // KD_EPI_RESIDUAL: the accumulators start from R (C layout: lane (l31, lh), block j, register 4 g + e <-> row l31, column 32 j + 8 g + 4 lh + e).
  // KD_EPI_SPLIT_LERP: the skip operand is read the same way, into registers of its own (lerp is not a sum), ahead of the K loop.
  f32x16 skp[EPI == KD_EPI_SPLIT_LERP ? 4 : 1];
  {
    const int rrow = min(m0 + wid * 32 + l31, p.M - 1);
    if constexpr (EPI == KD_EPI_RESIDUAL || EPI == KD_EPI_SPLIT_LERP) {
      const float* rp = p.R + out_row(rrow) + 4 * lh;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(rp + 32 * j + 8 * g);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if constexpr (EPI == KD_EPI_RESIDUAL) acc[j][4 * g + e] = v[e];
            else skp[j][4 * g + e] = v[e];
          }
        }
    }
    if constexpr (EPI != KD_EPI_RESIDUAL) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    }
  }

  // (consumed HERE as far as the compiler knows: its wait for the residual loads lands in front of the loop -- together with the first
  // stages, which are needed at once anyway -- and not as a conservative vmcnt inside it)
  asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
  if constexpr (EPI == KD_EPI_SPLIT_LERP) asm volatile("" : "+v"(skp[0]), "+v"(skp[1]), "+v"(skp[2]), "+v"(skp[3]));
  code_warm_end(warm);

  // fragment addresses inside a stage: this lane's row of the A sub-tile (chunk pair 4 c + 2 lh, + 1 of chunk c), W rows 32 j + l31
  const int rt = wid * 32 + l31;
  const int a0 = rt * 128 + (((2 * lh) ^ ((rt >> 1) & 7)) << 4);   // c = 1: ^ 64; second half of the pair: ^ 16
  const int o0 = swz64(l31, lh), o1 = swz64(l31, 2 + lh);
  // The K loop is software-pipelined by 16-k chunks: while the 12 MFMAs of a chunk run, the 10 fragment reads of the NEXT chunk (also
  // across the stage boundary, whose wait + barrier sits in the MIDDLE of a stage) and the hi / lo split of its fp32 row pieces are
  // spread over the gaps between them, one gap = one MFMA + at most 7 other instructions (an MFMA's issue shadow holds ~6:
  // profiles/r03_issue_model.md).  Round 4: the loop body is ONE basic block (the time stamps are a template flag) and every split piece
  // is pinned where it is written (empty asm) -- with the run-time `if (probe)` blocks inside the loop the compiler had sunk both 32-instruction
  // split sequences behind the last MFMA of their chunk, ~130 exposed clocks each (benchmarks/x3r_bench.py time line: 1 500 clocks per stage).
  f32x4 xr[2][2];                        // raw row pieces of a chunk, [buffer][half]
  bf16x8 wh[2][4], wl[2][4];
  u32x4 ahu[2], alu[2];
  auto read_a = [&](int slot, int c, int buf) {
    const char* st = smem + slot * STAGE;
    xr[buf][0] = *reinterpret_cast<const f32x4*>(st + (a0 ^ (c << 6)));
    xr[buf][1] = *reinterpret_cast<const f32x4*>(st + (a0 ^ (c << 6) ^ 16));
  };
  auto read_w = [&](int slot, int c, int buf, int j) {
    const char* wst = smem + slot * STAGE + ASTG + (c ? o1 : o0) + j * 32 * 64;
    wh[buf][j] = *reinterpret_cast<const bf16x8*>(wst);
    wl[buf][j] = *reinterpret_cast<const bf16x8*>(wst + IMG);
  };
  // piece q of the split of buffer `buf`: row elements 2 q, 2 q + 1 of the lane's 8 -> one register of the hi and of the lo fragment
  auto split_piece = [&](int buf, int q) {
    const float a = xr[buf][q >> 1][2 * (q & 1)], b = xr[buf][q >> 1][2 * (q & 1) + 1];
    unsigned h = pack_bf16(a, b);
    asm volatile("" : "+v"(h));          // (one v_cvt_pk: the shifts below read ITS result instead of converting a second time)
    unsigned l = pack_bf16(a - b16::bf_lo(h), b - b16::bf_hi(h));
    asm volatile("" : "+v"(l));          // materialised HERE: not sunk towards its first use in the next chunk
    ahu[buf][q] = h;
    alu[buf][q] = l;
  };
  auto mm = [&](int buf, int j, int term) {
    const bf16x8 a = __builtin_bit_cast(bf16x8, term == 1 ? alu[buf] : ahu[buf]);
    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(term == 0 ? wl[buf][j] : wh[buf][j], a, acc[j], 0, 0, 0);
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
#define KD_GAP() __builtin_amdgcn_sched_barrier(0)
  // the 12 MFMAs of the chunk in buffers `cur`; the reads and the split of chunk (rslot, rc) into buffers `nxt`; in the form without loader
  // waves the second chunk of a stage also carries the 8 staging requests of stage s + PDIST (`req` >= 0), one per gap
  auto chunk = [&](int cur, int nxt, int rslot, int rc, int req) {
    mm(cur, 0, 0); read_a(rslot, rc, nxt); KD_GAP();
    mm(cur, 1, 0); read_w(rslot, rc, nxt, 0); KD_GAP();
    mm(cur, 2, 0); read_w(rslot, rc, nxt, 1); if (!LW && req >= 0) issue_one(req, 0); KD_GAP();
    mm(cur, 3, 0); read_w(rslot, rc, nxt, 2); if (!LW && req >= 0) issue_one(req, 1); KD_GAP();
    mm(cur, 0, 1); read_w(rslot, rc, nxt, 3); if (!LW && req >= 0) issue_one(req, 2); KD_GAP();
    mm(cur, 1, 1); if (!LW && req >= 0) issue_one(req, 3); KD_GAP();
    mm(cur, 2, 1); split_piece(nxt, 0); if (!LW && req >= 0) issue_one(req, 4); KD_GAP();
    mm(cur, 3, 1); split_piece(nxt, 1); if (!LW && req >= 0) issue_one(req, 5); KD_GAP();
    mm(cur, 0, 2); split_piece(nxt, 2); if (!LW && req >= 0) issue_one(req, 6); KD_GAP();
    mm(cur, 1, 2); split_piece(nxt, 3); if (!LW && req >= 0) issue_one(req, 7); KD_GAP();
    mm(cur, 2, 2); KD_GAP();
    mm(cur, 3, 2); KD_GAP();
  };
  if constexpr (!LW) wait_vm(8 * (PDIST - 1));
  KD_BARRIER();
  read_a(0, 0, 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) read_w(0, 0, 0, j);
#pragma unroll
  for (int q = 0; q < 4; ++q) split_piece(0, q);
  if (probe) { p.clk[0] = __builtin_amdgcn_s_memtime(); p.clk[1] = __builtin_amdgcn_s_memrealtime(); }
  for (int s = 0; s < nk; ++s) {
    const int slot = s % NSTG, nslot = (s + 1) % NSTG;
    if (PROBE && probe && s == 8) p.clk[8] = __builtin_amdgcn_s_memtime();
    // ---- chunk 0 of stage s (buffers 0); chunk 1's fragments and split in its gaps ---------------------------------------------------------
    chunk(0, 1, slot, 1, -1);
    // ---- stage s + 1 in for every wave, everyone past stage s - 1 -------------------------------------------------------------------------
    if (PROBE && probe && s == 8) p.clk[9] = __builtin_amdgcn_s_memtime();
    if constexpr (!LW) wait_vm(8 * (PDIST - 2));
    if (PROBE && probe && s == 8) p.clk[10] = __builtin_amdgcn_s_memtime();
    KD_BARRIER();
    if (PROBE && probe && s == 8) p.clk[11] = __builtin_amdgcn_s_memtime();
    // ---- chunk 1 of stage s (buffers 1); the next stage's first chunk in its gaps (and, without loader waves, the requests of stage s + 3) ---
    chunk(1, 0, nslot, 0, s + PDIST);
    if (PROBE && probe && s == 8) p.clk[12] = __builtin_amdgcn_s_memtime();
  }
#undef KD_GAP
  if (probe) { p.clk[2] = __builtin_amdgcn_s_memtime(); p.clk[3] = __builtin_amdgcn_s_memrealtime(); p.clk[7] = (unsigned long long)nk; }
  if constexpr (!LW) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped tail requests still target this workgroup's LDS

  // ---- store through the wave's strip: 16 rows x 64 bytes per instruction ----------------------------------------------------------------
  char* strip = smem + NSTG * STAGE + wid * 2048;
  float* st_row[2];
  bool st_ok[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int r = m0 + wid * 32 + 16 * it + (lane >> 2);
    st_ok[it] = r < p.M;
    st_row[it] = p.C + out_row(min(r, p.M - 1)) + 4 * (lane & 3);
  }
  if constexpr (EPI == KD_EPI_SPLIT_LERP) {                          // torch.lerp(skip, x, fac), ATen's two-branch form
    const float fac = *p.fac;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float skip = skp[j][r], diff = acc[j][r] - skip;
        acc[j][r] = (fabsf(fac) < 0.5f) ? skip + fac * diff : acc[j][r] - diff * (1.0f - fac);
      }
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
        if (st_ok[it]) st16(st_row[it] + 32 * j + 16 * hb, o);
      }
    }
  if (probe) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); p.clk[14] = __builtin_amdgcn_s_memtime(); }      // stores out
  if (wide && wid_all == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); p.clk[32 + 3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime(); }
}

template <int AMODE, int EPI, bool LW, bool PROBE>
static int launch_lw(const RArgs& a, const char* nm, double flops, double bytes, hipStream_t s) {
  auto kern = gemm_x3r_kernel<AMODE, EPI, LW, PROBE>;
  constexpr int LDS = NSTG * STAGE + 4 * 2048;
  static LdsAttr attr_set;
  attr_set.ensure(reinterpret_cast<const void*>(kern), LDS);
  const long tiles = (long)((a.M + 127) / 128) * (a.N / 128);
  LaunchScope prof(nm, flops, bytes, s);
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(LW ? 512 : 256), LDS, s, a);
  return check_launch("kd_gemm_f32(x3 residual / merge)");
}
template <int AMODE, int EPI>
static int launch(const RArgs& a, const char* nm, double flops, double bytes, hipStream_t s) {
  // (the in-kernel time stamps of kd_prof_clock_buffer are their own instantiation: run-time `if (probe)` blocks inside the K loop split
  // it into several basic blocks, which the instruction scheduler does not cross)
  if (a.clk) return option("x3r_lw", 1) ? launch_lw<AMODE, EPI, true, true>(a, nm, flops, bytes, s) : launch_lw<AMODE, EPI, false, true>(a, nm, flops, bytes, s);
  return option("x3r_lw", 1) ? launch_lw<AMODE, EPI, true, false>(a, nm, flops, bytes, s) : launch_lw<AMODE, EPI, false, false>(a, nm, flops, bytes, s);
}

}  // namespace x3r

// Eligibility + dispatch (called by kd_gemm_f32 ahead of the round-1 tile kernel).  Returns 1 if the descriptor was not taken.
int gemm_x3r_try(const GemmP& d, hipStream_t s, int* rc) {
  using namespace x3r;
  const int mode = option("x3r", 1);
  if (!mode) return 1;
  if (d.precision != KD_PREC_SPLIT3 || d.norm || !d.Wp || d.debug || d.a_split || d.c_split) return 1;
  if (d.a_mode != KD_A_PLAIN && d.a_mode != KD_A_MERGE2x2) return 1;
  if (d.epi != KD_EPI_STORE && d.epi != KD_EPI_RESIDUAL && d.epi != KD_EPI_SPLIT_LERP) return 1;
  if (d.epi == KD_EPI_SPLIT_LERP && (d.a_mode != KD_A_PLAIN || !d.R || !d.fac || ((d.N >> 2) & 127) || d.gh <= 0 || d.gw <= 0 || d.M % (d.gh * d.gw) || !option("x3r_split", 1)))
    return 1;                        // (an n-tile inside one quadrant: cout % 128 == 0)
  if ((d.K & 31) || (d.N & 127) || d.M < option("x3r_min_rows", 128) || d.out_add != 0.f) return 1;
  // One workgroup per CU (136 KiB of LDS).  Round 3 took only grids of at most one tile per CU (the level-2 projections and the merge into
  // level 2); with the loader waves and the one-basic-block K loop of round 4 the kernel is level with or ahead of the round-1 tile kernel's
  // two workgroups per CU on every shape with K >= 256 (benchmarks/x3r_bench.py, batch 32: merge into level 1 31 vs 35 us, level-1 out
  // projection 26 vs 26, level-1 down projection 45 vs 60, level 2 19 / 39 / 27 vs 27 / 58 / 47 us), so it takes them all; at K = 128 (the
  // level-0 out projection, 4 stages: the prologue and the store tail are most of a tile's time) the round-1 kernel stays ahead, 39 vs 41 us.
  // Option "x3r" = 2 takes every eligible shape, 0 none (A/B runs).  The TokenSplit + lerp epilogue (scatter as row bases of the store runs,
  // skip requested ahead of the K loop) lost to the round-1 kernel's on round 3's loop, 57.9 vs 53.8 and 44.6 vs 41.4 us; on this loop it is
  // back (option "x3r_split", benchmarks/x3r_bench.py).
  if (mode != 2 && d.K < 256) return 1;
  if (d.a_mode == KD_A_MERGE2x2 && ((d.K >> 2) & 31)) return 1;
  RArgs a{};
  a.A = d.A; a.Wp = reinterpret_cast<const char*>(d.Wp); a.C = d.C; a.R = d.R;
  a.M = d.M; a.N = d.N; a.K = d.K; a.nk = d.K / 32; a.gh = d.gh; a.gw = d.gw; a.fac = d.fac;
  a.warm = d.warm;
  a.clk = x3::g_clk;
  const double flops = 2.0 * d.M * (double)d.N * d.K;
  const double bytes = 4.0 * ((double)d.M * d.K + (double)d.N * d.K + (double)d.M * d.N * (d.epi == KD_EPI_STORE ? 1 : 2));
  char nm[96] = "gemm_x3r";
  if (prof_on()) snprintf(nm, sizeof(nm), "gemm_x3r<a%d,e%d> M=%d N=%d K=%d", d.a_mode, d.epi, d.M, d.N, d.K);
  if (d.a_mode == KD_A_PLAIN && d.epi == KD_EPI_RESIDUAL) *rc = launch<KD_A_PLAIN, KD_EPI_RESIDUAL>(a, nm, flops, bytes, s);
  else if (d.epi == KD_EPI_SPLIT_LERP) *rc = launch<KD_A_PLAIN, KD_EPI_SPLIT_LERP>(a, nm, flops, bytes, s);
  else if (d.a_mode == KD_A_PLAIN) *rc = launch<KD_A_PLAIN, KD_EPI_STORE>(a, nm, flops, bytes, s);
  else if (d.epi == KD_EPI_STORE) *rc = launch<KD_A_MERGE2x2, KD_EPI_STORE>(a, nm, flops, bytes, s);
  else return 1;
  return 0;
}

}  // namespace kd

KD_TEXT_PAD(gemm_x3r)      // last function of this code object: kd_common.h, code warm-up
