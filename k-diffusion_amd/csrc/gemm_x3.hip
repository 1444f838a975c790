// fp32-parity ("split3") projections in the round-2 kernel shape: AdaRMSNorm -> wide projection (qkv + cosine-sim + RoPE, up-projection +
// GEGLU, plain store) with fp32 activations in HBM and every product as three bf16 MFMA terms (hi*hi + hi*lo + lo*hi, fp32 accumulate).
//
// Replaces gemm_astat.hip (round 1) for these shapes.  What changed, and why (profiles/r03_issue_model.md):
//   * swapped product D = W_frag x act_frag: a lane owns ONE activation row, so the whole epilogue (row factor, cosine-sim norm, RoPE
//     pairs, GEGLU, hi / lo split of the qkv operands) is in-lane arithmetic and every store is the lane's own 16 bytes (4 consecutive
//     features of its row) -- no transposing LDS strips, no cross-lane traffic but one half-wave sum;
//   * the wave's 32 rows come in through LDS as WHOLE rows (global_load_lds, fully coalesced, chunk-swizzled on the source side) and
//     are normalised, scaled and split ONCE into register fragments (a_hi / a_lo);
//   * the packed weight (kd_pack_weight_bf16x3: [n-tile][32-k stage][hi | lo][128 rows][32 k], 16 KiB per stage) streams through a
//     4- or 8-slot LDS ring, requested NSTG - 1 stages ahead with counted vmcnt (the epilogue's stores stay outstanding);
//   * the K loop is software-pipelined across stages: the fragment reads of the NEXT 16-k chunk -- also across the stage boundary, whose
//     wait + barrier sits in the MIDDLE of a stage -- are issued before the 12 MFMAs of the current chunk, so no MFMA waits for an LDS
//     round trip even with one wave per SIMD (K >= 256: the fragments of a row take 128 / 256 registers).  A stage is 16 ds_read_b128
//     per 24 MFMAs: inside the 6-issue-slot shadow of an MFMA measured in r03_issue_model.md;
//   * RoPE angles from the token's axial position and the head's frequencies (hardware sin / cos in revolutions) instead of cos / sin
//     tables: no vector loads inside the ring loop (hipcc waits vmcnt(0) for an ordinary load issued beside LDS-DMA, draining the ring).
#include "x3_common.h"

namespace kd {
namespace x3 {

struct XArgs {
  const float* A; const char* Wp; float* C;
  b16::u16* Cl; int c_split;        // GEGLU: result as bf16 hi (C) / lo (Cl) planes for an a_split down projection (gemm_x3t.hip)
  const float* scale; int scale_stride, rows_per_sample; float eps;
  int M, N, n_tiles, n_splits;
  int n_heads; const float* qk_scale; const float* pos; const float* freq; int qkv_packed;
  float out_add;
  const float* sigma; float sigma_data; int gh, gw, chan;    // KD_EPI_UNPATCH_NCHW: Karras c_out / c_skip per sample, patch grid, image channels (4 x 4 patches)
  const float* R;                   // KD_EPI_RESIDUAL: [M, N] added to the product (the accumulators start from it); one n-tile per workgroup
  int warm;
  unsigned long long* clk;
};

template <int NC /* K / 16 */, int EPI>
__global__ __launch_bounds__(256, NC <= 8 ? 2 : 1) void gemm_x3_astat_kernel(const XArgs p) {
  constexpr int K = NC * 16, NK = NC / 2;                       // ring stages per n-tile
  constexpr int NSTG = NC <= 8 ? 4 : 8, PDIST = NSTG - 1, PB = 4;   // PB: 1 KiB pieces of a stage per wave
  constexpr bool GEGLU = EPI == KD_EPI_GEGLU;
  constexpr int NCOL = GEGLU ? 64 : 128;
  constexpr int NST = GEGLU ? 8 : 16;                           // 16-byte stores per lane per n-tile
  constexpr int WAREA = (NSTG / 4) * STG;                       // row staging bytes per wave (the ring slots it borrows)
  // staging rounds: ALL 32 rows of the wave, KH of their K floats per round (K = 512: two rounds of 256).  Every lane then converts
  // its own row in every round; staging whole rows of half the lanes instead (round 3's first form) ran the conversion code twice with
  // half the lanes masked: 23 000 clocks of prologue at K = 512 (benchmarks/x3_bench.py time line) against 9 000 at K = 256
  constexpr int KH = WAREA / 128 >= K ? K : WAREA / 128;        // floats of a row per round: 128, 256, 256 at K = 128, 256, 512
  constexpr bool AG = NC >= 32;                                 // activation fragments in named AccVGPRs (see areg_write4)
  constexpr int NR = K / KH, CPR = KH / 4;                      // rounds; 16-byte chunks of a row per round
  constexpr int PIECES = 32 * CPR / 64;                         // 1 KiB pieces per round
  constexpr int SCL = K * 4 < 1024 ? 1024 : K * 4;              // bytes of a wave's scale vector area
  static_assert(NK % NSTG == 0 || NSTG % NK == 0, "ring slot of a stage must be a compile-time constant");
  static_assert(CPR >= 16 && NC % NR == 0, "source-side chunk swizzle spans 16 chunk positions");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform, and provably so: LDS-DMA bases (M0) and W addresses stay scalar
  const auto warm = code_warm_begin<(NC <= 8 ? 14 : 24) * 1024>((int)blockIdx.x < p.warm && tid < 64);
  // workgroup -> (row panel, n-split): the splits of one panel get ids 8 apart, i.e. the same XCD (one L2 fetches the panel's rows once)
  int panel, split;
  const int n_splits = p.n_splits, n_panels = gridDim.x / n_splits;
  if ((n_panels & 7) == 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    panel = (j / n_splits) * 8 + xcd;
    split = j % n_splits;
  } else {
    panel = blockIdx.x % n_panels;
    split = blockIdx.x / n_panels;
  }
  // (integer division by a run-time value goes through the vector unit: hand the uniform results back to scalar registers, or every
  // address and per-head constant derived from them is vector arithmetic, and the s_load operands below become waterfall loops)
  panel = __builtin_amdgcn_readfirstlane(panel);
  split = __builtin_amdgcn_readfirstlane(split);
  const int nt_begin = __builtin_amdgcn_readfirstlane((int)((long)p.n_tiles * split / n_splits));
  const int nt_end = __builtin_amdgcn_readfirstlane((int)((long)p.n_tiles * (split + 1) / n_splits));
  const int n_tiles = nt_end - nt_begin, total = n_tiles * NK;
  const int m0 = panel * 128;
  const bool probe = p.clk && blockIdx.x == 0 && tid == 0;
  if (probe) { p.clk[0] = __builtin_amdgcn_s_memtime(); p.clk[1] = __builtin_amdgcn_s_memrealtime(); }
  const WgStamp wgs = wg_stamp_begin(p.clk);

  // KD_EPI_RESIDUAL (out projection: no norm, + x): the accumulators START from the residual, read straight into the C layout (lane
  // (l31, lh), block j, register 4 g + e <-> row l31, feature 32 j + 8 g + 4 lh + e: 16 bytes per load, 32 contiguous bytes per row and
  // instruction) at kernel entry, in flight behind the row staging.  One n-tile per workgroup (the host forces n_splits = n_tiles): a
  // vector load between two tiles would make hipcc drain the weight ring (vmcnt(0)).
  f32x16 acc[4];
  if constexpr (EPI == KD_EPI_RESIDUAL) {
    const int rrow = min(m0 + wid * 32 + l31, p.M - 1);
    const float* rp = p.R + (size_t)rrow * p.N + nt_begin * NCOL + 4 * lh;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(rp + 32 * j + 8 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[j][4 * g + e] = v[e];
      }
  }

  const char* wp = p.Wp + (size_t)nt_begin * NK * STG + wid * (PB * 1024) + lane * 16;
  // stage s of this workgroup's W stream -> ring slot s % NSTG.  Stages past the end re-request the last one (into a slot nobody reads
  // any more): every wave then issues exactly PB pieces per stage, which keeps the vmcnt arithmetic uniform and lets the requests sit
  // INSIDE the MFMA group of a stage instead of behind a branch
  auto issue_piece = [&](int s_, int j) {
    const int s = min(s_, total - 1);
    const char* src = wp + (size_t)s * STG;
    char* dst = smem + (s_ % NSTG) * STG + wid * (PB * 1024);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + j * 1024),
                                     (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, 0, 0);
  };
  auto issue = [&](int s_) {
#pragma unroll
    for (int j = 0; j < PB; ++j) issue_piece(s_, j);
  };

  // ---- this wave's 32 rows -> a_hi / a_lo: chunk c of the lane's row holds k = 16 c + 8 lh .. + 7 -----------------------------------
  const int row = m0 + wid * 32 + l31;
  const bool ok = row < p.M;
  const int rowc = ok ? row : p.M - 1;
  bf16x8 a_hi[AG ? 1 : NC], a_lo[AG ? 1 : NC];
  float rs;
  {
    char* stage = smem + wid * WAREA;
    char* scl = smem + NSTG * STG + wid * SCL;
    const int r_first = min(m0 + wid * 32, p.M - 1), r_last = min(m0 + wid * 32 + 31, p.M - 1);
    constexpr bool nrm = EPI != KD_EPI_RESIDUAL;         // (the residual projection has no norm in front: the row is split as it is)
    const bool uni = nrm && (p.scale_stride == 0 || r_first / p.rows_per_sample == r_last / p.rows_per_sample);
    if (uni) {     // the sample's scale vector -> LDS (K * 4 bytes; lanes past the vector re-read its start: a whole 1 KiB piece lands)
      const char* ssrc = reinterpret_cast<const char*>(p.scale + (size_t)(r_first / p.rows_per_sample) * p.scale_stride);
#pragma unroll
      for (int i = 0; i < SCL / 1024; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ssrc + (i * 1024 + lane * 16) % (K * 4)),
                                         (__attribute__((address_space(3))) void*)(scl + i * 1024), 16, 0, 0);
    }
    const float* sp = p.scale + (size_t)(rowc / p.rows_per_sample) * p.scale_stride + 8 * lh;     // (not uni: straight from memory)
    const float* spl = reinterpret_cast<const float*>(scl) + 8 * lh;
    float ssq = 0.f;
    static_for<NR>([&](auto r_) {
      constexpr int r = decltype(r_)::value;
#pragma unroll
      for (int i = 0; i < PIECES; ++i) {
        const int ci = i * 64 + lane, rr = ci / CPR, qs = ci % CPR;
        const int grow = min(m0 + wid * 32 + rr, p.M - 1);
        const char* src = reinterpret_cast<const char*>(p.A + (size_t)grow * K + r * KH) + ((qs ^ (rr & 15)) << 4);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(stage + i * 1024), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // wave-private area: no barrier
      const char* rowp = stage + l31 * (KH * 4);
      static_for<NC / NR / 4>([&](auto c4_) {
        constexpr int c0 = r * (NC / NR) + 4 * decltype(c4_)::value;       // first of four 16-k chunks of the row
        f32x4 x0[4], x1[4], s0[4], s1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int q = 4 * (c0 - r * (NC / NR) + u) + 2 * lh;             // 16-byte chunk inside this round's part of the row
          x0[u] = *reinterpret_cast<const f32x4*>(rowp + ((q ^ (l31 & 15)) << 4));
          x1[u] = *reinterpret_cast<const f32x4*>(rowp + (((q + 1) ^ (l31 & 15)) << 4));
          if constexpr (nrm) {
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
#pragma unroll
          for (int e = 0; e < 4; ++e) ssq = fmaf(x0[u][e], x0[u][e], fmaf(x1[u][e], x1[u][e], ssq));
          u32x4 hi, lo;
          if constexpr (nrm) split8(x0[u] * s0[u], x1[u] * s1[u], hi, lo);
          else split8(x0[u], x1[u], hi, lo);
          if constexpr (AG) {
            areg_write4<8 * (c0 + u)>(hi);
            areg_write4<8 * (c0 + u) + 4>(lo);
          } else {
            asm volatile("" : "+v"(hi), "+v"(lo));     // materialise the fragments here (keeps x / scale registers short-lived)
            a_hi[c0 + u] = __builtin_bit_cast(bf16x8, hi);
            a_lo[c0 + u] = __builtin_bit_cast(bf16x8, lo);
          }
        });
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (NR > 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the area is overwritten by the next round
    });
    ssq += __shfl_xor(ssq, 32, 64);
    rs = nrm ? rsqrtf(ssq / (float)K + p.eps) : 1.f;
  }
  float py = 0.f, px = 0.f;
  if (EPI == KD_EPI_QKV) {
    float* qk_tab = reinterpret_cast<float*>(smem + NSTG * STG + 4 * SCL + 4 * 2048);
    if (tid < p.n_heads * 8) qk_tab[tid] = p.freq[tid];
    if (tid < p.n_heads) qk_tab[128 + tid] = sqrtf(p.qk_scale[tid]);
    const int tok = rowc % p.rows_per_sample;
    py = p.pos[2 * tok];
    px = p.pos[2 * tok + 1];
    asm volatile("" : "+v"(py), "+v"(px));            // consumed HERE as far as the compiler knows: its wait for the two loads lands before
  }                                                   // the ring starts, not as a vmcnt(0) in the first epilogue
  code_warm_end(warm);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  KD_BARRIER();                                        // every wave has taken its rows out of the slots it borrowed
  const bool full_panel = m0 + 128 <= p.M;
#pragma unroll
  for (int s = 0; s < PDIST; ++s) issue(s);
  if (probe) p.clk[4] = __builtin_amdgcn_s_memtime();          // end of the row prologue

  const int o0 = swz64(l31, lh), o1 = swz64(l31, 2 + lh);      // fragment offsets of the two 16-k chunks of a stage
  char* strip = smem + NSTG * STG + 4 * SCL + wid * 2048;      // this wave's [32 rows][16 floats] store strip
  const char* qkc = smem + NSTG * STG + 4 * SCL + 4 * 2048;    // [n_heads <= 16][8] RoPE frequencies, then [16] sqrt(cosine-sim scale)
  float* st_row[2];                                            // the two rows this lane STORES (lane = 4 * row + 16-byte piece), + its piece
  bool st_ok[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int r = m0 + wid * 32 + 16 * it + (lane >> 2);
    st_ok[it] = r < p.M;
    st_row[it] = p.C + (size_t)min(r, p.M - 1) * p.N + 4 * (lane & 3);
  }
  if constexpr (EPI != KD_EPI_RESIDUAL) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  }
  if constexpr (AG) asm volatile("s_nop 3" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));

  bf16x8 wh[2][4], wl[2][4];
  // KD_EPI_UNPATCH_NCHW (the out patch projection, N = 16 chan <= 64 outputs in ONE n-tile): the lane's W rows are read in the order
  // n' = (py chan + c) 4 + px instead of the reference's n = (py 4 + px) chan + c, so that an accumulator group of 4 registers is the 4
  // horizontally adjacent pixels of one image row and channel: 16-byte image accesses, 512 contiguous bytes per half-wave (the same
  // re-ordering the bf16 patch kernels do at pack time: here it is only a permuted row index of the fragment reads).  Blocks 2, 3 are empty.
  int ou0[2] = {0, 0}, ou1[2] = {0, 0};
  if constexpr (EPI == KD_EPI_UNPATCH_NCHW) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int np = 32 * j + l31, q = np >> 2, py = q / p.chan, cch = q - py * p.chan;
      const int n = np < 16 * p.chan ? (py * 4 + (np & 3)) * p.chan + cch : np;        // rows past N are zero in the packed tile
      ou0[j] = swz64(n, lh);
      ou1[j] = swz64(n, 2 + lh);
    }
  }
  auto read_frags = [&](int slot, int h, bf16x8 (&fh)[4], bf16x8 (&fl)[4]) {
    if constexpr (EPI == KD_EPI_UNPATCH_NCHW) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const char* st = smem + slot * STG + (h ? ou1[j] : ou0[j]);
        fh[j] = *reinterpret_cast<const bf16x8*>(st);
        fl[j] = *reinterpret_cast<const bf16x8*>(st + IMG);
      }
    } else {
      const char* st = smem + slot * STG + (h ? o1 : o0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        fh[j] = *reinterpret_cast<const bf16x8*>(st + j * 32 * 64);
        fl[j] = *reinterpret_cast<const bf16x8*>(st + IMG + j * 32 * 64);
      }
    }
  };
  // stage 0 in: its first chunk's fragments
  wait_vm(PB * (PDIST - 1));
  KD_BARRIER();
  read_frags(0, 0, wh[0], wl[0]);

  for (int nt = 0; nt < n_tiles; ++nt) {
    static_for<NK>([&](auto ks_) {
      constexpr int ks = decltype(ks_)::value;
      const int s = nt * NK + ks;
      // acc[j] += W fragment (hi / lo of buffer b, block j) x activation fragment (hi / lo) of chunk 2 ks + b
      auto mm = [&](auto b_, int j, bool w_lo, auto a_lo_) {
        constexpr int b = decltype(b_)::value, al = decltype(a_lo_)::value, c = 2 * ks + b;
        if (EPI == KD_EPI_UNPATCH_NCHW && j >= 2) return;          // (N <= 64: two W blocks)
        const bf16x8& w = w_lo ? wl[b][j] : wh[b][j];
        if constexpr (AG) mfma_ag<8 * c + 4 * al>(acc[j], w);
        else mfma_a(acc[j], w, al ? a_lo[c] : a_hi[c]);
      };
      constexpr std::integral_constant<int, 0> I0{};
      constexpr std::integral_constant<int, 1> I1{};
      // ---- chunk 0 of stage s (fragments in wh[0] / wl[0]): ONE MFMA, then the 8 fragment reads of chunk 1, then the other 11 MFMAs.
      // hipcc's wait for this chunk's fragments is an lgkmcnt(0) (it does not count across the stage's branches): in front of the first
      // MFMA it only sees reads that have had 12 MFMAs to land; behind the new reads it would wait for them as well --------------------
      if constexpr (AG) {
        mm(I0, 0, true, I0);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(ks % NSTG, 1, wh[1], wl[1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 1; j < 4; ++j) mm(I0, j, true, I0);
      } else {
        read_frags(ks % NSTG, 1, wh[1], wl[1]);
#pragma unroll
        for (int j = 0; j < 4; ++j) mm(I0, j, true, I0);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) mm(I0, j, false, I1);
#pragma unroll
      for (int j = 0; j < 4; ++j) mm(I0, j, false, I0);
      if constexpr (!AG) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 11, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- stage s + 1 in (everyone is past stage s - 1: its slot is refilled) -----------------------------------------------------------
      {
        // behind stage s + 1 in this wave's queue: the PDIST - 2 stages requested after it and the stores of an epilogue that ran since
        // its request (stage s + 1 was requested in the middle of stage s + 1 - PDIST; tile nt - 1's epilogue ran after stage nt NK - 1)
        int allow = PB * (PDIST - 2);
        if (full_panel && nt > 0 && ks + 2 <= PDIST) allow += NST;
        wait_vm(allow);
        KD_BARRIER();
      }
      // ---- chunk 1 of stage s.  Issue order pinned block by block: one MFMA, the 8 fragment reads of the next stage's first chunk, then
      // the 4 LDS-DMA requests of stage s + PDIST one at a time between the remaining MFMAs (an LDS-DMA instruction costs 60 - 180
      // issue cycles: in MFMA shadows instead of in front of the group) --------------------------------------------------------------
      mm(I1, 0, true, I0);
      if constexpr (AG) __builtin_amdgcn_sched_barrier(0);
      read_frags((ks + 1) % NSTG, 0, wh[0], wl[0]);
      if constexpr (!AG) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      mm(I1, 1, true, I0);
      mm(I1, 2, true, I0);
      issue_piece(s + PDIST, 0);
      __builtin_amdgcn_sched_barrier(0);
      mm(I1, 3, true, I0);
      mm(I1, 0, false, I1);
      mm(I1, 1, false, I1);
      issue_piece(s + PDIST, 1);
      __builtin_amdgcn_sched_barrier(0);
      mm(I1, 2, false, I1);
      mm(I1, 3, false, I1);
      mm(I1, 0, false, I0);
      issue_piece(s + PDIST, 2);
      __builtin_amdgcn_sched_barrier(0);
      mm(I1, 1, false, I0);
      mm(I1, 2, false, I0);
      issue_piece(s + PDIST, 3);
      mm(I1, 3, false, I0);
      __builtin_amdgcn_sched_barrier(0);
    });
    // asm MFMAs: 16-pass XDL write -> vector read of the accumulators needs 18 wait states the compiler does not know about
    if constexpr (AG) asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
    if (probe && nt == 0) p.clk[5] = __builtin_amdgcn_s_memtime();    // end of the first tile's K loop

    // ---- epilogue of n-tile nt, in the lane that owns the row: features n0 + 32 j + 8 g + 4 lh + (0..3) per accumulator group g -------
    const int n0 = (nt_begin + nt) * NCOL;
    // One 32-feature block of the wave's 32 rows -> memory.  Stored straight from the accumulator layout a wave-instruction writes 32
    // rows x 32 bytes: 64 separate 16-byte requests, and the epilogue waits on the address path (7 000 of a 16 000-clock tile at level 1,
    // benchmarks/x3_bench.py time line).  Through a wave-private LDS strip ([32 rows][16 floats], two passes per block, chunk position
    // XOR (row >> 2) & 1: conflict-free both ways) four consecutive lanes hold 64 contiguous bytes of ONE row, so an instruction writes
    // 16 rows x 64 bytes as 16 requests.  LDS executes a wave's operations in order: no wait between the strip writes and reads.
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
    if (GEGLU) {
      const float rsh = 0.5f * rs;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        f32x4 blk[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x2 a = geglu_pair(f32x2{acc[2 * jj][4 * g], acc[2 * jj][4 * g + 1]} * rsh, f32x2{acc[2 * jj + 1][4 * g], acc[2 * jj + 1][4 * g + 1]} * rs);
          const f32x2 b = geglu_pair(f32x2{acc[2 * jj][4 * g + 2], acc[2 * jj][4 * g + 3]} * rsh, f32x2{acc[2 * jj + 1][4 * g + 2], acc[2 * jj + 1][4 * g + 3]} * rs);
          blk[g] = f32x4{a.x, a.y, b.x, b.y};
        }
        if (p.c_split) {            // (uniform branch) the down projection's A operand: bf16 hi / lo planes, 16-byte stores after a half-wave exchange
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
          const size_t off = (size_t)rowc * p.N + n0 + 32 * jj;
          b16::store_block_bf16(reinterpret_cast<b16::u16*>(p.C) + off, hi, lh, ok);
          b16::store_block_bf16(p.Cl + off, lo, lh, ok);
        } else {
          store_block(blk, n0 + 32 * jj);
        }
      }
    } else if (EPI == KD_EPI_QKV) {
#pragma unroll
      for (int vv = 0; vv < 2; ++vv) {
        const int vec = (n0 >> 6) + vv;                       // (q | k | v, head) vector index of these 64 columns
        const int which = vec >= 2 * p.n_heads ? 2 : (vec >= p.n_heads ? 1 : 0), head = vec - which * p.n_heads;     // (no division: scalar)
        f32x16& a0 = acc[2 * vv];
        f32x16& a1 = acc[2 * vv + 1];
        if (which < 2) {
          // the head's RoPE frequencies (this lane's four: 4 lh ..) and sqrt(cosine-sim scale) from the table parked in LDS by the prologue:
          // one ds_read_b128 + one ds_read_b32, no vector-memory load beside the LDS-DMA ring and no per-element selects
          const f32x4 fv = *reinterpret_cast<const f32x4*>(qkc + (head * 8 + 4 * lh) * 4);
          const float qsc = *reinterpret_cast<const float*>(qkc + 512 + head * 4);
          const float fr[4] = {fv[0], fv[1], fv[2], fv[3]};
          b16::qk_prep_blocks(a0, a1, rs, qsc, p.eps, py, px, fr);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) { a0[r] *= rs; a1[r] *= rs; }
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          f32x4 blk[4];
          const f32x16& a = jj ? a1 : a0;
          if (p.qkv_packed) {                   // (one uniform branch per block, not a select per element)
#pragma unroll
            for (int g = 0; g < 4; ++g) blk[g] = pack_split4(f32x4{a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]});
          } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) blk[g] = f32x4{a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
          }
          store_block(blk, n0 + 64 * vv + 32 * jj);
        }
      }
    } else if (EPI == KD_EPI_UNPATCH_NCHW) {
      // tokens -> NCHW image (image_transformer_v2.py:758-760) with the Denoiser's c_out * y + c_skip * x_in (layers.py:90): accumulator
      // group 8 j + 2 g + lh of the lane's token = (py, channel), its 4 registers = the 4 pixels of that patch row
      const int per = p.gh * p.gw, bb = rowc / per, rr = rowc - bb * per, ty = rr / p.gw, tx = rr - ty * p.gw;
      float c_out = 1.f, c_skip = 0.f;
      if (p.sigma) {
        const float sg = p.sigma[bb], sd = p.sigma_data, var = sg * sg + sd * sd;
        c_skip = sd * sd / var;
        c_out = sg * sd / sqrtf(var);
      }
      const int Himg = 4 * p.gh, Wimg = 4 * p.gw;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int q = 8 * j + 2 * g + lh, py = q / p.chan, cch = q - py * p.chan;
          if (q < 4 * p.chan && ok) {
            const size_t o = (((size_t)bb * p.chan + cch) * Himg + 4 * ty + py) * Wimg + 4 * tx;
            f32x4 v = f32x4{acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]} * (rs * c_out);
            if (p.sigma) v = v + *reinterpret_cast<const f32x4*>(p.R + o) * c_skip;
            st16(p.C + o, v);
          }
        }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 blk[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) blk[g] = f32x4{acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]} * rs + p.out_add;
        store_block(blk, n0 + 32 * j);
      }
    }
    if (probe && nt == 0) p.clk[6] = __builtin_amdgcn_s_memtime();    // end of the first tile's epilogue
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // v_mov -> SrcC of an asm MFMA (K = 512): the compiler pads that hazard for its own MFMAs only
    if constexpr (AG) asm volatile("s_nop 3" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the clamped tail requests still target this workgroup's LDS
  if (probe) { p.clk[2] = __builtin_amdgcn_s_memtime(); p.clk[3] = __builtin_amdgcn_s_memrealtime(); p.clk[7] = (unsigned long long)total; }
  wg_stamp_end(wgs);
}

// ===========================================================================================================================================
// K = 256, TWO workgroups per CU.  At one workgroup per CU (NC = 16 above: 128 + 64 + 64 registers of fragments and accumulators per wave)
// nothing runs while a wave is in its row prologue or in an epilogue: the level-1 qkv projection spends 5.4 k of every 14 k clocks per
// n-tile in the cosine-sim / RoPE / store epilogue with the matrix pipe idle (benchmarks/x3_bench.py time line).  Same cure as ffn_x3.hip's
// width-128 kernel: the row fragments go to the named AccVGPRs a0..a127 (all of that half of a two-wave budget), the n range is walked in
// HALF tiles of 64 W rows (one head vector, or 32 GEGLU outputs: 32 accumulator registers), whose rows are one contiguous 4 KiB run of each
// [128 rows][32 k] image of the packed weight, through a 4-stage ring (a stage = two 32-k sub-stages [hi 4 KiB | lo 4 KiB], 24 MFMAs).
// <= 128 ArchVGPRs, 77 KiB of LDS: two independent workgroups per CU fill each other's prologues and epilogues.
template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_x3h_kernel(const XArgs p) {
  constexpr int NC = 16, K = 256, NK = 8;                       // 32-k sub-stages of an n-tile in the packed image
  constexpr int NSTG = 4, PDIST = NSTG - 1, PB = 4, UNIT = 4;   // ring stages per half tile
  constexpr bool GEGLU = EPI == KD_EPI_GEGLU;
  constexpr int HCOL = GEGLU ? 32 : 64;                         // output columns of a half tile
  constexpr int NST = GEGLU ? 4 : 8;                            // 16-byte stores per lane per half tile
  constexpr int KH = 128, NR = K / KH, CPR = KH / 4, PIECES = 32 * CPR / 64, SCL = 1024, SUB = 8192, HALF = 4096, CB = 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const auto warm = code_warm_begin<20 * 1024>((int)blockIdx.x < p.warm && tid < 64);
  int panel, split;
  const int n_splits = p.n_splits, n_panels = gridDim.x / n_splits;
  if ((n_panels & 7) == 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    panel = (j / n_splits) * 8 + xcd;
    split = j % n_splits;
  } else {
    panel = blockIdx.x % n_panels;
    split = blockIdx.x / n_panels;
  }
  panel = __builtin_amdgcn_readfirstlane(panel);
  split = __builtin_amdgcn_readfirstlane(split);
  const int ht_begin = __builtin_amdgcn_readfirstlane((int)((long)p.n_tiles * split / n_splits));      // (n_tiles counts HALF tiles here)
  const int ht_end = __builtin_amdgcn_readfirstlane((int)((long)p.n_tiles * (split + 1) / n_splits));
  const int n_ht = ht_end - ht_begin, total = n_ht * UNIT;
  const int m0 = panel * 128;
  const bool probe = p.clk && blockIdx.x == (gridDim.x * 5) / 8 && tid == 0;
  if (probe) { p.clk[0] = __builtin_amdgcn_s_memtime(); p.clk[1] = __builtin_amdgcn_s_memrealtime(); }
  const WgStamp wgs = wg_stamp_begin(p.clk);

  // stage q of this workgroup's stream: half tile ht_begin + q / 4, sub-stages 2 (q % 4), + 1 of its n-tile; piece j: sub-stage j >> 1, hi / lo
  // image j & 1 -- this wave's KiB of the half tile's 4 KiB run.  Past the end: the last stage again (never read).
  const char* wp = p.Wp + wid * 1024 + lane * 16;
  auto issue_piece = [&](int q_, int j) {
    const int q = min(q_, total - 1), ht = ht_begin + (q >> 2), u = q & 3;
    const char* src = wp + ((size_t)(ht >> 1) * NK + 2 * u + (j >> 1)) * STG + (j & 1) * IMG + (ht & 1) * HALF;
    char* dst = smem + (q_ % NSTG) * STG + (j >> 1) * SUB + (j & 1) * HALF + wid * 1024;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };

  // ---- this wave's 32 rows -> a0..a127 (two staging rounds of 128 floats per row) ------------------------------------------------------
  const int row = m0 + wid * 32 + l31;
  const bool ok = row < p.M;
  const int rowc = ok ? row : p.M - 1;
  float rs;
  {
    char* stage = smem + wid * STG;
    char* scl = smem + NSTG * STG + wid * SCL;
    const int r_first = min(m0 + wid * 32, p.M - 1), r_last = min(m0 + wid * 32 + 31, p.M - 1);
    const bool uni = p.scale_stride == 0 || r_first / p.rows_per_sample == r_last / p.rows_per_sample;
    if (uni) {
      const char* ssrc = reinterpret_cast<const char*>(p.scale + (size_t)(r_first / p.rows_per_sample) * p.scale_stride);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ssrc + lane * 16),
                                       (__attribute__((address_space(3))) void*)scl, 16, 0, 0);
    }
    const float* sp = p.scale + (size_t)(rowc / p.rows_per_sample) * p.scale_stride + 8 * lh;
    const float* spl = reinterpret_cast<const float*>(scl) + 8 * lh;
    float ssq = 0.f;
    static_for<NR>([&](auto r_) {
      constexpr int r = decltype(r_)::value;
#pragma unroll
      for (int i = 0; i < PIECES; ++i) {
        const int ci = i * 64 + lane, rr = ci / CPR, qs = ci % CPR;
        const int grow = min(m0 + wid * 32 + rr, p.M - 1);
        const char* src = reinterpret_cast<const char*>(p.A + (size_t)grow * K + r * KH) + ((qs ^ (rr & 15)) << 4);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(stage + i * 1024), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const char* rowp = stage + l31 * (KH * 4);
      static_for<NC / NR / CB>([&](auto c4_) {           // CB chunks at a time (registers: 128 ArchVGPRs for everything)
        constexpr int c0 = r * (NC / NR) + CB * decltype(c4_)::value;
        f32x4 x0[CB], x1[CB], s0[CB], s1[CB];
#pragma unroll
        for (int u = 0; u < CB; ++u) {
          const int q = 4 * (c0 - r * (NC / NR) + u) + 2 * lh;
          x0[u] = *reinterpret_cast<const f32x4*>(rowp + ((q ^ (l31 & 15)) << 4));
          x1[u] = *reinterpret_cast<const f32x4*>(rowp + (((q + 1) ^ (l31 & 15)) << 4));
          if (uni) {
            s0[u] = *reinterpret_cast<const f32x4*>(spl + 16 * (c0 + u));
            s1[u] = *reinterpret_cast<const f32x4*>(spl + 16 * (c0 + u) + 4);
          } else {
            s0[u] = *reinterpret_cast<const f32x4*>(sp + 16 * (c0 + u));
            s1[u] = *reinterpret_cast<const f32x4*>(sp + 16 * (c0 + u) + 4);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        static_for<CB>([&](auto u_) {
          constexpr int u = decltype(u_)::value;
#pragma unroll
          for (int e = 0; e < 4; ++e) ssq = fmaf(x0[u][e], x0[u][e], fmaf(x1[u][e], x1[u][e], ssq));
          u32x4 hi, lo;
          split8(x0[u] * s0[u], x1[u] * s1[u], hi, lo);
          areg_write4_lo<8 * (c0 + u)>(hi);
          areg_write4_lo<8 * (c0 + u) + 4>(lo);
        });
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (r + 1 < NR) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the area is overwritten by the next round
    });
    ssq += __shfl_xor(ssq, 32, 64);
    rs = rsqrtf(ssq / (float)K + p.eps);
  }
  float py = 0.f, px = 0.f;
  if (EPI == KD_EPI_QKV) {
    float* qk_tab = reinterpret_cast<float*>(smem + NSTG * STG + 4 * SCL + 4 * 2048);
    if (tid < p.n_heads * 8) qk_tab[tid] = p.freq[tid];
    if (tid < p.n_heads) qk_tab[128 + tid] = sqrtf(p.qk_scale[tid]);
    const int tok = rowc % p.rows_per_sample;
    py = p.pos[2 * tok];
    px = p.pos[2 * tok + 1];
    asm volatile("" : "+v"(py), "+v"(px));
  }
  code_warm_end(warm);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  KD_BARRIER();                                        // every wave has taken its rows out of the slot it borrowed
  const bool full_panel = m0 + 128 <= p.M;
#pragma unroll
  for (int s = 0; s < PDIST; ++s)
#pragma unroll
    for (int j = 0; j < PB; ++j) issue_piece(s, j);
  if (probe) p.clk[4] = __builtin_amdgcn_s_memtime();

  const int o0 = swz64(l31, lh), o1 = swz64(l31, 2 + lh);
  char* strip = smem + NSTG * STG + 4 * SCL + wid * 2048;
  const char* qkc = smem + NSTG * STG + 4 * SCL + 4 * 2048;
  float* st_row[2];
  bool st_ok[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int r = m0 + wid * 32 + 16 * it + (lane >> 2);
    st_ok[it] = r < p.M;
    st_row[it] = p.C + (size_t)min(r, p.M - 1) * p.N + 4 * (lane & 3);
  }
  f32x16 acc[2];
  bf16x8 uh[2][2], ul[2][2];                           // [chunk parity][W block of the half tile]
  auto read_up = [&](int slot, int cc, bf16x8 (&fh)[2], bf16x8 (&fl)[2]) {
    const char* st = smem + slot * STG + (cc >> 1) * SUB + ((cc & 1) ? o1 : o0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      fh[j] = *reinterpret_cast<const bf16x8*>(st + j * 32 * 64);
      fl[j] = *reinterpret_cast<const bf16x8*>(st + HALF + j * 32 * 64);
    }
  };
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
  wait_vm(PB * (PDIST - 1));
  // (Starting one of a CU's two workgroups half a period late -- by wave-slot parity or by split parity, s_sleep -- so that their K loops
  // and epilogues interleave was measured slower: 52.3 / 53.8 vs 49.0 us.  The tile phase is close to what the stores allow: 100 MB in
  // ~30 us at level 1.)
  KD_BARRIER();
  read_up(0, 0, uh[0], ul[0]);

  constexpr std::integral_constant<int, 0> I0{};
  constexpr std::integral_constant<int, 1> I1{};
  constexpr std::integral_constant<int, 2> I2{};
  constexpr std::integral_constant<int, 3> I3{};
  constexpr std::integral_constant<int, 4> I4{};
  constexpr std::integral_constant<int, 5> I5{};
  for (int h = 0; h < n_ht; ++h) {
    if (probe && h == 1) p.clk[8] = __builtin_amdgcn_s_memtime();
    static_for<UNIT>([&](auto u_) {
      constexpr int u = decltype(u_)::value;
      const int s = h * UNIT + u;
      // the 6 MFMAs of chunk cc: (w_lo x a_hi), (w_hi x a_lo), (w_hi x a_hi) for the two W blocks in alternation
      auto mm = [&](auto cc_, auto i_) {
        constexpr int cc = decltype(cc_)::value, i = decltype(i_)::value, j = i & 1, term = i >> 1, c = 4 * u + cc;
        const bf16x8& w = term == 0 ? ul[cc & 1][j] : uh[cc & 1][j];
        constexpr int al = term == 1;
        if constexpr (c == 0 && term == 0) mfma_ag0<8 * c + 4 * al>(acc[j], w);      // first MFMA of the chain: C = 0
        else mfma_ag<8 * c + 4 * al>(acc[j], w);
      };
      mm(I0, I0);
      __builtin_amdgcn_sched_barrier(0);
      read_up(u, 1, uh[1], ul[1]);
      __builtin_amdgcn_sched_barrier(0);
      mm(I0, I1); mm(I0, I2); mm(I0, I3); mm(I0, I4); mm(I0, I5);
      mm(I1, I0);
      __builtin_amdgcn_sched_barrier(0);
      read_up(u, 2, uh[0], ul[0]);
      __builtin_amdgcn_sched_barrier(0);
      mm(I1, I1); mm(I1, I2); mm(I1, I3); mm(I1, I4); mm(I1, I5);
      __builtin_amdgcn_sched_barrier(0);
      {
        // behind stage s + 1 in this wave's queue: the stage requested after it and the stores of an epilogue that ran since its request
        int allow = PB * (PDIST - 2);
        if (full_panel && h > 0 && u + 2 <= PDIST) allow += NST;
        wait_vm(allow);
        KD_BARRIER();
      }
      mm(I2, I0);
      __builtin_amdgcn_sched_barrier(0);
      read_up(u, 3, uh[1], ul[1]);
      __builtin_amdgcn_sched_barrier(0);
      mm(I2, I1); mm(I2, I2);
      issue_piece(s + PDIST, 0);
      __builtin_amdgcn_sched_barrier(0);
      mm(I2, I3); mm(I2, I4); mm(I2, I5);
      issue_piece(s + PDIST, 1);
      __builtin_amdgcn_sched_barrier(0);
      mm(I3, I0);
      __builtin_amdgcn_sched_barrier(0);
      read_up((u + 1) % NSTG, 0, uh[0], ul[0]);
      __builtin_amdgcn_sched_barrier(0);
      mm(I3, I1); mm(I3, I2);
      issue_piece(s + PDIST, 2);
      __builtin_amdgcn_sched_barrier(0);
      mm(I3, I3); mm(I3, I4);
      issue_piece(s + PDIST, 3);
      mm(I3, I5);
      __builtin_amdgcn_sched_barrier(0);
    });
    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[0]), "+v"(acc[1]));      // asm MFMA results -> vector reads
    if (probe && h == 1) p.clk[9] = __builtin_amdgcn_s_memtime();

    // ---- epilogue of the half tile, in the lane that owns the row -----------------------------------------------------------------------
    const int n0 = (ht_begin + h) * HCOL;
    if (GEGLU) {
      const float rsh = 0.5f * rs;
      f32x4 blk[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x2 a = geglu_pair(f32x2{acc[0][4 * g], acc[0][4 * g + 1]} * rsh, f32x2{acc[1][4 * g], acc[1][4 * g + 1]} * rs);
        const f32x2 b = geglu_pair(f32x2{acc[0][4 * g + 2], acc[0][4 * g + 3]} * rsh, f32x2{acc[1][4 * g + 2], acc[1][4 * g + 3]} * rs);
        blk[g] = f32x4{a.x, a.y, b.x, b.y};
      }
      store_block(blk, n0);
    } else if (EPI == KD_EPI_QKV) {
      const int vec = n0 >> 6;                              // (q | k | v, head) vector index of these 64 columns
      const int which = vec >= 2 * p.n_heads ? 2 : (vec >= p.n_heads ? 1 : 0), head = vec - which * p.n_heads;
      if (which < 2) {
        const f32x4 fv = *reinterpret_cast<const f32x4*>(qkc + (head * 8 + 4 * lh) * 4);
        const float qsc = *reinterpret_cast<const float*>(qkc + 512 + head * 4);
        const float fr[4] = {fv[0], fv[1], fv[2], fv[3]};
        b16::qk_prep_blocks(acc[0], acc[1], rs, qsc, p.eps, py, px, fr);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][r] *= rs; acc[1][r] *= rs; }
      }
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        f32x4 blk[4];
        const f32x16& a = acc[jj];
        if (p.qkv_packed) {
#pragma unroll
          for (int g = 0; g < 4; ++g) blk[g] = pack_split4(f32x4{a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]});
        } else {
#pragma unroll
          for (int g = 0; g < 4; ++g) blk[g] = f32x4{a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
        }
        store_block(blk, n0 + 32 * jj);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x4 blk[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) blk[g] = f32x4{acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]} * rs + p.out_add;
        store_block(blk, n0 + 32 * j);
      }
    }
    if (probe && h == 1) p.clk[10] = __builtin_amdgcn_s_memtime();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the clamped tail requests still target this workgroup's LDS
  if (probe) { p.clk[2] = __builtin_amdgcn_s_memtime(); p.clk[3] = __builtin_amdgcn_s_memrealtime(); p.clk[7] = (unsigned long long)total; }
  wg_stamp_end(wgs);
}

unsigned long long* g_clk = nullptr;


template <int NC, int EPI>
static int launch(const XArgs& a0, const char* nm, double flops, double bytes, hipStream_t s, int force_splits = 0) {
  auto kern = gemm_x3_astat_kernel<NC, EPI>;
  constexpr int K = NC * 16, NSTG = NC <= 8 ? 4 : 8;
  constexpr int LDS = NSTG * STG + 4 * (K * 4 < 1024 ? 1024 : K * 4) + 4 * 2048 + 1024;     // ring + scale vectors + store strips + per-head constants
  static LdsAttr attr_set;
  attr_set.ensure(reinterpret_cast<const void*>(kern), LDS);
  // n-splits of a panel: every workgroup pays the row prologue (about one n-tile's K loop) and then its share of the n-tiles; the grid
  // runs in ceil(workgroups / resident slots) rounds.  The divisor of n_tiles with the smallest rounds x (1 + tiles per split) wins
  // (ties: fewer splits = fewer redundant prologues).
  const int panels = (a0.M + 127) / 128, slots = (NC <= 8 ? 2 : 1) * cu_count();
  int best = 1;
  long best_cost = -1;
  for (int sp = 1; sp <= a0.n_tiles; ++sp) {
    if (a0.n_tiles % sp) continue;
    const long rounds = ((long)panels * sp + slots - 1) / slots;
    const long cost = rounds * (1 + a0.n_tiles / sp);
    if (best_cost < 0 || cost < best_cost) { best = sp; best_cost = cost; }
  }
  const int forced = force_splits ? force_splits : option("x3_splits", 0);
  XArgs a = a0;
  a.n_splits = forced > 0 && forced <= a0.n_tiles ? forced : best;
  LaunchScope prof(nm, flops, bytes, s);
  hipLaunchKernelGGL(kern, dim3((unsigned)(panels * a.n_splits)), dim3(256), LDS, s, a);
  return check_launch("kd_gemm_f32(x3 astat)");
}

template <int EPI>
static int launch_half(const XArgs& a0, const char* nm, double flops, double bytes, hipStream_t s) {
  auto kern = gemm_x3h_kernel<EPI>;
  constexpr int LDS = 4 * STG + 4 * 1024 + 4 * 2048 + 1024;
  static LdsAttr attr_set;
  attr_set.ensure(reinterpret_cast<const void*>(kern), LDS);
  // the same cost model as launch() with two resident workgroups per CU; a prologue costs about two half tiles
  const int panels = (a0.M + 127) / 128, slots = 2 * cu_count();
  int best = 1;
  long best_cost = -1;
  for (int sp = 1; sp <= a0.n_tiles; ++sp) {
    if (a0.n_tiles % sp) continue;
    const long rounds = ((long)panels * sp + slots - 1) / slots;
    const long cost = rounds * (2 + a0.n_tiles / sp);
    if (best_cost < 0 || cost < best_cost) { best = sp; best_cost = cost; }
  }
  const int forced = option("x3_splits", 0);
  XArgs a = a0;
  a.n_splits = forced > 0 && forced <= a0.n_tiles && a0.n_tiles % forced == 0 ? forced : best;
  LaunchScope prof(nm, flops, bytes, s);
  hipLaunchKernelGGL(kern, dim3((unsigned)(panels * a.n_splits)), dim3(256), LDS, s, a);
  return check_launch("kd_gemm_f32(x3 half tiles)");
}

}  // namespace x3

// Eligibility + dispatch (called by kd_gemm_f32 ahead of the round-1 A-stationary kernel).  Returns 1 if the descriptor was not taken.
int gemm_x3_try(const GemmP& d, hipStream_t s, int* rc) {
  using namespace x3;
  if (!option("x3", 1)) return 1;
  if (d.precision != KD_PREC_SPLIT3 || d.a_mode != KD_A_PLAIN || !d.Wp || d.debug) return 1;
  const bool unpatch = d.epi == KD_EPI_UNPATCH_NCHW;
  if (unpatch && (d.K != 128 || !d.norm || d.ph != 4 || d.pw != 4 || d.N != 16 * d.chan || d.N > 64 || !option("x3_unpatch", 1))) return 1;
  if (d.epi != KD_EPI_STORE && d.epi != KD_EPI_QKV && d.epi != KD_EPI_GEGLU && d.epi != KD_EPI_RESIDUAL && !unpatch) return 1;
  // norm -> wide projection, or (round 3) the plain residual projection behind the attention core: C = R + A W^T
  // (on request, option "x3_res": gemm_x3r.hip takes that shape by default, 24.2 us.  K = 512 only: 25.3 vs 27.1 us at the headline shape; at K = 128 it is level with the round-1 tile kernel (both at the memory roof) and
  // at K = 256 slower, 32.0 vs 26.4 us -- a workgroup there pays a whole row prologue for one or two n-tiles: benchmarks/x3_bench.py)
  if (d.epi == KD_EPI_RESIDUAL ? (d.norm || !d.R || d.K != 512 || !option("x3_res", 0)) : !d.norm) return 1;
  if (d.K != 128 && d.K != 256 && d.K != 512) return 1;
  const int ncol = d.epi == KD_EPI_GEGLU ? 64 : 128;
  if ((!unpatch && d.N % ncol) || d.M < option("x3_min_rows", 512) || (d.norm && d.rows_per_sample <= 0)) return 1;
  if (d.epi == KD_EPI_QKV && (!d.rope_pos || !d.rope_freq || d.n_heads > 16)) return 1;   // tables only / many heads: round-1 kernel
  if (d.c_split && (d.epi != KD_EPI_GEGLU || !d.C_lo || (d.N & 31))) return 1;
  XArgs a{};
  a.A = d.A; a.Wp = reinterpret_cast<const char*>(d.Wp); a.C = d.C;
  a.scale = d.norm ? d.scale : nullptr; a.R = d.R; a.scale_stride = d.scale_stride; a.rows_per_sample = d.rows_per_sample > 0 ? d.rows_per_sample : d.M; a.eps = d.eps;
  a.M = d.M; a.N = d.N; a.n_tiles = unpatch ? 1 : d.N / ncol; a.n_splits = 1;
  a.sigma = d.sigma; a.sigma_data = d.sigma_data; a.gh = d.gh; a.gw = d.gw; a.chan = d.chan;
  a.n_heads = d.n_heads; a.qk_scale = d.qk_scale; a.pos = d.rope_pos; a.freq = d.rope_freq; a.qkv_packed = d.qkv_packed;
  a.out_add = d.out_add;
  a.Cl = reinterpret_cast<b16::u16*>(d.C_lo); a.c_split = d.c_split;
  a.warm = d.warm;
  a.clk = g_clk;
  const double n_eff = d.epi == KD_EPI_GEGLU ? 2.0 * d.N : (double)d.N;
  const double flops = 2.0 * d.M * n_eff * d.K;
  const double bytes = 4.0 * ((double)d.M * d.K + n_eff * d.K + (double)d.M * d.N * (d.epi == KD_EPI_RESIDUAL || (unpatch && d.sigma) ? 2 : 1));
  char nm[96] = "gemm_x3_astat";
  if (prof_on()) snprintf(nm, sizeof(nm), "gemm_x3_astat<e%d> M=%d N=%d K=%d", d.epi, d.M, d.N, d.K);
  if (d.K == 256 && d.epi != KD_EPI_RESIDUAL && !d.c_split && option("x3_half", 1)) {      // two workgroups per CU, half tiles
    a.n_tiles = d.N / (d.epi == KD_EPI_GEGLU ? 32 : 64);
    if (prof_on()) snprintf(nm, sizeof(nm), "gemm_x3_astat<e%d,h> M=%d N=%d K=%d", d.epi, d.M, d.N, d.K);
    if (d.epi == KD_EPI_QKV) *rc = launch_half<KD_EPI_QKV>(a, nm, flops, bytes, s);
    else if (d.epi == KD_EPI_GEGLU) *rc = launch_half<KD_EPI_GEGLU>(a, nm, flops, bytes, s);
    else *rc = launch_half<KD_EPI_STORE>(a, nm, flops, bytes, s);
    return 0;
  }
#define KD_X3(NCV, EP) if (d.K == NCV * 16 && d.epi == EP) { *rc = launch<NCV, EP>(a, nm, flops, bytes, s, EP == KD_EPI_RESIDUAL ? a.n_tiles : 0); return 0; }
  KD_X3(8, KD_EPI_STORE) KD_X3(8, KD_EPI_QKV) KD_X3(8, KD_EPI_GEGLU)
  KD_X3(16, KD_EPI_STORE) KD_X3(16, KD_EPI_QKV) KD_X3(16, KD_EPI_GEGLU)
  KD_X3(32, KD_EPI_STORE) KD_X3(32, KD_EPI_QKV) KD_X3(32, KD_EPI_GEGLU)
  KD_X3(32, KD_EPI_RESIDUAL)
  KD_X3(8, KD_EPI_UNPATCH_NCHW)
#undef KD_X3
  return 1;
}

void x3_set_clock_buffer(unsigned long long* p) { x3::g_clk = p; }

}  // namespace kd

KD_TEXT_PAD(gemm_x3)      // last function of this code object: kd_common.h, code warm-up
