// Fused fp32-in / fp32-out GEMM for gfx950:  C = epilogue( prologue(A) @ W^T ), two arithmetic modes
// behind one kernel template (the reference's nn.Linear is fp32, image_transformer_v2.py:126-139):
//
//   KD_PREC_EXACT  v_mfma_f32_32x32x2_f32: bit-for-bit an fmaf chain, 157 TFLOP/s peak (1/16 of bf16).
//   KD_PREC_SPLIT3 every fp32 operand is split into hi = bf16(x), lo = bf16(x - hi) (16 significand bits
//                  kept) and x*w ~= hi*hi + hi*lo + lo*hi runs on v_mfma_f32_32x32x16_bf16 with fp32
//                  accumulation: 3 MFMAs at 16x the fp32 rate, relative error per product <= ~2^-15.
//                  A is split on its way into LDS (after the norm prologue); W is pre-split ONCE into a
//                  packed image ([n-tile][k-step][hi|lo][128 rows][32 bf16], LDS swizzle pre-applied,
//                  kd_pack_weight_bf16x3) that streams into LDS with global_load_lds (no VGPRs, no
//                  ds_write pass).  At the HDiT shapes (K = 128..512) this turns the level-0/1 GEMMs
//                  from MFMA-bound into HBM-bound.
//
// Tiling: 128x128 output tile per 256-thread workgroup (4 wave64s as 2x2, each wave a 64x64
// sub-tile = 2x2 MFMA 32x32 accumulators), K stepped by BK = 32 through a double-buffered LDS tile
// with register prefetch (one barrier per K-step), two workgroups per CU.
//   exact : LDS rows are padded to BK+4 floats so that the ds_read_b128 operand fetches (16 distinct
//           rows per lane group) are bank-conflict free.  Each lane fetches one float4 per operand per
//           4 MFMAs: MFMA step s of an 8-deep k-chunk consumes k = 4*(lane>>5) + s from both A and B
//           (any consistent k permutation is a valid GEMM).
//   split3: four bf16 images per stage (A_hi, A_lo, B_hi, B_lo), 64-byte rows, the 16-byte chunk index
//           XOR-swizzled with (row>>2)&3: the ds_read_b128 of an MFMA operand (32 rows x one chunk) then
//           touches every 16-byte slot of the 256-byte bank row exactly once per 16-lane group.
//
// Prologues / epilogues (all fused; see include/kdiff_hip.h):
//   A gather : plain | 2x2 token merge | NCHW patch gather * c_in(sigma)
//   norm     : per-row rsqrt(mean x^2 + eps) (accumulated while the A tile streams through) and
//              a per-(sample, k) scale applied on load  == AdaRMSNorm / RMSNorm
//   epilogue : store(+const) | + residual | GEGLU | 2x2 token split + lerp(skip) |
//              NCHW un-patch * c_out + x * c_skip
#include "kd_common.h"
#include <cstdlib>
#include <type_traits>

namespace kd {

constexpr int BM = 128, BN = 128, BK = 32;

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;

// exact mode: fp32 tiles, rows padded to S floats
constexpr int S = BK + 4;
constexpr int NLD = BM * BK / 4 / 256;        // float4 loads per thread per operand tile (= 4)
constexpr int RT = BK / 4;                    // threads that share one tile row (= 8)
constexpr size_t EPI_STRIPS = 4 * 8 * 64 * sizeof(float);   // 8 KiB
constexpr size_t LDS_EXACT = (size_t)(2 * BM * S + 2 * BN * S + BM) * sizeof(float);
// split3 mode: per stage A_hi | A_lo | B_hi | B_lo, each [128 rows][32 bf16] = 8 KiB
constexpr int IMG = BM * BK * 2;              // bytes of one bf16 image
constexpr int STAGE = 4 * IMG;                // 32 KiB
constexpr size_t LDS_SPLIT = (size_t)2 * STAGE + BM * sizeof(float) + EPI_STRIPS;
constexpr int WP_BLOCK = 2 * IMG;             // packed-weight bytes per (n-tile, k-step): hi image then lo image
constexpr int SCALE_TAB_MAX_K = 512;          // norm scales of a tile's sample staged in LDS when K fits (2 workgroups / CU stay resident)

// byte offset of (row, 16-byte chunk c) inside a swizzled [128][32] bf16 image
__device__ __host__ __forceinline__ int swz(int row, int c) { return row * 64 + ((c ^ ((row >> 2) & 3)) << 4); }

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
  bf16x2 v = {(__bf16)a, (__bf16)b};           // v_cvt_pk_bf16_f32 (round to nearest even)
  return __builtin_bit_cast(unsigned, v);
}
// x -> (hi, lo) bf16 pairs for 4 consecutive k
__device__ __forceinline__ void split4(const f32x4 v, u32x2& hi, u32x2& lo) {
  hi[0] = pack_bf16(v[0], v[1]);
  hi[1] = pack_bf16(v[2], v[3]);
  const float r0 = v[0] - __uint_as_float(hi[0] << 16), r1 = v[1] - __uint_as_float(hi[0] & 0xFFFF0000u);
  const float r2 = v[2] - __uint_as_float(hi[1] << 16), r3 = v[3] - __uint_as_float(hi[1] & 0xFFFF0000u);
  lo[0] = pack_bf16(r0, r1);
  lo[1] = pack_bf16(r2, r3);
}

__device__ __forceinline__ float karras_c_in(float sigma, float sd) { return 1.0f / sqrtf(sigma * sigma + sd * sd); }

// row of W that feeds tile row r of n-tile nt (GEGLU tiles interleave 32 value rows with their 32 gate rows)
__device__ __host__ __forceinline__ int w_row_of(int nt, int r, int N, bool geglu) {
  if (geglu) {
    const int n = nt * 64 + (r >> 6) * 32 + (r & 31);
    return (n < N) ? (((r >> 5) & 1) ? N + n : n) : -1;
  }
  return (nt * BN + r < N) ? nt * BN + r : -1;
}

// KS = 2 ("K split inside the workgroup", split3 + plain A + no norm only): 512 threads, two 256-thread groups that each
// run the whole pipeline below on their own half of K with their own pair of LDS stages; group 1 then hands its
// accumulators to group 0 through LDS (fixed order: deterministic), which runs the epilogue.  For shapes whose tile count
// does not even fill the chip once (level-2 out / down projections: 256 tiles, 16-48 K-steps each) this doubles the waves
// per SIMD and halves the serial K walk, where splitting K over workgroups would need atomics or a second pass.
template <int AMODE, bool NORM, int EPI, int PREC, int KS = 1>
__global__ __launch_bounds__(256 * KS, KS == 1 ? 2 : 1) void gemm_kernel(const GemmP p) {
  constexpr bool SPLIT = PREC == KD_PREC_SPLIT3;
  static_assert(KS == 1 || (SPLIT && !NORM && AMODE == KD_A_PLAIN), "K split: split3, plain A, no norm prologue");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const auto warm = code_warm_begin<32768>((int)blockIdx.x < p.warm && blockIdx.y == 0 && threadIdx.x < 64);   // kd_common.h: this kernel's code -> L2
  const int grp = KS == 1 ? 0 : (int)(threadIdx.x >> 8);
  float* As = smem;                       // exact mode
  float* Bs = smem + 2 * BM * S;
  char* stage0 = reinterpret_cast<char*>(smem) + grp * (2 * STAGE);   // split mode: this group's two stages
  float* rs = SPLIT ? reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + KS * 2 * STAGE) : Bs + 2 * BN * S;

  // 4 wave-private [8][64] epilogue strips: split mode has room for its own region; exact mode reuses the first
  // A stage (every wave is past the K loop's last barrier when the epilogue starts)
  float* ebuf = SPLIT ? rs + BM : smem;
  float* sc_tab = rs + BM + (SPLIT ? 4 * 8 * 64 : 0);      // [K] norm scales of this tile's sample (when p.scale_tab)
  const bool use_tab = NORM && p.scale_tab;

  const int tid = threadIdx.x & 255, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  constexpr int NCOL = (EPI == KD_EPI_GEGLU) ? 64 : BN;   // output columns covered per tile
  const int n_tiles = (p.N + NCOL - 1) / NCOL;
  // XCD-aware tile order: the dispatcher deals consecutive workgroups round-robin over the 8 XCDs (private
  // L2s), so give every XCD one CONTIGUOUS chunk of the (m-tile, n-tile) space with n fastest: the n-tiles
  // that re-read one A row-panel then run back to back on ONE L2 instead of missing in 8 of them.  The remap
  // is a bijection for any grid size (performance only -- correctness never depends on placement).
  int tile;
  {
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int nt = tile % n_tiles, mt = tile / n_tiles;
  const int m0 = mt * BM, n0 = nt * NCOL;
  const int M = p.M, N = p.N, K = p.K;
  const int nk_all = (K + BK - 1) / BK;
  const int nk = nk_all / KS, kt0 = grp * nk;             // this group's K-steps: kt0 .. kt0 + nk - 1 (host: nk_all % KS == 0)

  // ---- per-thread load coordinates: tile rows r0 + 32*j, j < NLD, k-chunk kc (same every K-step) --------
  // Loads are branch-free (out-of-range rows / k are clamped to a valid address and zeroed when the registers are
  // consumed): a load inside a divergent branch makes hipcc wait for it right there, which serialises the
  // memory latency of every row chunk and defeats the prefetch.
  const int kc = (tid % RT) * 4, r0 = tid / RT;
  bool row_ok[NLD];
  int a_b[NLD];          // sample index of the (clamped) row: norm scale / gather modes
  float a_cin[NLD];
  const float* a_ptr[NLD];   // plain mode: &A[row][kc]
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    const int gm = m0 + r0 + j * (256 / RT);
    row_ok[j] = gm < M;
    const int gmc = row_ok[j] ? gm : M - 1;
    a_b[j] = (AMODE == KD_A_PLAIN) ? gmc / p.rows_per_sample : gmc / (p.gh * p.gw);
    a_cin[j] = (AMODE == KD_A_PATCH_NCHW && p.sigma) ? karras_c_in(p.sigma[a_b[j]], p.sigma_data) : 1.0f;
    a_ptr[j] = p.A + (AMODE == KD_A_PLAIN ? (long)gmc * K + kc : 0);
  }
  int b_row_g[NLD];      // exact mode: global W row, -1 if out of range
#pragma unroll
  for (int j = 0; j < NLD; ++j) b_row_g[j] = SPLIT ? 0 : w_row_of(nt, r0 + j * (256 / RT), N, EPI == KD_EPI_GEGLU);

  // two register sets per operand: the K loop runs a distance-2 software pipeline (tile kt+2 is requested
  // before tile kt is consumed, so every load has a full iteration plus the neighbour workgroup's work to land)
  f32x4 ra0[NLD], ra1[NLD];
  f32x4 rb0[NLD], rb1[NLD];     // exact: fp32 W chunk; split3: one 16-byte chunk of the packed bf16 image
  float ssq[NLD];
#pragma unroll
  for (int j = 0; j < NLD; ++j) ssq[j] = 0.f;

  auto load_a = [&](int k0, f32x4 (&ra)[NLD]) {
    const int gk = min(k0 + kc, K - 4);          // clamped; the K tail is zeroed in store_tiles
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      if (AMODE == KD_A_PLAIN) {
        ra[j] = *reinterpret_cast<const f32x4*>(a_ptr[j] + (gk - kc));
      } else {
        const int gmc = min(m0 + r0 + j * (256 / RT), M - 1);
        const int rr = gmc - a_b[j] * (p.gh * p.gw), a_h = rr / p.gw, a_w = rr - a_h * p.gw;
        if (AMODE == KD_A_MERGE2x2) {
          const int Cin = K >> 2;
          const int q = gk / Cin, e = gk - q * Cin;
          const long src = (((long)a_b[j] * (2 * p.gh) + 2 * a_h + (q >> 1)) * (2 * p.gw) + 2 * a_w + (q & 1)) * Cin + e;
          ra[j] = *reinterpret_cast<const f32x4*>(p.A + src);
        } else {  // NCHW patch gather: k = (nh*pw + nw)*chan + c
          const int Himg = p.gh * p.ph, Wimg = p.gw * p.pw;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int k = gk + u;
            const int c = k % p.chan, q = k / p.chan;
            const int nh = q / p.pw, nw = q - nh * p.pw;
            const long src = (((long)a_b[j] * p.chan + c) * Himg + a_h * p.ph + nh) * Wimg + a_w * p.pw + nw;
            ra[j][u] = p.A[src];
          }
        }
      }
    }
  };
  // W tile: exact mode gathers fp32 rows; split mode copies the packed, pre-swizzled bf16 image verbatim
  const char* wp_tile = SPLIT ? reinterpret_cast<const char*>(p.Wp) + (size_t)nt * nk_all * WP_BLOCK : nullptr;
  auto load_b = [&](int kt, f32x4 (&rb)[NLD]) {
    if (SPLIT) {
      const f32x4* src = reinterpret_cast<const f32x4*>(wp_tile + (size_t)kt * WP_BLOCK) + tid;
#pragma unroll
      for (int j = 0; j < NLD; ++j) rb[j] = src[j * 256];      // linear copy of the 16 KiB image: 1 KiB per wave-instruction
    } else {
      const int gk = kt * BK + kc;
#pragma unroll
      for (int j = 0; j < NLD; ++j) {
        f32x4 w = {0.f, 0.f, 0.f, 0.f};
        if (b_row_g[j] >= 0 && gk < K) w = *reinterpret_cast<const f32x4*>(p.W + (long)b_row_g[j] * K + gk);
        rb[j] = w;
      }
    }
  };
  // registers -> LDS, when the loads have landed: zero the out-of-range rows / K tail, apply c_in and the norm
  // prologue (row sum of squares on the raw value, then the per-(sample, k) scale), split, store
  auto store_tiles = [&](int buf, int k0, const f32x4 (&ra)[NLD], const f32x4 (&rb)[NLD]) {
    const bool k_ok = k0 + kc < K;
    const int gk = min(k0 + kc, K - 4);
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int row = r0 + j * (256 / RT);
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      f32x4 v = (row_ok[j] && k_ok) ? ra[j] : zero;
      if (AMODE == KD_A_PATCH_NCHW) v = v * a_cin[j];
      if (NORM) {
        ssq[j] += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        const f32x4 sc = use_tab ? *reinterpret_cast<const f32x4*>(sc_tab + gk)
                                 : *reinterpret_cast<const f32x4*>(p.scale + (long)a_b[j] * p.scale_stride + gk);
        v = v * sc;
      }
      if (SPLIT) {
        u32x2 hi, lo;
        split4(v, hi, lo);
        char* img = stage0 + buf * STAGE + swz(row, kc >> 3) + (kc & 7) * 2;
        *reinterpret_cast<u32x2*>(img) = hi;
        *reinterpret_cast<u32x2*>(img + IMG) = lo;
        *reinterpret_cast<f32x4*>(stage0 + buf * STAGE + 2 * IMG + (tid + j * 256) * 16) = rb[j];
      } else {
        *reinterpret_cast<f32x4*>(As + (buf * BM + row) * S + kc) = v;
        *reinterpret_cast<f32x4*>(Bs + (buf * BN + row) * S + kc) = rb[j];
      }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (use_tab) {   // every row of the tile belongs to one sample (host guarantees rows_per_sample % BM == 0 or a shared gain)
    const float* src = p.scale + (long)(m0 / p.rows_per_sample) * p.scale_stride;
    for (int k = tid * 4; k < K; k += 1024) *reinterpret_cast<f32x4*>(sc_tab + k) = *reinterpret_cast<const f32x4*>(src + k);
    __syncthreads();
  }
  code_warm_end(warm);

  const int frag_off = (lane & 31) * S + 4 * (lane >> 5);
  const int l31 = lane & 31, lh = lane >> 5;
  auto compute = [&](int buf) {
    if (p.debug & 2) return;
    if (SPLIT) {
      const char* st = stage0 + buf * STAGE;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int o = swz(l31, 2 * s + lh);               // (row & 31) decides the swizzle: tile rows are 32-aligned
        bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int ao = (wr * 64 + i * 32) * 64 + o, bo = (wc * 64 + i * 32) * 64 + o;
          ah[i] = *reinterpret_cast<const bf16x8*>(st + ao);
          al[i] = *reinterpret_cast<const bf16x8*>(st + IMG + ao);
          bh[i] = *reinterpret_cast<const bf16x8*>(st + 2 * IMG + bo);
          bl[i] = *reinterpret_cast<const bf16x8*>(st + 3 * IMG + bo);
        }
        // term-major order: consecutive MFMAs hit DIFFERENT accumulators (a back-to-back chain on one accumulator
        // stalls on its own result; with four in rotation each is revisited after three other issues)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
      }
    } else {
      const float* a_base = As + (buf * BM + wr * 64) * S + frag_off;
      const float* b_base = Bs + (buf * BN + wc * 64) * S + frag_off;
#pragma unroll
      for (int kk = 0; kk < BK; kk += 8) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(a_base + kk);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(a_base + 32 * S + kk);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(b_base + kk);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(b_base + 32 * S + kk);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b0[s], acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b1[s], acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b0[s], acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b1[s], acc[1][1], 0, 0, 0);
        }
      }
    }
  };

  // distance-2 pipeline, unrolled by two so that both register sets have static names
  load_a(kt0 * BK, ra0); load_b(kt0, rb0);
  if (nk > 1) { load_a((kt0 + 1) * BK, ra1); load_b(kt0 + 1, rb1); }
  store_tiles(0, kt0 * BK, ra0, rb0);
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    // tile kt is in LDS buffer 0, set 1 holds tile kt+1 (in flight), set 0 is free
    if (kt + 2 < nk) { load_a((kt0 + kt + 2) * BK, ra0); load_b(kt0 + kt + 2, rb0); }
    compute(0);
    if (kt + 1 < nk) store_tiles(1, (kt0 + kt + 1) * BK, ra1, rb1);
    __syncthreads();
    if (kt + 1 >= nk) break;
    // tile kt+1 is in LDS buffer 1, set 0 holds tile kt+2 (in flight), set 1 is free
    if (kt + 3 < nk) { load_a((kt0 + kt + 3) * BK, ra1); load_b(kt0 + kt + 3, rb1); }
    compute(1);
    if (kt + 2 < nk) store_tiles(0, (kt0 + kt + 2) * BK, ra0, rb0);
    __syncthreads();
  }

  if (KS == 2) {
    // group 1 -> group 0: 64 accumulator floats per thread through group 1's (now idle) stages, 16 bytes per lane per
    // access at (j * 256 + tid): conflict-free, and the same tid holds the same fragment positions in both groups
    f32x4* xch = reinterpret_cast<f32x4*>(reinterpret_cast<char*>(smem) + 2 * STAGE);
    if (grp == 1) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            xch[((i * 2 + j) * 4 + q) * 256 + tid] = f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
    }
    __syncthreads();
    if (grp == 1) return;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 o = xch[((i * 2 + j) * 4 + q) * 256 + tid];
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[i][j][4 * q + u] += o[u];
        }
  }

  if (NORM) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const float s = wave_sum_xor(ssq[j], RT);
      if ((tid % RT) == 0) rs[r0 + j * (256 / RT)] = rsqrtf(s / (float)K + p.eps);
    }
    __syncthreads();
  }

  // ---- epilogue -------------------------------------------------------------------------------
  if (p.debug & 1) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }
  // Vector path: every wave transposes its 64x64 accumulator block through a private 8-row LDS strip so that a
  // lane owns 4 CONSECUTIVE output columns: residual / skip reads and C writes are 16 bytes per lane in 256-byte
  // row segments (a quarter of the instructions of the MFMA-layout dword stores, and full cache lines).  Loads are
  // branch-free (clamped address), only the stores are guarded.  Strips are wave-local: LDS executes one wave's
  // instructions in order, so no workgroup barrier is needed between the strip writes and reads.
  const int col_l = lane & 31;
  constexpr int NW = (EPI == KD_EPI_GEGLU) ? 32 : 64;          // strip width = output columns owned by this wave
  const int wn0 = n0 + wc * NW;                                 // first output column of this wave
  float* strip = ebuf + wid * (8 * 64);
  const bool vec_ok = (N & 3) == 0 &&
                      (EPI != KD_EPI_SPLIT_LERP || (N & 15) == 0) &&
                      (EPI != KD_EPI_UNPATCH_NCHW || (p.pw == 4 && N <= 64 && (p.gw & 7) == 0));
  const int qk_vec = wn0 >> 6, qk_which = EPI == KD_EPI_QKV ? qk_vec / p.n_heads : 2, qk_head = EPI == KD_EPI_QKV ? qk_vec - qk_which * p.n_heads : 0;
  const float qk_sqrt = (EPI == KD_EPI_QKV && qk_which < 2) ? sqrtf(p.qk_scale[qk_head]) : 1.0f;
  if (vec_ok) {
    float c_out[2] = {1.f, 1.f}, c_skip[2] = {0.f, 0.f};        // unpatch: Karras scalings of this lane's two samples
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float rsv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) rsv[r] = NORM ? rs[wr * 64 + i * 32 + mfma32_row(r, lane)] : 1.0f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // ---- accumulators of rows 8g..8g+7 -> strip[8][NW] ----------------------------------------
        if (EPI == KD_EPI_GEGLU) {
#pragma unroll
          for (int q = 0; q < 4; q += 2) {                // rows r, r+1: adjacent accumulator registers -> packed fp32 math
            const int r = 4 * g + q, row8 = q + 4 * (lane >> 5);
            const f32x2 rs2 = {rsv[r], rsv[r + 1]};
            const f32x2 gate = f32x2{acc[i][1][r], acc[i][1][r + 1]} * rs2;
            const f32x2 val = f32x2{acc[i][0][r], acc[i][0][r + 1]} * rs2;
            const f32x2 o = (p.debug & 8) ? val * gate : geglu_pair(val * 0.5f, gate);
            strip[row8 * NW + col_l] = o.x;
            strip[(row8 + 1) * NW + col_l] = o.y;
          }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int r = 4 * g + q, row8 = q + 4 * (lane >> 5);
            strip[row8 * NW + col_l] = acc[i][0][r] * rsv[r];
            strip[row8 * NW + 32 + col_l] = acc[i][1][r] * rsv[r];
          }
        }
        const int row_base = m0 + wr * 64 + i * 32 + 8 * g;
        if (EPI == KD_EPI_UNPATCH_NCHW) {
          // strip row = token (8 consecutive tokens of one image row), column = (nh*4 + nw)*chan + c.  One item =
          // (token, c, nh): its 4 nw values are 16 contiguous bytes of the NCHW image, consecutive tokens continue them.
          if (wn0 < N) {
            const int items = p.chan * p.ph * 8, hw = p.gh * p.gw;
            for (int it = lane; it < items; it += 64) {
              const int tok = it & 7, cb = it >> 3, c = cb % p.chan, nh = cb / p.chan;
              const int gm = row_base + tok, gmc = min(gm, M - 1);
              const int b = gmc / hw, rr = gmc - b * hw, h = rr / p.gw, w = rr - h * p.gw;
              f32x4 v;
#pragma unroll
              for (int nw = 0; nw < 4; ++nw) v[nw] = strip[tok * NW + (nh * 4 + nw) * p.chan + c];
              const long o = (((long)b * p.chan + c) * (p.gh * p.ph) + h * p.ph + nh) * (p.gw * 4) + w * 4;
              if (p.sigma) {
                const float sg = p.sigma[b], sd = p.sigma_data, var = sg * sg + sd * sd;
                const float co = sg * sd / sqrtf(var), cs = sd * sd / var;
                const f32x4 xin = *reinterpret_cast<const f32x4*>(p.R + o);
                v = v * co + xin * cs;
              }
              if (gm < M) *reinterpret_cast<f32x4*>(p.C + o) = v;
            }
          }
          (void)c_out; (void)c_skip;
          continue;
        }
        // ---- strip -> global: lane owns (row8, 4 columns) ---------------------------------------------
        constexpr int PER_LANE = NW / 32;                       // float4 items per lane (8 rows * NW/4 items / 64 lanes)
        f32x4 v[PER_LANE], rv[PER_LANE], qcs[PER_LANE], qsn[PER_LANE];
        long off[PER_LANE];
        bool ok[PER_LANE];
#pragma unroll
        for (int t = 0; t < PER_LANE; ++t) {
          const int idx = lane + 64 * t, row8 = idx / (NW / 4), c4 = (idx % (NW / 4)) * 4;
          const int gm = row_base + row8, gn = wn0 + c4;
          ok[t] = gm < M && gn < N;
          const int gmc = min(gm, M - 1), gnc = min(gn, N - 4);
          v[t] = *reinterpret_cast<const f32x4*>(strip + row8 * NW + c4);
          if (EPI == KD_EPI_SPLIT_LERP) {
            const int hw = p.gh * p.gw, Cout = N >> 2;
            const int b = gmc / hw, rr = gmc - b * hw, h = rr / p.gw, w = rr - h * p.gw;
            const int q = gnc / Cout, e = gnc - q * Cout;
            off[t] = (((long)b * (2 * p.gh) + 2 * h + (q >> 1)) * (2 * p.gw) + 2 * w + (q & 1)) * Cout + e;
          } else {
            off[t] = (long)gmc * N + gnc;
          }
          if (EPI == KD_EPI_RESIDUAL || EPI == KD_EPI_SPLIT_LERP) rv[t] = *reinterpret_cast<const f32x4*>(p.R + off[t]);
          if (EPI == KD_EPI_QKV && qk_which < 2) {
            // this wave's 64 columns are ONE (q|k|v, head) vector of the row; 16 lanes hold it (4 dims each).
            // RoPE chunks requested branch-free for both items before either is used.
            const long tr = ((long)(gmc % p.rows_per_sample) * p.n_heads + qk_head) * KD_ROT + 4 * (lane & 3);
            qcs[t] = *reinterpret_cast<const f32x4*>(p.rope_cos + tr);
            qsn[t] = *reinterpret_cast<const f32x4*>(p.rope_sin + tr);
          }
        }
        if (EPI == KD_EPI_QKV && qk_which < 2) {
#pragma unroll
          for (int t = 0; t < PER_LANE; ++t) v[t] = prep_row16_regs(v[t], lane & 15, qk_sqrt, qcs[t], qsn[t], p.eps);
        }
#pragma unroll
        for (int t = 0; t < PER_LANE; ++t) {
          f32x4 o = v[t];
          if (EPI == KD_EPI_STORE) {
            o = o + p.out_add;
          } else if (EPI == KD_EPI_RESIDUAL) {
            o = o + rv[t];
          } else if (EPI == KD_EPI_SPLIT_LERP) {
            const float fac = *p.fac;
#pragma unroll
            for (int u = 0; u < 4; ++u) {                        // torch.lerp(skip, x, fac), ATen's two-branch form
              const float skip = rv[t][u], diff = o[u] - skip;
              o[u] = (fabsf(fac) < 0.5f) ? skip + fac * diff : o[u] - diff * (1.0f - fac);
            }
          }
          if (EPI == KD_EPI_QKV && SPLIT && p.qkv_packed) {      // operand format of the split attention cores (kdiff_hip.h)
            u32x2 hi, lo;
            split4(o, hi, lo);
            o = f32x4{__uint_as_float(hi[0]), __uint_as_float(hi[1]), __uint_as_float(lo[0]), __uint_as_float(lo[1])};
          }
          if (ok[t]) *reinterpret_cast<f32x4*>(p.C + off[t]) = o;
        }
      }
    }
    return;
  }

  // Scalar path (N % 4 != 0, patch widths other than 4, ...): MFMA-layout dword accesses with per-element guards.
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row_t = wr * 64 + i * 32 + mfma32_row(r, lane);
      const int gm = m0 + row_t;
      if (gm >= M) continue;
      const float rscale = NORM ? rs[row_t] : 1.0f;
      if (EPI == KD_EPI_GEGLU) {
        const int gn = n0 + wc * 32 + col_l;
        if (gn < N) p.C[(long)gm * N + gn] = (acc[i][0][r] * rscale) * gelu_erf_fast(acc[i][1][r] * rscale);
        continue;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int gn = n0 + wc * 64 + j * 32 + col_l;
        if (gn >= N) continue;
        const float v = acc[i][j][r] * rscale;
        if (EPI == KD_EPI_STORE) {
          p.C[(long)gm * N + gn] = v + p.out_add;
        } else if (EPI == KD_EPI_RESIDUAL) {
          const long o = (long)gm * N + gn;
          p.C[o] = v + p.R[o];
        } else if (EPI == KD_EPI_SPLIT_LERP) {
          const int hw = p.gh * p.gw, Cout = N >> 2;
          const int b = gm / hw, rr = gm % hw, h = rr / p.gw, w = rr % p.gw;
          const int q = gn / Cout, e = gn - q * Cout;
          const long o = (((long)b * (2 * p.gh) + 2 * h + (q >> 1)) * (2 * p.gw) + 2 * w + (q & 1)) * Cout + e;
          const float skip = p.R[o], fac = *p.fac;
          const float diff = v - skip;                       // torch.lerp(skip, x, fac), ATen's two-branch form
          p.C[o] = (fabsf(fac) < 0.5f) ? skip + fac * diff : v - diff * (1.0f - fac);
        } else if (EPI == KD_EPI_UNPATCH_NCHW) {
          const int hw = p.gh * p.gw;
          const int b = gm / hw, rr = gm % hw, h = rr / p.gw, w = rr % p.gw;
          const int c = gn % p.chan, q = gn / p.chan, nh = q / p.pw, nw = q - nh * p.pw;
          const long o = (((long)b * p.chan + c) * (p.gh * p.ph) + h * p.ph + nh) * (p.gw * p.pw) + w * p.pw + nw;
          if (p.sigma) {
            const float sg = p.sigma[b], sd = p.sigma_data;
            const float var = sg * sg + sd * sd;
            const float c_skip = sd * sd / var, c_out = sg * sd / sqrtf(var);
            p.C[o] = v * c_out + p.R[o] * c_skip;
          } else {
            p.C[o] = v;
          }
        }
      }
    }
  }
}

// ---- one-off weight packing for the split mode ---------------------------------------------------
// out block (nt, ks): [hi image][lo image], each the swizzled [128][32] bf16 LDS image of W rows w_row_of(nt, r)
// and k = ks*32 .. +32 (zero where the row or k is out of range).  One thread per (block, row, 16-byte chunk).
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ W, char* __restrict__ out, int N, int K, int geglu,
                                                           int n_tiles, int nk) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)n_tiles * nk * BN * 4;
  if (idx >= total) return;
  const int c = idx & 3, r = (idx >> 2) & (BN - 1);
  const long blk = idx >> 9;
  const int ks = blk % nk, nt = blk / nk;
  const int wrow = w_row_of(nt, r, N, (geglu & 1) != 0);
  float v[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    // layout bit 2 (down projection of the fused FF block, ffn_x3.hip; with bit 1 -- layout 3 -- its up projection behind a fused out
    // projection): inside every group of 16 k the order in which an MFMA's accumulator registers hold a row's features: 0-3, 8-11, 4-7,
    // 12-15 (chunk c = 2 g + h of the stage: k 16 g + 4 h + {0..3, 8..11}), so that a C-layout result IS the next product's B operand
    const int k = ks * BK + ((geglu & 2) ? 16 * (c >> 1) + 4 * (c & 1) + (u & 3) + 8 * (u >> 2) : c * 8 + u);
    v[u] = (wrow >= 0 && k < K) ? W[(long)wrow * K + k] : 0.f;
  }
  u32x2 h0, l0, h1, l1;
  split4(f32x4{v[0], v[1], v[2], v[3]}, h0, l0);
  split4(f32x4{v[4], v[5], v[6], v[7]}, h1, l1);
  char* dst = out + blk * WP_BLOCK + swz(r, c);
  *reinterpret_cast<uint4*>(dst) = uint4{h0[0], h0[1], h1[0], h1[1]};
  *reinterpret_cast<uint4*>(dst + IMG) = uint4{l0[0], l0[1], l1[0], l1[1]};
}

template <int AMODE, bool NORM, int EPI, int PREC, int KS = 1>
static int launch(const GemmP& d, hipStream_t s) {
  constexpr size_t LDS_BASE = PREC == KD_PREC_SPLIT3 ? LDS_SPLIT + (KS - 1) * 2 * STAGE : LDS_EXACT;
  constexpr size_t LDS_BYTES = LDS_BASE + (NORM ? SCALE_TAB_MAX_K * sizeof(float) : 0);
  constexpr int NCOL = (EPI == KD_EPI_GEGLU) ? 64 : BN;
  const long tiles = (long)((d.M + BM - 1) / BM) * ((d.N + NCOL - 1) / NCOL);
  auto kern = gemm_kernel<AMODE, NORM, EPI, PREC, KS>;
  static LdsAttr attr_set;
  attr_set.ensure(reinterpret_cast<const void*>(kern), (int)LDS_BYTES);
  const double n_eff = (EPI == KD_EPI_GEGLU) ? 2.0 * d.N : (double)d.N;
  char nm[96] = "gemm";
  if (prof_on()) snprintf(nm, sizeof(nm), "gemm_%s<a%d,n%d,e%d%s> M=%d N=%d K=%d", PREC == KD_PREC_SPLIT3 ? "bf16x3" : "f32", AMODE, (int)NORM, EPI, KS == 2 ? ",ks2" : "", d.M, d.N, d.K);
  // algorithmic bytes: A once, W once, C once (+ the residual / skip / x_in operand of the epilogues that read one)
  const double r_bytes = (EPI == KD_EPI_RESIDUAL || EPI == KD_EPI_SPLIT_LERP || (EPI == KD_EPI_UNPATCH_NCHW && d.sigma)) ? 4.0 * d.M * d.N : 0.0;
  LaunchScope prof(nm, 2.0 * d.M * n_eff * d.K, 4.0 * ((double)d.M * d.K + n_eff * d.K + (double)d.M * d.N) + r_bytes, s);
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256 * KS), LDS_BYTES, s, d);
  return check_launch("kd_gemm_f32");
}

}  // namespace kd

namespace kd { int gemm_astat_try(const GemmP& d, hipStream_t s, int* rc); }    // gemm_astat.hip
namespace kd { int gemm_x3_try(const GemmP& d, hipStream_t s, int* rc); }       // gemm_x3.hip
namespace kd { int gemm_x3t_try(const GemmP& d, hipStream_t s, int* rc); }      // gemm_x3t.hip
namespace kd { int gemm_skinny_try(const GemmP& d, hipStream_t s, int* rc); }   // gemm_skinny.hip
namespace kd { int gemm_x3r_try(const GemmP& d, hipStream_t s, int* rc); }      // gemm_x3r.hip
namespace kd { int gemm_x3s_try(const GemmP& d, hipStream_t s, int* rc); }      // gemm_x3s.hip

using namespace kd;

extern "C" int kd_gemm_f32(const KdGemm* dp, void* stream) {
  if (!dp) return fail(KD_EINVAL, "kd_gemm_f32: null descriptor");
  const KdGemm& d = *dp;
  hipStream_t s = (hipStream_t)stream;
  if (d.M <= 0 || d.N <= 0 || d.K <= 0 || (d.K & 3)) return fail(KD_EINVAL, "kd_gemm_f32: bad M/N/K %d/%d/%d (K %% 4 != 0?)", d.M, d.N, d.K);
  if (!d.A || !d.C || (!d.W && d.precision == KD_PREC_EXACT)) return fail(KD_EINVAL, "kd_gemm_f32: null A/W/C");
  if (d.norm && (!d.scale || d.rows_per_sample <= 0 || (d.scale_stride & 3))) return fail(KD_EINVAL, "kd_gemm_f32: norm needs scale, rows_per_sample, scale_stride%%4==0");
  if (d.a_mode != KD_A_PLAIN && (d.gh <= 0 || d.gw <= 0 || d.M % (d.gh * d.gw))) return fail(KD_EINVAL, "kd_gemm_f32: gather mode needs gh, gw with M %% (gh*gw) == 0");
  if (d.a_mode == KD_A_MERGE2x2 && (d.K & 15)) return fail(KD_EINVAL, "kd_gemm_f32: merge needs K %% 16 == 0");
  if (d.a_mode == KD_A_PATCH_NCHW && (d.ph <= 0 || d.pw <= 0 || d.chan <= 0 || d.K != d.ph * d.pw * d.chan)) return fail(KD_EINVAL, "kd_gemm_f32: patch gather needs K == ph*pw*chan");
  if (d.epi == KD_EPI_RESIDUAL && !d.R) return fail(KD_EINVAL, "kd_gemm_f32: residual needs R");
  if (d.epi == KD_EPI_SPLIT_LERP && (!d.R || !d.fac || (d.N & 3) || d.gh <= 0 || d.gw <= 0 || d.M % (d.gh * d.gw))) return fail(KD_EINVAL, "kd_gemm_f32: split needs R, fac, N%%4==0, gh, gw");
  if (d.epi == KD_EPI_UNPATCH_NCHW && (d.ph <= 0 || d.pw <= 0 || d.chan <= 0 || d.N != d.ph * d.pw * d.chan || d.gh <= 0 || d.gw <= 0 || d.M % (d.gh * d.gw) || (d.sigma && !d.R)))
    return fail(KD_EINVAL, "kd_gemm_f32: unpatch needs N == ph*pw*chan, gh, gw (and R when sigma is given)");
  if (d.norm && d.rows_per_sample <= 0) return fail(KD_EINVAL, "kd_gemm_f32: rows_per_sample");
  if (d.epi == KD_EPI_QKV && (d.n_heads <= 0 || d.N != 3 * d.n_heads * 64 || d.rows_per_sample <= 0 || !d.qk_scale || !d.rope_cos || !d.rope_sin))
    return fail(KD_EINVAL, "kd_gemm_f32: qkv epilogue needs N == 3*n_heads*64, rows_per_sample, qk_scale, rope_cos, rope_sin");
  if (d.qkv_packed && (d.epi != KD_EPI_QKV || d.precision != KD_PREC_SPLIT3))
    return fail(KD_EINVAL, "kd_gemm_f32: qkv_packed needs the qkv epilogue and split3 precision");
  if ((d.a_split || d.c_split) && d.precision != KD_PREC_SPLIT3) return fail(KD_EINVAL, "kd_gemm_f32: a_split / c_split are KD_PREC_SPLIT3 operand formats");
  if (d.a_split && (d.norm || d.a_mode != KD_A_PLAIN || !d.A_lo)) return fail(KD_EINVAL, "kd_gemm_f32: a_split needs A_lo, a_mode KD_A_PLAIN and no norm prologue (kd_norm_split_f32 does the norm)");
  if (d.c_split && !d.C_lo) return fail(KD_EINVAL, "kd_gemm_f32: c_split needs C_lo");
  GemmP e;
  static_cast<KdGemm&>(e) = d;
  e.debug = option("gemm_debug", 0);
  e.warm = option("code_warm", KD_CODE_WARM_DEFAULT);
  if (e.rows_per_sample <= 0) e.rows_per_sample = e.M;
  // all rows of a 128-row tile share their scale vector: stage it in LDS once per tile
  e.scale_tab = e.norm && e.a_mode == KD_A_PLAIN && e.K <= SCALE_TAB_MAX_K && (e.scale_stride == 0 || e.rows_per_sample % BM == 0);

  if (e.a_split) {
    if (e.rows_per_sample <= 0) e.rows_per_sample = e.M;
    int rc = 0;
    if (!gemm_x3t_try(e, s, &rc)) return rc;
    return fail(KD_EINVAL, "kd_gemm_f32: no kernel for pre-split A with epi=%d M=%d N=%d K=%d (K %% 32, N %% 128 (GEGLU: 64))", d.epi, d.M, d.N, d.K);
  }
  {
    const bool skinny_on = option("skinny", 1) != 0;
    int rc = 0;
    if (skinny_on && !gemm_skinny_try(e, s, &rc)) return rc;   // <= 128 rows (one per sample): the conditioning chain
    if (!gemm_x3s_try(e, s, &rc)) return rc;                   // few rows (small batches), split3: the latency form, gemm_x3s.hip
  }
  {
    const bool astat_on = option("astat", 1) != 0;
    int rc = 0;
    if (astat_on && !gemm_x3_try(e, s, &rc)) return rc;        // norm -> wide projection, split3: round-3 A-stationary kernel (gemm_x3.hip)
    if (e.c_split) return fail(KD_EINVAL, "kd_gemm_f32: c_split is produced by the GEGLU projections of gemm_x3.hip (K = 128 / 256, norm) and gemm_x3t.hip (a_split) only");
    if (astat_on && !gemm_astat_try(e, s, &rc)) return rc;     // ... and its round-1 predecessor (RoPE tables instead of positions; A/B runs)
    if (!gemm_x3r_try(e, s, &rc)) return rc;                   // no norm in front (residual projections, merges), split3: gemm_x3r.hip
  }
  if (e.precision != KD_PREC_EXACT && e.precision != KD_PREC_SPLIT3) return fail(KD_EINVAL, "kd_gemm_f32: unknown precision %d", e.precision);
  if (e.precision == KD_PREC_SPLIT3 && !e.Wp) return fail(KD_EINVAL, "kd_gemm_f32: split3 needs the packed weight image Wp (kd_pack_weight_bf16x3)");
  {
    // residual / plain projections whose tiles do not even fill the chip once and that walk a long K: split K inside the
    // workgroup (two wave groups, deterministic in-LDS reduction)
    const bool ks_on = option("ksplit", 1) != 0;
    const long tiles = (long)((e.M + BM - 1) / BM) * ((e.N + BN - 1) / BN);
    const int nk = (e.K + BK - 1) / BK;
    if (ks_on && e.precision == KD_PREC_SPLIT3 && e.Wp && e.a_mode == KD_A_PLAIN && !e.norm && !e.debug && tiles <= 256 && nk >= 8 && nk % 4 == 0) {
      if (e.epi == KD_EPI_RESIDUAL) return launch<KD_A_PLAIN, false, KD_EPI_RESIDUAL, KD_PREC_SPLIT3, 2>(e, s);
      if (e.epi == KD_EPI_STORE) return launch<KD_A_PLAIN, false, KD_EPI_STORE, KD_PREC_SPLIT3, 2>(e, s);
    }
  }
#define KD_CASE(AM, NO, EP)                                             \
  if (e.a_mode == AM && (e.norm != 0) == NO && e.epi == EP)             \
    return e.precision == KD_PREC_SPLIT3 ? launch<AM, NO, EP, KD_PREC_SPLIT3>(e, s) : launch<AM, NO, EP, KD_PREC_EXACT>(e, s);
  KD_CASE(KD_A_PLAIN, true, KD_EPI_STORE)
  KD_CASE(KD_A_PLAIN, false, KD_EPI_STORE)
  KD_CASE(KD_A_PLAIN, false, KD_EPI_RESIDUAL)
  KD_CASE(KD_A_PLAIN, true, KD_EPI_QKV)
  KD_CASE(KD_A_PLAIN, false, KD_EPI_QKV)
  KD_CASE(KD_A_PLAIN, true, KD_EPI_GEGLU)
  KD_CASE(KD_A_PLAIN, false, KD_EPI_GEGLU)
  KD_CASE(KD_A_MERGE2x2, false, KD_EPI_STORE)
  KD_CASE(KD_A_PLAIN, false, KD_EPI_SPLIT_LERP)
  KD_CASE(KD_A_PATCH_NCHW, false, KD_EPI_STORE)
  KD_CASE(KD_A_PLAIN, true, KD_EPI_UNPATCH_NCHW)
  KD_CASE(KD_A_PLAIN, false, KD_EPI_UNPATCH_NCHW)
#undef KD_CASE
  return fail(KD_EINVAL, "kd_gemm_f32: unsupported combination a_mode=%d norm=%d epi=%d", d.a_mode, d.norm, d.epi);
}

extern "C" long long kd_packed_weight_bytes(int N, int K, int geglu) {
  if (N <= 0 || K <= 0) return 0;
  const long n_tiles = (N + ((geglu & 1) ? 64 : BN) - 1) / ((geglu & 1) ? 64 : BN), nk = (K + BK - 1) / BK;
  return n_tiles * nk * (long long)WP_BLOCK;
}

extern "C" int kd_pack_weight_bf16x3(const float* W, void* out, int N, int K, int geglu, void* stream) {
  if (!W || !out || N <= 0 || K <= 0 || geglu < 0 || geglu > 3) return fail(KD_EINVAL, "kd_pack_weight_bf16x3: bad arguments");
  const int n_tiles = (N + ((geglu & 1) ? 64 : BN) - 1) / ((geglu & 1) ? 64 : BN), nk = (K + BK - 1) / BK;
  const long total = (long)n_tiles * nk * BN * 4;
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W,
                     reinterpret_cast<char*>(out), N, K, geglu, n_tiles, nk);
  return check_launch("kd_pack_weight_bf16x3");
}

KD_TEXT_PAD(gemm)      // last function of this code object: kd_common.h, code warm-up
