// Attention cores of the fp32-parity ("split3") mode in the round-2 bf16 shape (attn_bf16.hip), gfx950.
//
//   neighbourhood attention, kernel sizes 3 .. 13 (natten na2d, image_transformer_v2.py:399-410,428; 7 is what the shipped configs use,
//   3 / 5 / 9 run the same kernel with other constants, 11 / 13 a densely packed patch), for qkv STORED SPLIT by the qkv projection
//   (KdGemm.qkv_packed, prep = 2 of kd_attn_na2d_f32): every 4 head dims are 16 bytes [hi: 4 x bf16][lo: 4 x bf16], q and k already
//   cosine-sim-scaled and rotated.  Every product is hi*hi + hi*lo + lo*hi on the bf16 MFMA with fp32 accumulation (per-product
//   error <= ~2^-15), scores / softmax / accumulators fp32, out fp32.
//
// What changed against attn_f32.hip's neighbourhood core (round 1: halo rows through registers, split / transposed by VALU, 16 ds_write
// per key pair, 82.8 us per level-0 launch = 3.2 TB/s of algorithmic traffic):
//   * halo rows go HBM / L2 -> LDS by global_load_lds as they are stored (256-byte rows, 4 rows per wave-instruction), the 16-byte chunk
//     index XOR-swizzled on the SOURCE side: no staging VALU, no ds_write pass;
//   * a K fragment (8 head dims, hi and lo) is two ds_read_b128 of adjacent chunks -- [hi4 | lo4][hi4 | lo4] regrouped in registers;
//   * V^T fragments come out of the row-major V image through ds_read_b64_tr_b16: an 8-byte piece of a row is exactly the hi (or lo)
//     quad of 4 head dims, which is what the transposing read moves;
//   * K and V share ONE 78 KiB image (two workgroups per CU): V is requested right behind the score MFMAs' barrier and lands behind the
//     mask / softmax / probability split;
//   * mask by v_min3 against per-lane column / row validity, exp2 softmax (attn_bf16.hip).
#include "x3_common.h"

namespace kd {
namespace x3 { extern unsigned long long* g_clk; }      // gemm_x3.hip (kd_prof_clock_buffer)
namespace x3a {

using b16::bf16x8;
using b16::u32x2;
using b16::u32x4;

constexpr int DH = 64;
constexpr int NA_TH = 8, NA_TW = 16;
constexpr int ROWB = 256;                                    // bytes of an image row: 64 head dims as 16 chunks [hi4 | lo4]
// geometry of kernel size KS (3 .. 9): a wave's 4 x 8 queries see a patch of (4 + KS - 1) rows x (8 + KS - 1) <= 16 columns
template <int KS>
struct NaGeo {
  static constexpr int HR = NA_TH + KS - 1, HC = NA_TW + KS - 1;      // key halo of an 8 x 16 query tile (KS = 7: 14 x 22)
  static constexpr int PR = 4 + KS - 1;                               // patch rows of a wave, 16 keys wide
  static constexpr int NKT = (PR * 16 + 31) / 32;                     // key tiles of 32 (KS = 7: 5)
  // image rows, 4 per LDS-DMA instruction: a patch may poke past the halo's last key -- the highest row a fragment read touches is
  // (HR - 1) HC + (HC - (8 + KS - 1)) + 15 = HR HC - KS + 8
  static constexpr int ROWS = ((HR * HC - KS + 9 + 3) / 4) * 4;
  static constexpr int LDS = ROWS * ROWB;                             // KS = 7: 79 872 B, two workgroups per CU (3, 5 too; 9: one)
  static_assert(8 + KS - 1 <= 16, "16-key patch rows");
};

struct NArgs {
  const float* qkv; float* out;
  int batch, H, W, nh;
  int warm;
  unsigned long long* clk;       // kd_prof_clock_buffer: per-workgroup entry / exit stamps (x3_common.h: wg_stamp_begin)
};

#define KD_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

__device__ __forceinline__ void glds16(const void* src, void* dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

// chunk c (0..15) of an image row sits at slot c ^ rsw(r), r = the row's index in the global core and in the 11 / 13 neighbourhood form, its
// halo COLUMN in the neighbourhood kernel (sizes 3 .. 9).  rsw swaps the two 2-bit fields of r & 15: 16 consecutive rows (columns) take 16
// different slots for one chunk (the ds_read_b128 of a K fragment: 16 lanes, 16 rows), and FOUR consecutive ones differ in slot bits
// 2-3, so the 4 rows x 4 chunks of a ds_read_b64_tr_b16 group land in 16 different slots.  rsw(r + 8) = rsw(r) ^ 2.
// (Keys past the halo's last column -- a patch may poke 1 - 2 keys over -- then read another chunk of a real row: finite values, always
// masked.)
__device__ __forceinline__ int rsw(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

using s16x4 = short __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x2 tr_read(const char* img, int addr) {
  return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(img + addr)));
}
// v_min3_f32 / v_max3_f32 through the compiler's own patterns (hazards of MFMA results are the compiler's to pad)
__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
__device__ __forceinline__ float min3f(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }

template <int KS>
__global__ __launch_bounds__(256, NaGeo<KS>::LDS <= 80 * 1024 ? 2 : 1) void attn_na2d_x3_kernel(const NArgs a) {
  using G = NaGeo<KS>;
  constexpr int HR = G::HR, HC = G::HC, PR = G::PR, NKT = G::NKT, ROWS = G::ROWS;
  extern __shared__ __attribute__((aligned(16))) char img[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, h2 = lane >> 5;
  const auto warm = code_warm_begin<10240>((int)blockIdx.x < a.warm && tid < 64);
  const x3::WgStamp wgs = x3::wg_stamp_begin(a.clk);
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
  const size_t row_bytes = (size_t)3 * a.nh * DH * 4;
  const char* base = reinterpret_cast<const char*>(a.qkv) + (size_t)b * T * row_bytes + head * (DH * 4);
  const int ty0 = ty * NA_TH, tx0 = tx * NA_TW;
  const int hy0 = max(0, min(ty0 - KS / 2, a.H - HR)), hx0 = max(0, min(tx0 - KS / 2, a.W - HC));

  // ---- halo rows -> image: image row 4 pc + (lane >> 4) = halo position (y, x), 16 rows per round of the 4 waves.  Rows past the halo's
  // last key and out-of-image positions (images smaller than the halo) take a real token: they lie outside every window.  The byte offsets of
  // this lane's pieces are computed ONCE (the K pass) and kept for the V pass: the same rows, `nh * 256` bytes further ------------------------------
  constexpr int NPC = (ROWS / 4 + 3) / 4;                // staging rounds of a wave
  unsigned soff[NPC];
  {
    int row = 4 * wid + (lane >> 4);
    int y = row / HC, x = row % HC;
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      const int ky = min(hy0 + y, a.H - 1), kx = min(hx0 + x, a.W - 1);
      soff[i] = (unsigned)((ky * a.W + kx) * (int)row_bytes + (((lane & 15) ^ rsw(x)) << 4));
      x += 16;
      if (x >= HC) { x -= HC; ++y; }
    }
  }
  auto stage = [&](int part_bytes) {
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      const int pc = wid + 4 * i;
      if (pc < ROWS / 4) glds16(base + (size_t)soff[i] + part_bytes, img + pc * 1024);
    }
  };
  stage(a.nh * DH * 4);                                  // K
  // ---- this lane's query: B fragments of the 4 k-steps (head dims 16 st + 8 h2 .. + 7: chunks 4 st + 2 h2, + 1) --------------------------
  const int qy_raw = ty0 + 4 * wy_ + (l31 >> 3), qx_raw = tx0 + 8 * wx_ + (l31 & 7);
  const bool q_ok = qy_raw < a.H && qx_raw < a.W;
  const int qy = min(qy_raw, a.H - 1), qx = min(qx_raw, a.W - 1);
  const int q_tok = qy * a.W + qx;
  // Q operands per k-step: a stored chunk is [hi4 | lo4] of 4 head dims.  A K chunk goes into the MFMA AS IT IS READ (k-slots hi0..3, lo0..3 of
  // its 4 dims) against qa = [qhi0..3, qhi0..3] of the same dims: K_hi Q_hi + K_lo Q_hi of those dims in one instruction, no regrouping of the
  // K registers; the third term K_hi Q_lo takes the hi quads of both chunks (the only regrouped fragment: 4 moves per tile and step instead
  // of 8) against ql = [qlo of chunk 0, qlo of chunk 1].
  bf16x8 qa[4][2], ql[4];
  {
    const u32x4* qp = reinterpret_cast<const u32x4*>(base + (size_t)q_tok * row_bytes + 32 * h2);
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const u32x4 c0 = qp[4 * st], c1 = qp[4 * st + 1];
      qa[st][0] = __builtin_bit_cast(bf16x8, u32x4{c0[0], c0[1], c0[0], c0[1]});
      qa[st][1] = __builtin_bit_cast(bf16x8, u32x4{c1[0], c1[1], c1[0], c1[1]});
      ql[st] = __builtin_bit_cast(bf16x8, u32x4{c0[2], c0[3], c1[2], c1[3]});
    }
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

  // ---- S^T = K Q^T over the wave's key tiles: tile t, local key 32 t + i = patch row 2 t + (i >> 4), column i & 15 -------------------
  f32x16 S[NKT];
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) S[t][i] = 0.f;
  // (round 4: the chunk swizzle of an image row is a function of its halo COLUMN, not of its row index: a lane's column is the same in
  // every patch row, so its fragment addresses differ from tile to tile by a compile-time constant -- immediate offsets of the LDS reads
  // instead of ~250 address instructions per wave)
  // The XOR part of an address (chunk pair of the k-step: bits 6 - 7, second chunk of the pair: bit 4) touches only bits below 8 and the tile
  // stride is a multiple of 256: (ka0 + t * stride) ^ m == (ka0 ^ m) + t * stride, so ONE base register per (k-step, chunk) serves every tile.
  const int kr0 = korg + (l31 >> 4) * HC + (l31 & 15);
  const int ka0 = kr0 * ROWB + (((2 * h2) ^ rsw(col_lo + (l31 & 15))) << 4);      // chunk 2 h2 of the row; chunk + 1: ^ 16; step st: ^ (st << 6)
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    bf16x8 kc0[NKT], kc1[NKT], kh[NKT];
    const char* k0p = img + (ka0 ^ (st << 6));
    const char* k1p = img + (ka0 ^ (st << 6) ^ 16);
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
      const u32x4 c0 = *reinterpret_cast<const u32x4*>(k0p + t * (2 * HC * ROWB));
      const u32x4 c1 = *reinterpret_cast<const u32x4*>(k1p + t * (2 * HC * ROWB));
      kc0[t] = __builtin_bit_cast(bf16x8, c0);
      kc1[t] = __builtin_bit_cast(bf16x8, c1);
      kh[t] = __builtin_bit_cast(bf16x8, u32x4{c0[0], c0[1], c1[0], c1[1]});
    }
    // term-major: consecutive MFMAs go to different key tiles
#pragma unroll
    for (int t = 0; t < NKT; ++t) S[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kc0[t], qa[st][0], S[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NKT; ++t) S[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kc1[t], qa[st][1], S[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NKT; ++t) S[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh[t], ql[st], S[t], 0, 0, 0);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  KD_BARRIER();                                          // every wave has its K fragments: the image is free
  stage(2 * a.nh * DH * 4);                              // V, in flight behind the softmax

  // ---- window mask + softmax ------------------------------------------------------------------------------------------------------------
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) S[t][i] = min3f(S[t][i], colv[i & 7], rowv[2 * t + (i >> 3)]);
  float m = -INFINITY;
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int i = 0; i < 16; i += 2) m = max3f(m, S[t][i], S[t][i + 1]);
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float l;
  {
    constexpr float LOG2E = 1.4426950408889634f;
    const f32x2 mb = {-m * LOG2E, -m * LOG2E};
    f32x2 l2 = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        const f32x2 x = __builtin_elementwise_fma(f32x2{S[t][i], S[t][i + 1]}, f32x2{LOG2E, LOG2E}, mb);
        const f32x2 pv = {__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
        S[t][i] = pv.x;
        S[t][i + 1] = pv.y;
        l2 += pv;
      }
    l = l2.x + l2.y;
  }
  l += __shfl_xor(l, 32, 64);
  // probabilities -> hi / lo B fragments: step (t, u) = accumulator registers 8 u .. 8 u + 7 of tile t (k-slots: patch row 2 t + u,
  // columns 4 h2 + {0..3} and + 8)
  bf16x8 ph[NKT][2], pl[NKT][2];
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      u32x4 hi, lo;
      x3::split8(f32x4{S[t][8 * u], S[t][8 * u + 1], S[t][8 * u + 2], S[t][8 * u + 3]},
                 f32x4{S[t][8 * u + 4], S[t][8 * u + 5], S[t][8 * u + 6], S[t][8 * u + 7]}, hi, lo);
      ph[t][u] = __builtin_bit_cast(bf16x8, hi);
      pl[t][u] = __builtin_bit_cast(bf16x8, lo);
    }
  KD_WAIT_VM(0);
  KD_BARRIER();                                          // V image complete

  // ---- O^T = V^T P^T.  V^T fragment of step (t, u), feature block e: rows vr .. vr + 3 and vr + 8 .. + 11 of the image, the lane group's
  // 8-byte pieces [hi4] (or + 8: [lo4]) of chunks (lane & 3) + 4 ((lane >> 4) & 1) + 8 e, transposed by the read itself ---------------------
  f32x16 O[2];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int i = 0; i < 16; ++i) O[e][i] = 0.f;
  const int vr0 = korg + 4 * h2 + ((lane & 15) >> 2);
  const int vc = (lane & 3) + 4 * ((lane >> 4) & 1);
  const int va0 = vr0 * ROWB + ((vc ^ rsw(col_lo + 4 * h2 + ((lane & 15) >> 2))) << 4);
  // four base addresses (feature block e, rows / rows + 8): block e = 1: ^ 128; rows + 8: (^ 32) + 8 rows; the step (t, u) and the lo quad (+ 8)
  // are immediate offsets (the same bits-below-8 argument as for the K fragments)
  const int vb[2][2] = {{va0, (va0 ^ 32) + 8 * ROWB}, {va0 ^ 128, ((va0 ^ 128) ^ 32) + 8 * ROWB}};
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      constexpr int step = HC * ROWB;
      bf16x8 vh[2], vl[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int a0 = vb[e][0] + (2 * t + u) * step, a1 = vb[e][1] + (2 * t + u) * step;
        const u32x2 h0 = tr_read(img, a0), h1 = tr_read(img, a1), l0 = tr_read(img, a0 + 8), l1 = tr_read(img, a1 + 8);
        vh[e] = __builtin_bit_cast(bf16x8, u32x4{h0[0], h0[1], h1[0], h1[1]});
        vl[e] = __builtin_bit_cast(bf16x8, u32x4{l0[0], l0[1], l1[0], l1[1]});
      }
      O[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl[0], ph[t][u], O[0], 0, 0, 0);
      O[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl[1], ph[t][u], O[1], 0, 0, 0);
      O[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh[0], pl[t][u], O[0], 0, 0, 0);
      O[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh[1], pl[t][u], O[1], 0, 0, 0);
      O[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh[0], ph[t][u], O[0], 0, 0, 0);
      O[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh[1], ph[t][u], O[1], 0, 0, 0);
    }

  if (q_ok) {
    const float inv = 1.0f / l;
    float* op = a.out + ((size_t)b * T + q_tok) * (a.nh * DH) + head * DH;
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        st16(op + e * 32 + 8 * g + 4 * h2, f32x4(f32x4{O[e][4 * g], O[e][4 * g + 1], O[e][4 * g + 2], O[e][4 * g + 3]} * inv));
  }
  x3::wg_stamp_end(wgs);
}

// ---- neighbourhood core, kernel sizes 11 and 13 (no shipped config uses them: the form that covers the reference's interface,
// image_transformer_v2.py:399-410, not a tuned one) ----------------------------------------------------------------------------------------
// The 4 x 8 query block of a wave now needs a patch of (4 + KS - 1) rows x (8 + KS - 1) = 18 / 20 columns: no power-of-two width any
// more, so the patch is walked DENSELY as in attn_bf16.hip's form of these sizes: local key kl = 20 r + c (columns padded to 20, a
// multiple of 4: the 4-key groups of the V^T reads never straddle two patch rows), tiles of 32 keys, every key's (row, column) by constant
// division, validity per accumulator register from those.  One workgroup per CU (127 / 150 KiB of halo image: K and V still share it).
template <int KS>
struct NaWide {
  static constexpr int HR = NA_TH + KS - 1, HC = NA_TW + KS - 1;
  static constexpr int PR = 4 + KS - 1, PC = 20;                            // patch rows; padded patch width
  static constexpr int NKT = (PR * PC + 31) / 32;
  static constexpr int ROWS = ((HR * HC + 2 * PC + 3) / 4) * 4;             // slack: padded columns and the last tile's tail poke past the halo
  static constexpr int LDS = ROWS * ROWB;
  static_assert(8 + KS - 1 <= PC && LDS <= 160 * 1024, "patch width / LDS");
};

template <int KS>
__global__ __launch_bounds__(256, 1) void attn_na2d_x3_wide_kernel(const NArgs a) {
  using G = NaWide<KS>;
  constexpr int HR = G::HR, HC = G::HC, PR = G::PR, PC = G::PC, NKT = G::NKT, ROWS = G::ROWS;
  extern __shared__ __attribute__((aligned(16))) char img[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, h2 = lane >> 5;
  const auto warm = code_warm_begin<16384>((int)blockIdx.x < a.warm && tid < 64);
  const int wy_ = wid >> 1, wx_ = wid & 1;
  const int tiles_x = (a.W + NA_TW - 1) / NA_TW, tiles_y = (a.H + NA_TH - 1) / NA_TH;
  int r = blockIdx.x;
  const int tx = r % tiles_x; r /= tiles_x;
  const int ty = r % tiles_y; r /= tiles_y;
  const int head = r % a.nh, b = r / a.nh;
  const int T = a.H * a.W;
  const size_t row_bytes = (size_t)3 * a.nh * DH * 4;
  const char* base = reinterpret_cast<const char*>(a.qkv) + (size_t)b * T * row_bytes + head * (DH * 4);
  const int ty0 = ty * NA_TH, tx0 = tx * NA_TW;
  const int hy0 = max(0, min(ty0 - KS / 2, a.H - HR)), hx0 = max(0, min(tx0 - KS / 2, a.W - HC));
  // image row 4 pc + (lane >> 4) = halo position (row / HC, row % HC); rows past the halo (the slack) take a real token: outside every window
  auto stage = [&](int part_bytes) {
    for (int pc = wid; pc < ROWS / 4; pc += 4) {
      const int row = 4 * pc + (lane >> 4);
      const int ky = min(hy0 + row / HC, a.H - 1), kx = min(hx0 + row % HC, a.W - 1);
      const char* src = base + (size_t)(unsigned)((ky * a.W + kx) * (int)row_bytes + (((lane & 15) ^ rsw(row)) << 4));
      glds16(src + part_bytes, img + pc * 1024);
    }
  };
  stage(a.nh * DH * 4);                                  // K
  const int qy_raw = ty0 + 4 * wy_ + (l31 >> 3), qx_raw = tx0 + 8 * wx_ + (l31 & 7);
  const bool q_ok = qy_raw < a.H && qx_raw < a.W;
  const int qy = min(qy_raw, a.H - 1), qx = min(qx_raw, a.W - 1);
  const int q_tok = qy * a.W + qx;
  bf16x8 qh[4], ql[4];
  {
    const u32x4* qp = reinterpret_cast<const u32x4*>(base + (size_t)q_tok * row_bytes + 32 * h2);
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const u32x4 c0 = qp[4 * st], c1 = qp[4 * st + 1];
      qh[st] = __builtin_bit_cast(bf16x8, u32x4{c0[0], c0[1], c1[0], c1[1]});
      ql[st] = __builtin_bit_cast(bf16x8, u32x4{c0[2], c0[3], c1[2], c1[3]});
    }
  }
  const int wy = max(0, min(qy - KS / 2, a.H - KS)) - hy0, wx = max(0, min(qx - KS / 2, a.W - KS)) - hx0;
  const int row_lo = min(max(0, min(min(ty0 + 4 * wy_, a.H - 1) - KS / 2, a.H - KS)) - hy0, HR - PR);
  const int col_lo = min(max(0, min(min(tx0 + 8 * wx_, a.W - 1) - KS / 2, a.W - KS)) - hx0, HC - (8 + KS - 1));
  const int korg = row_lo * HC + col_lo;
  const int r0 = wy - row_lo, c0w = wx - col_lo;
  auto img_row = [&](int kl) -> int { return min(korg + (kl / PC) * HC + (kl % PC), ROWS - 1); };     // image row of local key kl
  KD_WAIT_VM(0);
  code_warm_end(warm);
  KD_BARRIER();

  // ---- S^T = K Q^T, tile by tile: local key 32 t + l31 ----------------------------------------------------------------------------------
  f32x16 S[NKT];
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) S[t][i] = 0.f;
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    const int kr = img_row(32 * t + l31);
    const int ka = kr * ROWB + (((2 * h2) ^ rsw(kr)) << 4);
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const u32x4 c0 = *reinterpret_cast<const u32x4*>(img + (ka ^ (st << 6)));
      const u32x4 c1 = *reinterpret_cast<const u32x4*>(img + (ka ^ (st << 6) ^ 16));
      const bf16x8 kh = __builtin_bit_cast(bf16x8, u32x4{c0[0], c0[1], c1[0], c1[1]});
      const bf16x8 kl = __builtin_bit_cast(bf16x8, u32x4{c0[2], c0[3], c1[2], c1[3]});
      S[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh[st], S[t], 0, 0, 0);
      S[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql[st], S[t], 0, 0, 0);
      S[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh[st], S[t], 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  KD_BARRIER();                                          // every wave has its K fragments: the image is free
  stage(2 * a.nh * DH * 4);                              // V, in flight behind the softmax

  // ---- window mask (accumulator register i of tile t: local key 32 t + (i & 3) + 8 (i >> 2) + 4 h2) + softmax ----------------------------
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int kl = 32 * t + (i & 3) + 8 * (i >> 2) + 4 * h2;
      const int pr = kl / PC, pcn = kl % PC;
      const bool valid = (unsigned)(pr - r0) < (unsigned)KS && (unsigned)(pcn - c0w) < (unsigned)KS && pr < PR;
      S[t][i] = valid ? S[t][i] : -INFINITY;
    }
  float m = -INFINITY;
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int i = 0; i < 16; i += 2) m = max3f(m, S[t][i], S[t][i + 1]);
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float l;
  {
    constexpr float LOG2E = 1.4426950408889634f;
    const f32x2 mb = {-m * LOG2E, -m * LOG2E};
    f32x2 l2 = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        const f32x2 x = __builtin_elementwise_fma(f32x2{S[t][i], S[t][i + 1]}, f32x2{LOG2E, LOG2E}, mb);
        const f32x2 pv = {__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
        S[t][i] = pv.x;
        S[t][i + 1] = pv.y;
        l2 += pv;
      }
    l = l2.x + l2.y;
  }
  l += __shfl_xor(l, 32, 64);
  KD_WAIT_VM(0);
  KD_BARRIER();                                          // V image complete

  // ---- O^T = V^T P^T: k-slots of lane-half h2 at step (t, u) are local keys 32 t + 16 u + 4 h2 + {0..3} and the same + 8 -- two 4-key
  // groups, each inside one patch row; the probabilities are split step by step --------------------------------------------------------------
  f32x16 O[2];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int i = 0; i < 16; ++i) O[e][i] = 0.f;
  const int vsub = (lane & 15) >> 2;
  const int vc = (lane & 3) + 4 * ((lane >> 4) & 1);
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      u32x4 phi, plo;
      x3::split8(f32x4{S[t][8 * u], S[t][8 * u + 1], S[t][8 * u + 2], S[t][8 * u + 3]},
                 f32x4{S[t][8 * u + 4], S[t][8 * u + 5], S[t][8 * u + 6], S[t][8 * u + 7]}, phi, plo);
      const bf16x8 ph = __builtin_bit_cast(bf16x8, phi), pl = __builtin_bit_cast(bf16x8, plo);
      const int k0 = 32 * t + 16 * u + 4 * h2;
      const int ra = min(img_row(k0) + vsub, ROWS - 1), rb = min(img_row(k0 + 8) + vsub, ROWS - 1);
      bf16x8 vh[2], vl[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int a0 = ra * ROWB + (((vc + 8 * e) ^ rsw(ra)) << 4), a1 = rb * ROWB + (((vc + 8 * e) ^ rsw(rb)) << 4);
        const u32x2 h0 = tr_read(img, a0), h1 = tr_read(img, a1), l0 = tr_read(img, a0 + 8), l1 = tr_read(img, a1 + 8);
        vh[e] = __builtin_bit_cast(bf16x8, u32x4{h0[0], h0[1], h1[0], h1[1]});
        vl[e] = __builtin_bit_cast(bf16x8, u32x4{l0[0], l0[1], l1[0], l1[1]});
      }
      O[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl[0], ph, O[0], 0, 0, 0);
      O[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl[1], ph, O[1], 0, 0, 0);
      O[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh[0], pl, O[0], 0, 0, 0);
      O[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh[1], pl, O[1], 0, 0, 0);
      O[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh[0], ph, O[0], 0, 0, 0);
      O[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh[1], ph, O[1], 0, 0, 0);
    }
  if (q_ok) {
    const float inv = 1.0f / l;
    float* op = a.out + ((size_t)b * T + q_tok) * (a.nh * DH) + head * DH;
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        st16(op + e * 32 + 8 * g + 4 * h2, f32x4(f32x4{O[e][4 * g], O[e][4 * g + 1], O[e][4 * g + 2], O[e][4 * g + 3]} * inv));
  }
}

// ---- global attention, T = 32 NT keys (64, 128, 256) -----------------------------------------------------------------------------------------
// SDPA / flash-attn of the global level (image_transformer_v2.py:383,392) on the same operand scheme: one workgroup per (sample, head,
// block of 32 QW queries), QW = min(4, NT) waves; ALL keys of the (sample, head) go through one [T][256 B] image, K first, then V (64 KiB at
// T = 256: two workgroups per CU -- the round-1 core ran one 8-wave workgroup per CU, its halo staging through registers).  A wave owns
// 32 queries and the whole score row (NT tiles, no online softmax); the probabilities are split tile by tile inside the PV loop.
// (One 8-wave workgroup per (sample, head) with K AND V images requested together -- one memory latency instead of two -- was measured
// slower, 26.1 vs 21.2 us: with one workgroup per CU nothing overlaps that latency, and the 8-wave form spills.)
struct GArgs {
  const float* qkv; float* out;
  int batch, T, nh;
  int warm;
  unsigned long long* clk;
};

template <int NT>
__global__ __launch_bounds__((NT < 4 ? NT : 4) * 64, 2) void attn_global_x3_kernel(const GArgs a) {
  constexpr int QW = NT < 4 ? NT : 4, T = 32 * NT, QB = NT / QW;      // query blocks per (sample, head)
  extern __shared__ __attribute__((aligned(16))) char img[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, h2 = lane >> 5;
  const auto warm = code_warm_begin<10240>((int)blockIdx.x < a.warm && tid < 64);
  const x3::WgStamp wgs = x3::wg_stamp_begin(a.clk);
  int r;
  {   // XCD-aware order: the query blocks of one (sample, head) -- same K, V -- run on ONE L2
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int q = nwg >> 3, rem = nwg & 7;
    r = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + k;
  }
  const int qb = r % QB; r /= QB;
  const int head = r % a.nh, b = r / a.nh;
  const size_t row_bytes = (size_t)3 * a.nh * DH * 4;
  const char* base = reinterpret_cast<const char*>(a.qkv) + (size_t)b * T * row_bytes + head * (DH * 4);
  auto stage = [&](int part_bytes) {
    for (int pc = wid; pc < T / 4; pc += QW) {
      const int row = 4 * pc + (lane >> 4);
      glds16(base + (size_t)row * row_bytes + part_bytes + (((lane & 15) ^ rsw(row)) << 4), img + pc * 1024);
    }
  };
  stage(a.nh * DH * 4);                                  // K
  const int q_tok = 32 * (QW * qb + wid) + l31;
  bf16x8 qh[4], ql[4];
  {
    const u32x4* qp = reinterpret_cast<const u32x4*>(base + (size_t)q_tok * row_bytes + 32 * h2);
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const u32x4 c0 = qp[4 * st], c1 = qp[4 * st + 1];
      qh[st] = __builtin_bit_cast(bf16x8, u32x4{c0[0], c0[1], c1[0], c1[1]});
      ql[st] = __builtin_bit_cast(bf16x8, u32x4{c0[2], c0[3], c1[2], c1[3]});
    }
  }
  KD_WAIT_VM(0);
  code_warm_end(warm);
  KD_BARRIER();

  f32x16 S[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) S[t][i] = 0.f;
  const int ka = l31 * ROWB + (((2 * h2) ^ rsw(l31)) << 4);      // key 32 t + l31: + t * 32 * ROWB (32 rows on: the same swizzle word)
#pragma unroll
  for (int st = 0; st < 4; ++st) {
#pragma unroll
    for (int t0 = 0; t0 < NT; t0 += 4) {                        // four tiles' fragments at a time (registers)
      constexpr int TB = NT < 4 ? NT : 4;
      bf16x8 kh[TB], kl[TB];
#pragma unroll
      for (int t = 0; t < TB; ++t) {
        const char* kp = img + (t0 + t) * 32 * ROWB + (ka ^ (st << 6));
        const u32x4 c0 = *reinterpret_cast<const u32x4*>(kp);
        const u32x4 c1 = *reinterpret_cast<const u32x4*>(img + (t0 + t) * 32 * ROWB + (ka ^ (st << 6) ^ 16));
        kh[t] = __builtin_bit_cast(bf16x8, u32x4{c0[0], c0[1], c1[0], c1[1]});
        kl[t] = __builtin_bit_cast(bf16x8, u32x4{c0[2], c0[3], c1[2], c1[3]});
      }
#pragma unroll
      for (int t = 0; t < TB; ++t) S[t0 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl[t], qh[st], S[t0 + t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < TB; ++t) S[t0 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh[t], ql[st], S[t0 + t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < TB; ++t) S[t0 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh[t], qh[st], S[t0 + t], 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  KD_BARRIER();                                          // every wave has its K fragments: the image is free
  stage(2 * a.nh * DH * 4);                              // V, in flight behind the softmax

  float m = -INFINITY;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; i += 2) m = max3f(m, S[t][i], S[t][i + 1]);
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float l;
  {
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
    l = l2.x + l2.y;
  }
  l += __shfl_xor(l, 32, 64);
  KD_WAIT_VM(0);
  KD_BARRIER();                                          // V image complete

  f32x16 O[2];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int i = 0; i < 16; ++i) O[e][i] = 0.f;
  const int vr0 = 4 * h2 + ((lane & 15) >> 2);
  const int vc = (lane & 3) + 4 * ((lane >> 4) & 1);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      // k-slots of step (t, u): keys 32 t + 16 u + 4 h2 + {0..3} and + 8
      const int vr = vr0 + 32 * t + 16 * u;
      const int va = vr * ROWB + ((vc ^ rsw(vr)) << 4);
      u32x4 phi, plo;
      x3::split8(f32x4{S[t][8 * u], S[t][8 * u + 1], S[t][8 * u + 2], S[t][8 * u + 3]},
                 f32x4{S[t][8 * u + 4], S[t][8 * u + 5], S[t][8 * u + 6], S[t][8 * u + 7]}, phi, plo);
      const bf16x8 ph = __builtin_bit_cast(bf16x8, phi), pl = __builtin_bit_cast(bf16x8, plo);
      bf16x8 vh[2], vl[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int a0 = va ^ (e << 7), a1 = (a0 ^ 32) + 8 * ROWB;
        const u32x2 h0 = tr_read(img, a0), h1 = tr_read(img, a1), l0 = tr_read(img, a0 + 8), l1 = tr_read(img, a1 + 8);
        vh[e] = __builtin_bit_cast(bf16x8, u32x4{h0[0], h0[1], h1[0], h1[1]});
        vl[e] = __builtin_bit_cast(bf16x8, u32x4{l0[0], l0[1], l1[0], l1[1]});
      }
      O[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl[0], ph, O[0], 0, 0, 0);
      O[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl[1], ph, O[1], 0, 0, 0);
      O[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh[0], pl, O[0], 0, 0, 0);
      O[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh[1], pl, O[1], 0, 0, 0);
      O[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh[0], ph, O[0], 0, 0, 0);
      O[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh[1], ph, O[1], 0, 0, 0);
    }
  {
    const float inv = 1.0f / l;
    float* op = a.out + ((size_t)b * T + q_tok) * (a.nh * DH) + head * DH;
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        st16(op + e * 32 + 8 * g + 4 * h2, f32x4(f32x4{O[e][4 * g], O[e][4 * g + 1], O[e][4 * g + 2], O[e][4 * g + 3]} * inv));
  }
  x3::wg_stamp_end(wgs);
}

template <int NT>
static int launch_global(const GArgs& a, hipStream_t s) {
  constexpr int QW = NT < 4 ? NT : 4, T = 32 * NT, lds = T * ROWB;
  auto kern = attn_global_x3_kernel<NT>;
  static LdsAttr attr_set;
  attr_set.ensure(reinterpret_cast<const void*>(kern), lds);
  const long nb = (long)a.batch * a.nh * (NT / QW);
  LaunchScope prof("attn_global_x3", 4.0 * (double)a.batch * a.nh * T * T * DH, 16.0 * (double)a.batch * T * a.nh * DH, s);
  hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(QW * 64), lds, s, a);
  return check_launch("kd_attn_global_f32(x3)");
}

}  // namespace x3a

// Called by kd_attn_global_f32 (attn_f32.hip) for prep == 2, KD_PREC_SPLIT3, T = 64 / 128 / 256.  Returns 1 if not taken.
int attn_global_x3_try(const float* qkv, float* out, int batch, int T, int nh, hipStream_t s, int* rc) {
  using namespace x3a;
  if (!option("attn_x3", 1) || (T != 64 && T != 128 && T != 256)) return 1;
  GArgs a{qkv, out, batch, T, nh, option("code_warm", KD_CODE_WARM_DEFAULT), x3::g_clk};
  *rc = T == 256 ? launch_global<8>(a, s) : (T == 128 ? launch_global<4>(a, s) : launch_global<2>(a, s));
  return 0;
}

template <int KS, bool WIDE>
static int launch_na(const x3a::NArgs& a, hipStream_t s) {
  using namespace x3a;
  constexpr int lds = WIDE ? NaWide<(WIDE ? KS : 11)>::LDS : NaGeo<(WIDE ? 7 : KS)>::LDS;
  const void* kern;
  if constexpr (WIDE) kern = reinterpret_cast<const void*>(attn_na2d_x3_wide_kernel<KS>);
  else kern = reinterpret_cast<const void*>(attn_na2d_x3_kernel<KS>);
  static LdsAttr attr_set;
  attr_set.ensure(kern, lds);
  const long nb = (long)a.batch * a.nh * ((a.H + NA_TH - 1) / NA_TH) * ((a.W + NA_TW - 1) / NA_TW);
  char nm[64] = "attn_na2d_x3";
  if (prof_on()) {
    if (KS == 7) snprintf(nm, sizeof(nm), "attn_na2d_x3 %dx%d nh=%d", a.H, a.W, a.nh);
    else snprintf(nm, sizeof(nm), "attn_na2d_x3 k%d %dx%d nh=%d", KS, a.H, a.W, a.nh);
  }
  LaunchScope prof(nm, 4.0 * a.batch * (double)a.H * a.W * a.nh * DH * KS * KS, 16.0 * a.batch * (double)a.H * a.W * a.nh * DH, s);
  if constexpr (WIDE) hipLaunchKernelGGL(attn_na2d_x3_wide_kernel<KS>, dim3((unsigned)nb), dim3(256), lds, s, a);
  else hipLaunchKernelGGL(attn_na2d_x3_kernel<KS>, dim3((unsigned)nb), dim3(256), lds, s, a);
  return check_launch("kd_attn_na2d_f32(x3)");
}

// Called by kd_attn_na2d_f32 (attn_f32.hip) for prep == 2 (operands stored split), kernel sizes 3 .. 13 (odd).  Returns 1 if not taken.
int attn_na2d_x3_try(const float* qkv, float* out, int batch, int H, int W, int nh, int ks, hipStream_t s, int* rc) {
  using namespace x3a;
  if (!option("attn_x3", 1)) return 1;
  NArgs a{qkv, out, batch, H, W, nh, option("code_warm", KD_CODE_WARM_DEFAULT), x3::g_clk};
  switch (ks) {
    case 3: *rc = launch_na<3, false>(a, s); return 0;
    case 5: *rc = launch_na<5, false>(a, s); return 0;
    case 7: *rc = launch_na<7, false>(a, s); return 0;
    case 9: *rc = launch_na<9, false>(a, s); return 0;
    case 11: *rc = launch_na<11, true>(a, s); return 0;
    case 13: *rc = launch_na<13, true>(a, s); return 0;
    default: return 1;
  }
}

}  // namespace kd

KD_TEXT_PAD(attn_x3)      // last function of this code object: kd_common.h, code warm-up
