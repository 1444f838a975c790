// Attention cores of the HDiT denoiser in bf16 mode (KD_PREC_BF16), gfx950.
//
//   kd_attn_global_bf16  dense softmax attention per (sample, head): T <= 256 keys held whole in LDS, longer sequences streamed
//                        in 128-key blocks with an online softmax          (SDPA / flash-attn, image_transformer_v2.py:383,392)
//   kd_attn_window_bf16  shifted-window attention, roll / window / mask / unwindow as index arithmetic      (:319-337, :253-316)
//   kd_attn_na2d_bf16    neighbourhood attention, clamped windows, kernel sizes 3, 5, 7, 9                     (natten na2d, :428)
//
// qkv is the bf16 output of the qkv GEMM [tokens, 3, nh, 64] with q, k ALREADY prepared (cosine-sim scale + RoPE in that GEMM's
// epilogue); out is bf16 [tokens, nh * 64].  One scheme for all three cores:
//   * K rows and V rows go HBM -> LDS by global_load_lds, 8 rows x 128 bytes per wave-instruction, as row-major [key][64] images
//     whose 16-byte chunks are XOR-swizzled on the source side (asw() below): no staging VALU work, no ds_write pass;
//   * S^T = K Q^T : K rows are the MFMA A operand (conflict-free ds_read_b128), Q the B operand straight from HBM registers; a
//     lane owns ONE query (column) and 16 keys per 32-key tile, so the softmax reductions are in-lane plus one half-wave shuffle;
//   * O^T = V^T P^T : P (fp32 -> bf16 in registers) is the B operand as it sits in the accumulators; the V^T fragments come out
//     of the ROW-major V image through ds_read_b64_tr_b16 (hardware 4x4 transpose: lane i of a 16-lane group receives column i
//     of the four rows the group addresses -- benchmarks/probe/ds_read_tr_probe.cpp records the mapping);
//   * the O^T accumulators are the C-layout of the bf16 GEMM epilogues: normalise, pair the half-waves, 16-byte stores.
// fp32: scores, softmax, accumulators.  bf16: q, k, v, P, out.
#include "attn_bf16_core.h"

namespace kd {
namespace b16 {

// ---- dense core, the whole key set in LDS: global (T <= 256) and windows --------------------------------------------------
// NT = key tiles of 32; QW = waves per workgroup, wave w owns queries 32 (qblk * QW + w) ..; several workgroups per (sample, head)
// when QW < NT (each stages the full K / V: L2 hits, and the chip sees 2+ workgroups per CU instead of one).
template <int MODE, int NT, int QW>
__global__ __launch_bounds__(QW * 64, (QW <= 4 && MODE != MODE_WINDOW16) ? 2 : 1) void attn_dense_bf16_kernel(const DArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TP = NT * 32;
  char* Kimg = smem;
  char* Vimg = smem + TP * 128;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, h2 = lane >> 5;
  const auto warm = code_warm_begin<(MODE == MODE_WINDOW16 ? 16 : 10) * 1024>((int)blockIdx.x < a.warm && tid < 64);
  constexpr int NQB = (NT + QW - 1) / QW;            // query blocks per problem
  int r = blockIdx.x;
  const int qblk = r % NQB; r /= NQB;
  int b, head, wi = 0, wj = 0;
  if (MODE == MODE_GLOBAL) {
    head = r % a.nh; b = r / a.nh;
  } else {
    const int nww = a.W >> WinLog2<MODE>::v, nwh = a.H >> WinLog2<MODE>::v;
    wj = r % nww; r /= nww; wi = r % nwh; r /= nwh; head = r % a.nh; b = r / a.nh;
  }
  const int T = MODE == MODE_GLOBAL ? a.T : (1 << (2 * WinLog2<MODE>::v));          // key slots of this problem
  const size_t row_bytes = (size_t)3 * a.nh * DH * 2;
  const char* base = reinterpret_cast<const char*>(a.qkv) + (size_t)b * a.T * row_bytes + head * (DH * 2);

  // ---- K and V rows -> LDS images (8 rows per wave-instruction) ----------------------------------------------------------
  for (int pc = wid; pc < TP / 8; pc += QW) {
    const int row = 8 * pc + (lane >> 3);
    const int tok = slot_token<MODE>(a, min(row, T - 1), wi, wj);
    const int q = (lane & 7) ^ asw(row);
    const char* src = base + (size_t)tok * row_bytes + q * 16;
    glds16(src + a.nh * DH * 2, Kimg + pc * 1024);
    glds16(src + 2 * a.nh * DH * 2, Vimg + pc * 1024);
  }
  // ---- this lane's query: dims 16 st + 8 h2 .. +7 ------------------------------------------------------------------------------
  const int q_slot = (qblk * QW + wid) * 32 + l31;
  const bool q_ok = q_slot < T;
  const int q_tok = slot_token<MODE>(a, min(q_slot, T - 1), wi, wj);
  bf16x8 qf[4];
  {
    const u32x4* qp = reinterpret_cast<const u32x4*>(base + (size_t)q_tok * row_bytes + 16 * h2);
#pragma unroll
    for (int st = 0; st < 4; ++st) qf[st] = __builtin_bit_cast(bf16x8, qp[2 * st]);
  }
  KD_WAIT_VM(0);
  code_warm_end(warm);
  KD_BARRIER();
  if ((qblk * QW + wid) * 32 >= T) return;            // a wave without queries (no barrier follows)

  // ---- S^T = K Q^T ----------------------------------------------------------------------------------------------------------------
  f32x16 S[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) S[t][i] = 0.f;
  const int ka = l31 * 128 + ((h2 ^ asw(l31)) << 4);          // K fragment of tile t, k-step st: (ka ^ 32 st) + 4096 t
#pragma unroll
  for (int st = 0; st < 4; ++st) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Kimg + (ka ^ (32 * st)) + 4096 * t);
      S[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[st], S[t], 0, 0, 0);
    }
  }
  // ---- masks + softmax over keys -----------------------------------------------------------------------------------------------
  const int q_region = (MODE != MODE_GLOBAL) ? slot_region<MODE>(min(q_slot, T - 1), wi, wj, a.shift) : 0;
  if (MODE != MODE_GLOBAL || (T & 31)) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int ks = t * 32 + mfma32_row(i, lane);
        if (MODE == MODE_GLOBAL) {
          if (t * 32 + 32 > T) S[t][i] += (ks < T) ? 0.f : -INFINITY;
        } else {
          if ((1 << (2 * WinLog2<MODE>::v)) < TP) S[t][i] += (ks < T) ? 0.f : -INFINITY;
          if (a.shift) S[t][i] += (slot_region<MODE>(min(ks, T - 1), wi, wj, a.shift) == q_region) ? 0.f : -INFINITY;
        }
      }
  }
  const float m = score_max<NT>(S);
  float l = score_exp<NT>(S, m);
  l += __shfl_xor(l, 32, 64);

  // ---- O^T = V^T P^T ---------------------------------------------------------------------------------------------------------------
  f32x16 O[2];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int i = 0; i < 16; ++i) O[e][i] = 0.f;
  const int va = vt_addr(4 * h2 + vt_lane_row(lane), lane);      // k-step (t, u): rows + 32 t + 16 u, same swizzle word
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) pv_step(O, Vimg + (32 * t + 16 * u) * 128, va, p_frag(S[t], u));
  store_o(a.out + ((size_t)b * a.T + q_tok) * (a.nh * DH) + head * DH, O, 1.0f / l, h2, q_ok);
}

// ---- global core for T > 256: 128-key blocks double-buffered through LDS, online softmax ---------------------------------------
constexpr int GL_QW = 8, GL_KB = 128, GL_NTK = GL_KB / 32;
constexpr int GL_IMG = GL_KB * 128, GL_BUF = 2 * GL_IMG, GL_LDS = 2 * GL_BUF;

__global__ __launch_bounds__(GL_QW * 64) void attn_long_bf16_kernel(const DArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, h2 = lane >> 5;
  const auto warm = code_warm_begin<6144>((int)blockIdx.x < a.warm && tid < 64);
  const int T = a.T, nqb = (T + GL_QW * 32 - 1) / (GL_QW * 32);
  int r = blockIdx.x;
  const int qb = r % nqb; r /= nqb;
  const int head = r % a.nh, b = r / a.nh;
  const size_t row_bytes = (size_t)3 * a.nh * DH * 2;
  const char* base = reinterpret_cast<const char*>(a.qkv) + (size_t)b * T * row_bytes + head * (DH * 2);
  const int q_slot = qb * (GL_QW * 32) + wid * 32 + l31;
  const bool q_ok = q_slot < T;
  const int q_tok = min(q_slot, T - 1);
  // block kb -> buffer kb & 1: this wave moves pieces 2 wid, 2 wid + 1 of the K image and of the V image (4 per block)
  auto issue = [&](int kb) {
    char* buf = smem + (kb & 1) * GL_BUF;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int pc = 2 * wid + j, row = 8 * pc + (lane >> 3);
      const int tok = min(kb * GL_KB + row, T - 1);
      const int q = (lane & 7) ^ asw(row);
      const char* src = base + (size_t)tok * row_bytes + q * 16;
      glds16(src + a.nh * DH * 2, buf + pc * 1024);
      glds16(src + 2 * a.nh * DH * 2, buf + GL_IMG + pc * 1024);
    }
  };
  bf16x8 qf[4];
  {
    const u32x4* qp = reinterpret_cast<const u32x4*>(base + (size_t)q_tok * row_bytes + 16 * h2);
#pragma unroll
    for (int st = 0; st < 4; ++st) qf[st] = __builtin_bit_cast(bf16x8, qp[2 * st]);
  }
  KD_WAIT_VM(0);                                   // q in registers before any block is in flight (counted waits below see only blocks)
  issue(0);
  f32x16 O[2];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int i = 0; i < 16; ++i) O[e][i] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const int ka = l31 * 128 + ((h2 ^ asw(l31)) << 4);
  const int va = vt_addr(4 * h2 + vt_lane_row(lane), lane);
  const int nkb = (T + GL_KB - 1) / GL_KB;
  code_warm_end(warm);
  for (int kb = 0; kb < nkb; ++kb) {
    if (kb + 1 < nkb) { issue(kb + 1); KD_WAIT_VM(4); } else { KD_WAIT_VM(0); }
    KD_BARRIER();                                  // block kb is in for every wave
    const char* Kimg = smem + (kb & 1) * GL_BUF;
    const char* Vimg = Kimg + GL_IMG;
    const int k0 = kb * GL_KB;
    f32x16 S[GL_NTK];
#pragma unroll
    for (int t = 0; t < GL_NTK; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) S[t][i] = 0.f;
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int t = 0; t < GL_NTK; ++t) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Kimg + (ka ^ (32 * st)) + 4096 * t);
        S[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[st], S[t], 0, 0, 0);
      }
    if (k0 + GL_KB > T) {
#pragma unroll
      for (int t = 0; t < GL_NTK; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) S[t][i] += (k0 + t * 32 + mfma32_row(i, lane) < T) ? 0.f : -INFINITY;
    }
    const float m_new = fmaxf(m_run, score_max<GL_NTK>(S));
    const float alpha = __expf(m_run - m_new);       // first block: exp(-inf) = 0
    m_run = m_new;
    l_run = l_run * alpha + score_exp<GL_NTK>(S, m_new);
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int i = 0; i < 16; ++i) O[e][i] *= alpha;
#pragma unroll
    for (int t = 0; t < GL_NTK; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u) pv_step(O, Vimg + (32 * t + 16 * u) * 128, va, p_frag(S[t], u));
    KD_BARRIER();                                  // every wave is done with buffer kb & 1 before block kb + 2 overwrites it
  }
  const float l = l_run + __shfl_xor(l_run, 32, 64);
  store_o(a.out + ((size_t)b * T + q_tok) * (a.nh * DH) + head * DH, O, 1.0f / l, h2, q_ok);
}

// ---- neighbourhood core ---------------------------------------------------------------------------------------------------------------
// One 256-thread workgroup per (sample, head, 8x16 query tile); KS = kernel size (3, 5, 7, 9).  The (8 + KS - 1) x (16 + KS - 1)
// key halo of the tile goes to LDS once (K image + V image).  Wave (wy, wx) owns the 4x8 query block at rows 4wy.., columns
// 8wx..: the clamped windows of its queries lie inside a PR x 16-or-32-column patch of the halo (PR = 4 + KS - 1 rows), walked
// as local keys kl = PW * r + c (PW = 16 for KS <= 9, else 32) in tiles of 32; keys outside a query's window get a -inf bias
// from per-lane bit words (one word per tile).  Out-of-image halo positions (images smaller than the halo) are clamped to a real
// token: they are outside every window, so their probability is exactly 0.
struct NArgs {
  const u16* qkv; u16* out;
  int batch, H, W, nh;
  int warm;                  // code warm-up workgroups (kd_common.h)
};
constexpr int NA_TH = 8, NA_TW = 16;

template <int KS>
struct NaGeo {
  static constexpr int HR = NA_TH + KS - 1, HC = NA_TW + KS - 1;            // halo
  static constexpr int PR = 4 + KS - 1;                                      // patch rows of a wave
  static constexpr int PW = (8 + KS - 1 <= 16) ? 16 : 32;                    // patch width (keys per patch row)
  static constexpr int NKT = (PR * PW + 31) / 32;                            // key tiles per wave
  // image rows: a wave's patch may poke past the halo's last key -- the highest row any fragment read touches is
  // (HR - 1) HC + (HC - (8 + KS - 1)) + 15 = HR HC - KS + 8.  Kept tight: at KS = 7 the image takes 39 KiB (round 6: one image for K, then V), THREE workgroups per CU
  static constexpr int ROWS = ((HR * HC - KS + 9 + 7) / 8) * 8;
  static constexpr int LDS = ROWS * 128;                                     // ONE image: K, then V
};

// Round 6: K and V go through ONE image -- the V rows are requested behind the barrier that ends the score MFMAs and land behind the
// softmax -- so a workgroup holds 39 KiB instead of 78 (KS = 7) and THREE of them fit a CU (150 registers): a workgroup is a serial chain
// "wait for the halo -> scores -> softmax -> PV -> stores" whose wait nobody but the CU's other workgroups can cover.  Same box, same
// run: level 0 33.5 -> 30.95 us, level 1 20.5 -> 18.7 us, bf16 mode 443.4 -> 447.4 images/s (two images, two workgroups per CU before).
template <int KS>
__global__ __launch_bounds__(256, NaGeo<KS>::LDS <= 53 * 1024 ? 3 : 2) void attn_na2d_bf16_kernel(const NArgs a) {
  using G = NaGeo<KS>;
  constexpr int HR = G::HR, HC = G::HC, PR = G::PR, PW = G::PW, NKT = G::NKT, ROWS = G::ROWS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Kimg = smem;
  char* Vimg = smem;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, h2 = lane >> 5;
  const auto warm = code_warm_begin<7168>((int)blockIdx.x < a.warm && tid < 64);
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
  const size_t row_bytes = (size_t)3 * a.nh * DH * 2;
  const char* base = reinterpret_cast<const char*>(a.qkv) + (size_t)b * T * row_bytes + head * (DH * 2);
  const int ty0 = ty * NA_TH, tx0 = tx * NA_TW;
  const int hy0 = max(0, min(ty0 - KS / 2, a.H - HR)), hx0 = max(0, min(tx0 - KS / 2, a.W - HC));

  // ---- halo rows -> K / V images: image row 8 pc + (lane >> 3) = halo position (y, x), walked 32 rows at a time ----------------
  static_assert(PW == 16, "kernel sizes up to 9: 16-key patch rows");
  auto stage = [&](bool want_k, bool want_v) {
    int row = 8 * wid + (lane >> 3);
    int y = row / HC, x = row % HC;
    constexpr int DY = 32 / HC, DX = 32 % HC;
    for (int pc = wid; pc < ROWS / 8; pc += 4) {
      // rows past the halo's last key (a patch may poke there) take any real token: they are outside every window
      const int ky = min(hy0 + y, a.H - 1), kx = min(hx0 + x, a.W - 1);
      const char* src = base + (size_t)(unsigned)((ky * a.W + kx) * (int)row_bytes + (((lane & 7) ^ asw(row)) << 4));
      if (want_k) glds16(src + a.nh * DH * 2, Kimg + pc * 1024);
      if (want_v) glds16(src + 2 * a.nh * DH * 2, Vimg + pc * 1024);
      row += 32; x += DX; y += DY;
      if (x >= HC) { x -= HC; ++y; }
    }
  };
  stage(true, false);
  // ---- this lane's query ------------------------------------------------------------------------------------------------------------
  const int qy_raw = ty0 + 4 * wy_ + (l31 >> 3), qx_raw = tx0 + 8 * wx_ + (l31 & 7);
  const bool q_ok = qy_raw < a.H && qx_raw < a.W;
  const int qy = min(qy_raw, a.H - 1), qx = min(qx_raw, a.W - 1);
  const int q_tok = qy * a.W + qx;
  bf16x8 qf[4];
  {
    const u32x4* qp = reinterpret_cast<const u32x4*>(base + (size_t)q_tok * row_bytes + 16 * h2);
#pragma unroll
    for (int st = 0; st < 4; ++st) qf[st] = __builtin_bit_cast(bf16x8, qp[2 * st]);
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

  // ---- S^T = K Q^T over the wave's key tiles: tile t, local key 32 t + i = patch row 2 t + (i >> 4), column i & 15 -------------
  f32x16 S[NKT];
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) S[t][i] = 0.f;
  int ka[NKT];
  {
    const int kr0 = korg + (l31 >> 4) * HC + (l31 & 15);
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
      const int kr = kr0 + 2 * t * HC;
      ka[t] = kr * 128 + ((h2 ^ asw(kr)) << 4);
    }
  }
#pragma unroll
  for (int st = 0; st < 4; ++st) {
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Kimg + (ka[t] ^ (32 * st)));
      S[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[st], S[t], 0, 0, 0);
    }
  }
  {
    // every K fragment of this wave has been read (the MFMAs above consumed them): once all four waves are here the image takes the V rows,
    // which land while the softmax below runs
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    KD_BARRIER();
    stage(false, true);
  }
  // ---- window mask + softmax -------------------------------------------------------------------------------------------------------
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) S[t][i] = min3f(S[t][i], colv[i & 7], rowv[2 * t + (i >> 3)]);
  const float m = score_max<NKT>(S);
  float l = score_exp<NKT>(S, m);
  l += __shfl_xor(l, 32, 64);
  KD_WAIT_VM(0);
  KD_BARRIER();

  // ---- O^T = V^T P^T: k-slots of lane-half h2 at step (t, u) are patch row 2 t + u, columns 4 h2 + {0..3} and + 8 ----------------
  f32x16 O[2];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int i = 0; i < 16; ++i) O[e][i] = 0.f;
  const int vr0 = korg + 4 * h2 + vt_lane_row(lane);
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) pv_step(O, Vimg, vt_addr(vr0 + (2 * t + u) * HC, lane), p_frag(S[t], u));
  store_o(a.out + ((size_t)b * T + q_tok) * (a.nh * DH) + head * DH, O, 1.0f / l, h2, q_ok);
}

// (Round 6 also built a PERSISTENT form of this kernel -- a workgroup walking several tiles with the next tile's K and queries requested behind
// the scores and the current tile's V behind the barrier in front of them, two images, two workgroups per CU, V^T reads as inline asm so that
// hipcc's vmcnt(0) in front of its own ds_read_b64_tr_b16 would not drain the prefetch -- bit-identical, and SLOWER: 33.9 against 32.4 us at
// level 0, 19.0 against 19.2 at level 1, same box, interleaved.  With three workgroups per CU the one-tile form already moves its 134 MB in
// ~27 us of a 31 us launch, ~5 TB/s: the halo waits are covered, what is left is the memory system.  Removed; profiles/r06_na_core.md.)

// ---- neighbourhood core, kernel sizes 11 and 13 ------------------------------------------------------------------------------------
// The 4x8 query block of a wave now needs a patch of (4 + KS - 1) rows x (8 + KS - 1) = 18 / 20 columns: no longer a power-of-two
// width, so the patch is walked DENSELY: local key kl = 20 r + c (columns padded to 20, a multiple of 4 so that the 4-key groups of
// the V^T reads never straddle two patch rows), tiles of 32 keys, every key's (row, column) by constant division, validity per
// accumulator register from those.  One workgroup per CU (122 / 145 KiB of halo images), the whole register file per wave.
// Correctness form for the sizes no shipped config uses; the 3..9 kernel above is the tuned one.
template <int KS>
struct NaWide {
  static constexpr int HR = NA_TH + KS - 1, HC = NA_TW + KS - 1;
  static constexpr int PR = 4 + KS - 1, PC = 20;                            // patch rows; padded patch width
  static constexpr int NKT = (PR * PC + 31) / 32;
  static constexpr int ROWS = ((HR * HC + 2 * PC + 7) / 8) * 8;             // slack: padded columns and the last tile's tail poke past the halo
  static constexpr int LDS = 2 * ROWS * 128;
  static_assert(8 + KS - 1 <= PC, "patch width");
};

template <int KS>
__global__ __launch_bounds__(256, 1) void attn_na2d_wide_bf16_kernel(const NArgs a) {
  using G = NaWide<KS>;
  constexpr int HR = G::HR, HC = G::HC, PR = G::PR, PC = G::PC, NKT = G::NKT, ROWS = G::ROWS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Kimg = smem;
  char* Vimg = smem + ROWS * 128;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, h2 = lane >> 5;
  const auto warm = code_warm_begin<24576>((int)blockIdx.x < a.warm && tid < 64);
  const int wy_ = wid >> 1, wx_ = wid & 1;
  const int tiles_x = (a.W + NA_TW - 1) / NA_TW, tiles_y = (a.H + NA_TH - 1) / NA_TH;
  int r = blockIdx.x;
  const int tx = r % tiles_x; r /= tiles_x;
  const int ty = r % tiles_y; r /= tiles_y;
  const int head = r % a.nh, b = r / a.nh;
  const int T = a.H * a.W;
  const size_t row_bytes = (size_t)3 * a.nh * DH * 2;
  const char* base = reinterpret_cast<const char*>(a.qkv) + (size_t)b * T * row_bytes + head * (DH * 2);
  const int ty0 = ty * NA_TH, tx0 = tx * NA_TW;
  const int hy0 = max(0, min(ty0 - KS / 2, a.H - HR)), hx0 = max(0, min(tx0 - KS / 2, a.W - HC));
  for (int pc = wid; pc < ROWS / 8; pc += 4) {
    const int row = 8 * pc + (lane >> 3);
    const int ky = min(hy0 + row / HC, a.H - 1), kx = min(hx0 + row % HC, a.W - 1);       // rows past the halo: any real token (masked)
    const char* src = base + (size_t)(ky * a.W + kx) * row_bytes + (((lane & 7) ^ asw(row)) << 4);
    glds16(src + a.nh * DH * 2, Kimg + pc * 1024);
    glds16(src + 2 * a.nh * DH * 2, Vimg + pc * 1024);
  }
  const int qy_raw = ty0 + 4 * wy_ + (l31 >> 3), qx_raw = tx0 + 8 * wx_ + (l31 & 7);
  const bool q_ok = qy_raw < a.H && qx_raw < a.W;
  const int qy = min(qy_raw, a.H - 1), qx = min(qx_raw, a.W - 1);
  const int q_tok = qy * a.W + qx;
  bf16x8 qf[4];
  {
    const u32x4* qp = reinterpret_cast<const u32x4*>(base + (size_t)q_tok * row_bytes + 16 * h2);
#pragma unroll
    for (int st = 0; st < 4; ++st) qf[st] = __builtin_bit_cast(bf16x8, qp[2 * st]);
  }
  const int wy = max(0, min(qy - KS / 2, a.H - KS)) - hy0, wx = max(0, min(qx - KS / 2, a.W - KS)) - hx0;
  const int row_lo = min(max(0, min(min(ty0 + 4 * wy_, a.H - 1) - KS / 2, a.H - KS)) - hy0, HR - PR);
  const int col_lo = min(max(0, min(min(tx0 + 8 * wx_, a.W - 1) - KS / 2, a.W - KS)) - hx0, HC - (8 + KS - 1));
  const int korg = row_lo * HC + col_lo;
  const int r0 = wy - row_lo, c0 = wx - col_lo;
  auto img_row = [&](int kl) -> int { return korg + (kl / PC) * HC + (kl % PC); };     // LDS row of local key kl
  KD_WAIT_VM(0);
  code_warm_end(warm);
  KD_BARRIER();

  f32x16 S[NKT];
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) S[t][i] = 0.f;
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    const int kr = min(img_row(32 * t + l31), ROWS - 1);
    const int ka = kr * 128 + ((h2 ^ asw(kr)) << 4);
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Kimg + (ka ^ (32 * st)));
      S[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[st], S[t], 0, 0, 0);
    }
  }
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int kl = 32 * t + (i & 3) + 8 * (i >> 2) + 4 * h2;
      const int pr = kl / PC, pcn = kl % PC;
      const bool valid = (unsigned)(pr - r0) < (unsigned)KS && (unsigned)(pcn - c0) < (unsigned)KS && pr < PR;
      S[t][i] = valid ? S[t][i] : -INFINITY;
    }
  const float m = score_max<NKT>(S);
  float l = score_exp<NKT>(S, m);
  l += __shfl_xor(l, 32, 64);

  f32x16 O[2];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int i = 0; i < 16; ++i) O[e][i] = 0.f;
  const int c = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1), sub = (lane & 1) * 8;
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const bf16x8 pf = p_frag(S[t], u);
      // k-slots of lane-half h2: keys 32 t + 16 u + 4 h2 + {0..3} and the same + 8 -- two 4-key groups, each inside one patch row
      const int k0 = 32 * t + 16 * u + 4 * h2;
      const int ra = min(img_row(k0) + vt_lane_row(lane), ROWS - 1), rb = min(img_row(k0 + 8) + vt_lane_row(lane), ROWS - 1);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (s16x4 __attribute__((address_space(3)))*)(Vimg + ra * 128 + (((4 * e + c) ^ asw(ra)) << 4) + sub));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (s16x4 __attribute__((address_space(3)))*)(Vimg + rb * 128 + (((4 * e + c) ^ asw(rb)) << 4) + sub));
        const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2v = __builtin_bit_cast(u32x2, hi);
        O[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, u32x4{l2[0], l2[1], h2v[0], h2v[1]}), pf, O[e], 0, 0, 0);
      }
    }
  store_o(a.out + ((size_t)b * T + q_tok) * (a.nh * DH) + head * DH, O, 1.0f / l, h2, q_ok);
}

template <int KS>
static int launch_na_wide(const NArgs& a, hipStream_t s) {
  auto k = attn_na2d_wide_bf16_kernel<KS>;
  static LdsAttr set;
  set.ensure(reinterpret_cast<const void*>(k), NaWide<KS>::LDS);
  const long nb = (long)a.batch * a.nh * ((a.H + NA_TH - 1) / NA_TH) * ((a.W + NA_TW - 1) / NA_TW);
  char nm[64] = "attn_na2d_bf16";
  if (prof_on()) snprintf(nm, sizeof(nm), "attn_na2d_bf16 k%d %dx%d nh=%d", KS, a.H, a.W, a.nh);
  LaunchScope prof(nm, 4.0 * a.batch * (double)a.H * a.W * a.nh * DH * KS * KS, 8.0 * a.batch * (double)a.H * a.W * a.nh * DH, s);
  hipLaunchKernelGGL(k, dim3((unsigned)nb), dim3(256), NaWide<KS>::LDS, s, a);
  return check_launch("kd_attn_na2d_bf16");
}

template <int MODE, int NT, int QW>
static int launch_dense(const DArgs& a, long nproblems, const char* name, hipStream_t s) {
  constexpr int LDS = NT * 32 * 256, NQB = (NT + QW - 1) / QW;
  auto k = attn_dense_bf16_kernel<MODE, NT, QW>;
  static LdsAttr set;
  set.ensure(reinterpret_cast<const void*>(k), LDS);
  const int n_slots = MODE == MODE_GLOBAL ? a.T : (1 << (2 * WinLog2<MODE>::v));
  LaunchScope prof(name, 4.0 * (double)nproblems * n_slots * n_slots * DH, 2.0 * (double)a.batch * a.T * a.nh * DH * 4.0, s);
  hipLaunchKernelGGL(k, dim3((unsigned)(nproblems * NQB)), dim3(QW * 64), LDS, s, a);
  return check_launch(name);
}

template <int KS>
static int launch_na(const NArgs& a, hipStream_t s) {
  const long nb = (long)a.batch * a.nh * ((a.H + NA_TH - 1) / NA_TH) * ((a.W + NA_TW - 1) / NA_TW);
  auto k = attn_na2d_bf16_kernel<KS>;
  static LdsAttr set;
  constexpr int LDS = NaGeo<KS>::LDS;
  set.ensure(reinterpret_cast<const void*>(k), LDS);
  char nm[64] = "attn_na2d_bf16";
  if (prof_on()) snprintf(nm, sizeof(nm), "attn_na2d_bf16 k%d %dx%d nh=%d", KS, a.H, a.W, a.nh);
  LaunchScope prof(nm, 4.0 * a.batch * (double)a.H * a.W * a.nh * DH * KS * KS, 8.0 * a.batch * (double)a.H * a.W * a.nh * DH, s);
  hipLaunchKernelGGL(k, dim3((unsigned)nb), dim3(256), LDS, s, a);
  return check_launch("kd_attn_na2d_bf16");
}

}  // namespace b16
}  // namespace kd

using namespace kd;
using namespace kd::b16;

extern "C" int kd_attn_global_bf16(const void* qkv, void* out, int batch, int T, int nh, void* stream) {
  if (!qkv || !out || batch <= 0 || nh <= 0 || T <= 0) return fail(KD_EINVAL, "kd_attn_global_bf16: bad arguments");
  DArgs a{reinterpret_cast<const u16*>(qkv), reinterpret_cast<u16*>(out), batch, T, nh, 0, 0, 0, 0, option("code_warm", KD_CODE_WARM_DEFAULT)};
  hipStream_t s = (hipStream_t)stream;
  const long nb = (long)batch * nh;
  if (T > 256) {
    static LdsAttr set;
    set.ensure(reinterpret_cast<const void*>(attn_long_bf16_kernel), GL_LDS);
    const long nqb = (T + GL_QW * 32 - 1) / (GL_QW * 32);
    LaunchScope prof("attn_global_bf16", 4.0 * (double)nb * T * T * DH, 2.0 * (double)batch * T * nh * DH * 4.0, s);
    hipLaunchKernelGGL(attn_long_bf16_kernel, dim3((unsigned)(nb * nqb)), dim3(GL_QW * 64), GL_LDS, s, a);
    return check_launch("kd_attn_global_bf16");
  }
  const int qw = option("attn_global_qw", 8);
  if (T <= 32) return launch_dense<MODE_GLOBAL, 1, 1>(a, nb, "attn_global_bf16", s);
  if (T <= 64) return launch_dense<MODE_GLOBAL, 2, 2>(a, nb, "attn_global_bf16", s);
  if (T <= 128) return launch_dense<MODE_GLOBAL, 4, 4>(a, nb, "attn_global_bf16", s);
  if (qw >= 8) return launch_dense<MODE_GLOBAL, 8, 8>(a, nb, "attn_global_bf16", s);
  if (qw == 2) return launch_dense<MODE_GLOBAL, 8, 2>(a, nb, "attn_global_bf16", s);
  return launch_dense<MODE_GLOBAL, 8, 4>(a, nb, "attn_global_bf16", s);
}

extern "C" int kd_attn_window_bf16(const void* qkv, void* out, int batch, int H, int W, int nh, int ws, int shift, void* stream) {
  if (!qkv || !out || batch <= 0 || nh <= 0 || H <= 0 || W <= 0) return fail(KD_EINVAL, "kd_attn_window_bf16: bad arguments");
  if (ws != 4 && ws != 8 && ws != 16) return fail(KD_EINVAL, "kd_attn_window_bf16: window_size %d unsupported (4, 8 or 16)", ws);
  if ((H % ws) || (W % ws)) return fail(KD_EINVAL, "kd_attn_window_bf16: grid %dx%d not divisible by the window", H, W);
  if (shift < 0 || shift >= ws) return fail(KD_EINVAL, "kd_attn_window_bf16: bad shift %d", shift);
  DArgs a{reinterpret_cast<const u16*>(qkv), reinterpret_cast<u16*>(out), batch, H * W, nh, H, W, ws, shift, option("code_warm", KD_CODE_WARM_DEFAULT)};
  const long nb = (long)batch * nh * (H / ws) * (W / ws);
  hipStream_t s = (hipStream_t)stream;
  if (ws == 8) return launch_dense<MODE_WINDOW, 2, 2>(a, nb, "attn_window_bf16", s);
  if (ws == 4) return launch_dense<MODE_WINDOW4, 1, 1>(a, nb, "attn_window_bf16", s);
  return launch_dense<MODE_WINDOW16, 8, 4>(a, nb, "attn_window_bf16", s);
}

extern "C" int kd_attn_na2d_bf16(const void* qkv, void* out, int batch, int H, int W, int nh, int ks, void* stream) {
  if (!qkv || !out || batch <= 0 || nh <= 0) return fail(KD_EINVAL, "kd_attn_na2d_bf16: bad arguments");
  if (ks < 3 || ks > 13 || !(ks & 1)) return fail(KD_EINVAL, "kd_attn_na2d_bf16: kernel_size %d unsupported (3, 5, 7, 9, 11 or 13)", ks);
  if (H < ks || W < ks) return fail(KD_EINVAL, "kd_attn_na2d_bf16: grid %dx%d smaller than the %dx%d neighbourhood", H, W, ks, ks);
  NArgs a{reinterpret_cast<const u16*>(qkv), reinterpret_cast<u16*>(out), batch, H, W, nh, option("code_warm", KD_CODE_WARM_DEFAULT)};
  hipStream_t s = (hipStream_t)stream;
  switch (ks) {
    case 3: return launch_na<3>(a, s);
    case 5: return launch_na<5>(a, s);
    case 7: return launch_na<7>(a, s);
    case 9: return launch_na<9>(a, s);
    case 11: return launch_na_wide<11>(a, s);
    default: return launch_na_wide<13>(a, s);
  }
}

KD_TEXT_PAD(attn_bf16)      // last function of this code object: kd_common.h, code warm-up
