// Attention cores of the HDiT denoiser for gfx950, fp32 parity mode.
//
//   kd_qk_prep_f32     cosine-sim scaling + axial RoPE on q,k in place (stand-alone pass)
//   kd_attn_global_f32 dense softmax attention per (sample, head), T <= 256 tokens
//   kd_attn_window_f32 shifted-window attention, 8x8 windows (roll / window / mask / unwindow
//                      folded into index arithmetic; the boolean mask is never materialised)
//   kd_attn_na2d_f32   7x7 neighbourhood attention with clamped windows
//
// All three cores read q,k,v straight out of the qkv GEMM output [tokens, 3, nh, 64] and can apply
// the q/k preparation on the fly (prep != 0), so neither a rearranged copy nor a prepared qkv ever
// round-trips through HBM.
//
// Dense cores (global / window) run on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32):
//   S^T = K Q^T  ("swapped" product: a lane ends up with 16 keys x ONE query, so the softmax row
//                 reductions are in-lane plus one cross-half shuffle)
//   O^T = V^T P^T (again swapped: the P registers feed the B operand unchanged, and the row
//                 normaliser stays a per-lane scalar)
// K and V tiles are staged once per (sample, head[, window]) in LDS with rows padded to 68 floats
// (conflict-free ds_read_b128 for K fragments, ds_read_b32 for V^T fragments).
//
// The neighbourhood core (and, with KD_PREC_SPLIT3, the global / window cores) run on the bf16 MFMA with the 3-term split
// of every fp32 operand (hi*hi + hi*lo + lo*hi, fp32 accumulate): see attn_na2d_kernel / attn_global_split_kernel below.
#include "kd_common.h"
#include <cstdlib>
#include <type_traits>

namespace kd {

constexpr int DH = 64;          // head dim (fixed: every config, k_diffusion/config.py:135-136)
constexpr int LDS_ROW = DH + 4; // padded LDS row (floats)
constexpr int ROT = 16;         // rotary angles per head: dims [0,16) pair with [16,32)

__global__ __launch_bounds__(256) void qk_prep_kernel(float* qkv, const float* scale_h, const float* cos_t, const float* sin_t,
                                                      long rows_total, int tokens_per_sample, int nh, float eps) {
  // one 16-lane group per (token, t in {q,k}, head) row
  const long g = ((long)blockIdx.x * 256 + threadIdx.x) >> 4;
  const int c = threadIdx.x & 15;
  if (g >= rows_total) return;   // whole 16-lane groups exit together
  const int head = g % nh;
  const long r2 = g / nh;
  const int t = r2 & 1;
  const long tok = r2 >> 1;
  float* row = qkv + (tok * 3 + t) * (long)(nh * DH) + head * DH;
  const int tl = tok % tokens_per_sample;
  const float* cs = cos_t + ((long)tl * nh + head) * ROT;
  const float* sn = sin_t + ((long)tl * nh + head) * ROT;
  f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * c);
  v = prep_row16(v, c, sqrtf(scale_h[head]), cs, sn, eps);
  *reinterpret_cast<f32x4*>(row + 4 * c) = v;
}

// ---- dense cores ------------------------------------------------------------------------------------
// window modes carry log2(window_size) in the template value: 8x8 (the shipped config), 4x4, 16x16 windows
enum { MODE_GLOBAL = 0, MODE_WINDOW = 1, MODE_WINDOW4 = 2, MODE_WINDOW16 = 3 };
template <int MODE> struct WinLog2 { static constexpr int v = MODE == MODE_WINDOW ? 3 : (MODE == MODE_WINDOW4 ? 2 : 4); };

struct DenseArgs {
  const float* qkv; float* out;
  const float* scale_h; const float* cos_t; const float* sin_t;
  int batch, T, nh;          // T = tokens per sample
  int H, W, ws, shift;       // window mode
  float eps;
  int warm;                  // code warm-up workgroups (kd_common.h)
};

// slot -> token index inside the sample (or -1), for the problem owned by this block
template <int MODE>
__device__ __forceinline__ int slot_token(const DenseArgs& a, int slot, int wi, int wj) {
  if (MODE == MODE_GLOBAL) return slot < a.T ? slot : -1;
  constexpr int L = WinLog2<MODE>::v, WS = 1 << L;
  const int ai = slot >> L, bj = slot & (WS - 1);
  int i = wi * WS + ai - a.shift; if (i < 0) i += a.H;     // rolled[i] = orig[(i - shift) mod H]  (:274)
  int j = wj * WS + bj - a.shift; if (j < 0) j += a.W;
  return i * a.W + j;
}
// wrapped-region id of a window slot (make_shifted_window_masks, :285-316)
template <int MODE>
__device__ __forceinline__ int slot_region(int slot, int wi, int wj, int shift) {
  constexpr int L = WinLog2<MODE>::v, WS = 1 << L;
  return ((wi == 0 && (slot >> L) < shift) ? 2 : 0) + ((wj == 0 && (slot & (WS - 1)) < shift) ? 1 : 0);
}

template <int MODE, int MAXT, bool PREP>
__global__ __launch_bounds__(MAXT * 64) void attn_dense_kernel(const DenseArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const auto warm = code_warm_begin<16384>((int)blockIdx.x < a.warm && blockIdx.y == 0 && threadIdx.x < 64);   // kd_common.h: this kernel's code -> L2
  constexpr int TP = MAXT * 32;
  float* Ks = smem;                 // [TP][LDS_ROW]
  float* Vs = smem + TP * LDS_ROW;  // [TP][LDS_ROW]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  int b, head, wi = 0, wj = 0;
  if (MODE == MODE_GLOBAL) {
    head = blockIdx.x % a.nh; b = blockIdx.x / a.nh;
  } else {
    const int nww = a.W >> WinLog2<MODE>::v, nwh = a.H >> WinLog2<MODE>::v;
    int r = blockIdx.x;
    wj = r % nww; r /= nww; wi = r % nwh; r /= nwh; head = r % a.nh; b = r / a.nh;
  }
  const int n_slots = (MODE == MODE_GLOBAL) ? a.T : (1 << (2 * WinLog2<MODE>::v));
  const int ntiles = (n_slots + 31) >> 5;
  const long row_stride = 3L * a.nh * DH;                       // floats between consecutive tokens
  const float* base = a.qkv + (long)b * a.T * row_stride + head * DH;
  const float sqrt_scale = PREP ? sqrtf(a.scale_h[head]) : 1.f;

  // ---- stage K (prepared) and V into LDS: 16 lanes per row -----------------------------------
  {
    const int c = tid & 15;
    for (int slot = tid >> 4; slot < ntiles * 32; slot += MAXT * 4) {
      const int tok = slot < n_slots ? slot_token<MODE>(a, slot, wi, wj) : -1;
      f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
      if (tok >= 0) {
        const float* rp = base + (long)tok * row_stride;
        kv = *reinterpret_cast<const f32x4*>(rp + a.nh * DH + 4 * c);
        vv = *reinterpret_cast<const f32x4*>(rp + 2 * a.nh * DH + 4 * c);
      }
      if (PREP) {
        const int tk = tok >= 0 ? tok : 0;
        const float* cs = a.cos_t + ((long)tk * a.nh + head) * ROT;
        kv = prep_row16(kv, c, sqrt_scale, cs, a.sin_t + ((long)tk * a.nh + head) * ROT, a.eps);
      }
      *reinterpret_cast<f32x4*>(Ks + slot * LDS_ROW + 4 * c) = kv;
      *reinterpret_cast<f32x4*>(Vs + slot * LDS_ROW + 4 * c) = vv;
    }
  }

  // ---- this wave's 32 queries as the B operand: lane holds Q[q = lane&31][8c + 4h + 0..3] ---------
  const int h2 = lane >> 5;
  const int q_slot = wid * 32 + (lane & 31);
  const bool wave_active = wid < ntiles;
  const int q_tok = (wave_active && q_slot < n_slots) ? slot_token<MODE>(a, q_slot, wi, wj) : -1;
  f32x4 q[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) q[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (q_tok >= 0) {
    const float* rp = base + (long)q_tok * row_stride;
#pragma unroll
    for (int c = 0; c < 8; ++c) q[c] = *reinterpret_cast<const f32x4*>(rp + 8 * c + 4 * h2);
  }
  if (PREP) {
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) ss += q[c][0] * q[c][0] + q[c][1] * q[c][1] + q[c][2] * q[c][2] + q[c][3] * q[c][3];
    ss += __shfl_xor(ss, 32, 64);
    const float f = sqrt_scale * rsqrtf(ss + a.eps);
#pragma unroll
    for (int c = 0; c < 8; ++c) q[c] = q[c] * f;
    // rotary pairs (d, d+16) for d < 16 sit in chunks (c, c+2), c in {0,1}, of the SAME lane
    const int tk = q_tok >= 0 ? q_tok : 0;
    const float* cs = a.cos_t + ((long)tk * a.nh + head) * ROT;
    const float* sn = a.sin_t + ((long)tk * a.nh + head) * ROT;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const f32x4 cc = *reinterpret_cast<const f32x4*>(cs + 8 * c + 4 * h2);
      const f32x4 sc = *reinterpret_cast<const f32x4*>(sn + 8 * c + 4 * h2);
      const f32x4 x1 = q[c], x2 = q[c + 2];
      q[c] = x1 * cc - x2 * sc;
      q[c + 2] = x2 * cc + x1 * sc;
    }
  }
  code_warm_end(warm);
  __syncthreads();
  if (!wave_active) return;

  // ---- S^T[key][query] = K Q^T ---------------------------------------------------------------
  f32x16 S[MAXT];
#pragma unroll
  for (int t = 0; t < MAXT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) S[t][r] = 0.f;
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    if (t < ntiles) {
      const float* kp = Ks + (t * 32 + (lane & 31)) * LDS_ROW + 4 * h2;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const f32x4 kf = *reinterpret_cast<const f32x4*>(kp + 8 * c);
#pragma unroll
        for (int s = 0; s < 4; ++s) S[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s], q[c][s], S[t], 0, 0, 0);
      }
    }
  }

  // ---- mask + softmax over keys (per query = per lane&31; keys spread over regs, tiles, halves) ---
  // The mask enters as an additive bias (0 / -inf) that is recomputed in both passes instead of a
  // select written back into S: hipcc (ROCm 7.2) miscompiles `S[t][r] = ok ? S[t][r] : -inf` on an
  // MFMA accumulator (it overwrites element 0's AGPR with -inf before the conditional copy).
  const int q_region = (MODE != MODE_GLOBAL) ? slot_region<MODE>(q_slot, wi, wj, a.shift) : 0;
  auto key_bias = [&](int t, int r) -> float {
    const int ks = t * 32 + mfma32_row(r, lane);
    bool ok = ks < n_slots;
    if (MODE != MODE_GLOBAL && a.shift) ok = ok && (slot_region<MODE>(ks, wi, wj, a.shift) == q_region);
    return ok ? 0.f : -INFINITY;
  };
  float m = -INFINITY;
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    if (t < ntiles) {
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, S[t][r] + key_bias(t, r));
    }
  }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float l = 0.f;
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    if (t < ntiles) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = expf((S[t][r] + key_bias(t, r)) - m);
        S[t][r] = p;
        l += p;
      }
    }
  }
  l += __shfl_xor(l, 32, 64);

  // ---- O^T[e][query] = V^T P^T ------------------------------------------------------------------
  f32x16 O[2];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[e][r] = 0.f;
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    if (t < ntiles) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float* vp = Vs + (t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2) * LDS_ROW + (lane & 31);
        O[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[0], S[t][r], O[0], 0, 0, 0);
        O[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[32], S[t][r], O[1], 0, 0, 0);
      }
    }
  }

  if (q_tok >= 0) {
    const float inv = 1.0f / l;
    float* op = a.out + ((long)b * a.T + q_tok) * (a.nh * DH) + head * DH;
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v = {O[e][4 * g] * inv, O[e][4 * g + 1] * inv, O[e][4 * g + 2] * inv, O[e][4 * g + 3] * inv};
        *reinterpret_cast<f32x4*>(op + e * 32 + 8 * g + 4 * h2) = v;
      }
  }
}

// ---- neighbourhood core -----------------------------------------------------------------------------
// One 256-thread workgroup (4 wave64s) per (sample, head, 8x16 query tile).  The 14x22 key halo of the tile is
// staged in LDS ONCE, K first (cosine-sim scale + RoPE applied on the way in when prep != 0), then V in the same
// buffer, both as split bf16 (hi = bf16(x), lo = bf16(x - hi)): the two products of the tile then run on the bf16
// MFMA with the 3-term split (hi*hi + hi*lo + lo*hi, fp32 accumulate; per-product error <= ~2^-15, like the GEMMs).
//   wave (wy, wx) owns the 4x8 query block at rows 4wy.., columns 8wx.. (32 queries = the MFMA's 32 columns); the
//   clamped 7x7 windows of its queries lie inside a 10-row x 16-column patch of the halo, walked as 160 "local keys"
//   kl = 16*r + c  ->  5 key tiles of 32
//   S^T[key][query] = K Q^T : K rows are the A operand (8 consecutive head dims per lane, 16-byte LDS reads from
//                     128-byte rows, chunk index XOR-swizzled with (row>>1)&7), Q the B operand from registers
//   mask            : 5 bit-words per lane (7 runs of 7 bits), applied as an additive 0 / -inf bias; softmax in
//                     registers: a lane owns ONE query (column) and 16 keys per tile, so the row reductions are
//                     in-lane plus one cross-half shuffle
//   O^T[e][query]   = V^T P^T: the P registers feed the B operand directly (an MFMA contracts over 16 keys in the
//                     k-slot order [4h..4h+3, 8+4h..8+4h+3] of lane-half h; any order is valid as long as A uses
//                     the same one; 4-key groups never straddle a patch row because rows are 16 keys wide), so V is
//                     staged TRANSPOSED ([e][key], 636-byte rows: conflict-free dword-pair reads)
// LDS: 81 408 B per workgroup -> two workgroups (8 waves) per CU.  The V halo is requested right after the QK^T
// MFMAs so that its latency hides behind the softmax.  Against a dense tiling the masked keys waste ~70% of the matrix
// work, but at the bf16 rate that is ~15 us per launch at the largest level; the kernel is bound by staging instead.
struct NaArgs {
  const float* qkv; float* out;
  const float* scale_h; const float* cos_t; const float* sin_t;
  int batch, H, W, nh;
  float eps;
  int warm;                  // code warm-up workgroups (kd_common.h)
};

constexpr int NA_K = 7, NA_TH = 8, NA_TW = 16;
constexpr int NA_HR = NA_TH + NA_K - 1, NA_HC = NA_TW + NA_K - 1;    // 14 x 22 halo
constexpr int NA_KEYS = NA_HR * NA_HC;                               // 308
constexpr int NA_KROWS = 310;                                        // K image rows (a patch may poke 2 keys past the halo)
constexpr int NA_KT = 5;                                             // key tiles per wave
constexpr int NA_STAGE_IT = (NA_KROWS + 15) / 16;                    // 20 staging rounds of 16 rows (16 lanes per row)
constexpr int NA_VT_STRIDE = 636;                                    // bytes per e-row of the transposed V image (318 keys; 159 dwords: odd, so the 32 e-rows of a fragment read hit 32 different banks)
constexpr int NA_IMG_K = NA_KROWS * 128;                             // bytes of one K image  [310][64] bf16
constexpr int NA_IMG_V = DH * NA_VT_STRIDE;                          // bytes of one V^T image [64][316] bf16
constexpr int NA_LDS = 2 * (NA_IMG_K > NA_IMG_V ? NA_IMG_K : NA_IMG_V);

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ unsigned pack2_bf16(float a, float b) {
  using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
  bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}
// 4 floats -> (hi, lo) bf16 quads.  The residuals are formed with packed fp32 arithmetic (v_pk_add_f32: two lanes' worth
// per issue): these conversions are the bulk of the VALU work of the staging phases.
__device__ __forceinline__ void split4_bf16(const f32x4 v, u32x2& hi, u32x2& lo) {
  using f32x2 = __attribute__((ext_vector_type(2))) float;
  hi[0] = pack2_bf16(v[0], v[1]);
  hi[1] = pack2_bf16(v[2], v[3]);
  const f32x2 h01 = {__uint_as_float(hi[0] << 16), __uint_as_float(hi[0] & 0xFFFF0000u)};
  const f32x2 h23 = {__uint_as_float(hi[1] << 16), __uint_as_float(hi[1] & 0xFFFF0000u)};
  const f32x2 r01 = f32x2{v[0], v[1]} - h01, r23 = f32x2{v[2], v[3]} - h23;
  lo[0] = pack2_bf16(r01[0], r01[1]);
  lo[1] = pack2_bf16(r23[0], r23[1]);
}
// q / k / v chunks as the consumers want them: either split here (fp32 input) or taken as stored (PK: the qkv GEMM already
// wrote every 4-dim chunk as [hi: 4 x bf16][lo: 4 x bf16] in the 16 bytes of its 4 floats -- kd_gemm_f32 qkv_packed)
template <bool PK>
__device__ __forceinline__ void split_in(const f32x4 v, u32x2& hi, u32x2& lo) {
  if (PK) {
    hi = u32x2{__float_as_uint(v[0]), __float_as_uint(v[1])};
    lo = u32x2{__float_as_uint(v[2]), __float_as_uint(v[3])};
  } else {
    split4_bf16(v, hi, lo);
  }
}
// byte offset of (row, 16-byte chunk c of 8) in a [rows][64] bf16 image with 128-byte rows
__device__ __forceinline__ int na_kswz(int row, int c) { return row * 128 + ((c ^ ((row >> 1) & 7)) << 4); }

// FULL: the image is at least as large as the halo (H >= 14, W >= 22), so every halo key lies inside the image and only
// the pad rows past key 307 need zeroing: no per-key bounds tests / selects in the staging loops.
template <int PMODE, bool FULL>      // PMODE 0: q, k prepared (fp32)  1: raw fp32, prepared here  2: prepared and stored split (packed)
__global__ __launch_bounds__(256, 2) void attn_na2d_kernel(const NaArgs a) {
  constexpr bool PREP = PMODE == 1, PK = PMODE == 2;
  extern __shared__ __attribute__((aligned(16))) char na_smem[];
  const auto warm = code_warm_begin<32768>((int)blockIdx.x < a.warm && blockIdx.y == 0 && threadIdx.x < 64);   // kd_common.h: this kernel's code -> L2
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, h2 = lane >> 5;
  const int wy_ = wid >> 1, wx_ = wid & 1;                            // this wave's 4x8 query block
  const int tiles_x = (a.W + NA_TW - 1) / NA_TW, tiles_y = (a.H + NA_TH - 1) / NA_TH;
  // XCD-aware tile order: consecutive workgroups are dealt round-robin over the 8 XCDs (private L2s); give every XCD a
  // CONTIGUOUS run of tiles so that the halos of neighbouring tiles (2.4x re-read of K and V) hit ONE L2 instead of
  // being fetched from HBM by eight of them.  Bijective for any grid size; placement never affects correctness.
  int r;
  {
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int q = nwg >> 3, rem = nwg & 7;
    r = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + k;
  }
  const int tx = r % tiles_x; r /= tiles_x;
  const int ty = r % tiles_y; r /= tiles_y;
  const int head = r % a.nh; const int b = r / a.nh;
  const int T = a.H * a.W;
  const long row_stride = 3L * a.nh * DH;
  const float* base = a.qkv + (long)b * T * row_stride + head * DH;
  const float sqrt_scale = PREP ? sqrtf(a.scale_h[head]) : 1.f;
  const int ty0 = ty * NA_TH, tx0 = tx * NA_TW;
  const int hy0 = clampi(ty0 - NA_K / 2, 0, max(0, a.H - NA_HR));
  const int hx0 = clampi(tx0 - NA_K / 2, 0, max(0, a.W - NA_HC));

  // ---- halo loads: 16 lanes per key row, 16 rows per round --------------------------------------------------------
  const int c16 = tid & 15, rsub = tid >> 4;
  // round `it` of a staging pass handles halo key hr = 16*it + rsub; (ky, kx) advance incrementally (16 < 22 columns:
  // at most one row wrap per round) instead of a divide per round
  struct HaloIt {
    int ky, kx, hr;
    __device__ __forceinline__ void next() { hr += 16; kx += 16; if (kx >= NA_HC) { kx -= NA_HC; ++ky; } }
  };
  const HaloIt halo0{rsub / NA_HC, rsub % NA_HC, rsub};       // rsub < 16 < 22: row 0
  auto halo_tok = [&](const HaloIt& h) -> int {       // token index of the key, -1 outside the image / halo
    const int ky = hy0 + h.ky, kx = hx0 + h.kx;
    if (FULL) return h.hr < NA_KEYS ? ky * a.W + kx : -1;
    return (h.hr < NA_KEYS && ky < a.H && kx < a.W) ? ky * a.W + kx : -1;
  };
  // byte offset of a token's row inside this (sample, head) slice: 32-bit (a sample's qkv is far below 2 GiB), so the
  // loads are "scalar base + 32-bit lane offset" instead of 64-bit pointer arithmetic per load
  const unsigned row_bytes = (unsigned)(row_stride * sizeof(float));
  const char* kbase_c = reinterpret_cast<const char*>(base + a.nh * DH + 4 * c16);
  const char* vbase_c = reinterpret_cast<const char*>(base + 2 * a.nh * DH + 4 * c16);
  // ---- this lane's query (column l31 of its wave): 32 of its 64 dims, 8-wide chunks 2*step + h2 ----------------
  const int qy_raw = ty0 + 4 * wy_ + (l31 >> 3), qx_raw = tx0 + 8 * wx_ + (l31 & 7);
  const bool q_ok = qy_raw < a.H && qx_raw < a.W;
  const int qy = min(qy_raw, a.H - 1), qx = min(qx_raw, a.W - 1);     // overhanging lanes shadow a real query, never store
  const int q_tok = qy * a.W + qx;
  f32x4 qf[8];
  {
    const float* rp = base + (long)q_tok * row_stride;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      qf[2 * st] = *reinterpret_cast<const f32x4*>(rp + 16 * st + 8 * h2);
      qf[2 * st + 1] = *reinterpret_cast<const f32x4*>(rp + 16 * st + 8 * h2 + 4);
    }
  }

  // ---- K -> LDS: prepare (scale_for_cosine_sim + RoPE), split, swizzled row-major images ------------------------
  // two batches of rounds, each requested in full before it is consumed (memory-level parallelism vs registers)
  char* Khi = na_smem;
  char* Klo = na_smem + NA_IMG_K;
  HaloIt hk_load = halo0, hk_use = halo0;
  auto stage_k = [&](auto n_tag) {
    constexpr int N = decltype(n_tag)::value;
    f32x4 kreg[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int tok = halo_tok(hk_load);
      kreg[i] = *reinterpret_cast<const f32x4*>(kbase_c + (unsigned)max(tok, 0) * row_bytes);
      hk_load.next();
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int tok = halo_tok(hk_use);
      const int hr = hk_use.hr;
      f32x4 v = kreg[i];
      if (tok < 0) v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (PREP) {
        const int tk = tok < 0 ? 0 : tok;
        v = prep_row16(v, c16, sqrt_scale, a.cos_t + ((long)tk * a.nh + head) * ROT, a.sin_t + ((long)tk * a.nh + head) * ROT, a.eps);
      }
      if (hr < NA_KROWS) {
        u32x2 hi, lo;
        split_in<PK>(v, hi, lo);
        const int o = na_kswz(hr, c16 >> 1) + (c16 & 1) * 8;
        *reinterpret_cast<u32x2*>(Khi + o) = hi;
        *reinterpret_cast<u32x2*>(Klo + o) = lo;
      }
      hk_use.next();
    }
  };
  stage_k(std::integral_constant<int, 10>{});
  stage_k(std::integral_constant<int, NA_STAGE_IT - 10>{});

  // ---- q preparation + split into B-operand fragments --------------------------------------------------------------
  if (PREP) {
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) ss += qf[i][0] * qf[i][0] + qf[i][1] * qf[i][1] + qf[i][2] * qf[i][2] + qf[i][3] * qf[i][3];
    ss += __shfl_xor(ss, 32, 64);
    const float f = sqrt_scale * rsqrtf(ss + a.eps);
#pragma unroll
    for (int i = 0; i < 8; ++i) qf[i] = qf[i] * f;
    // rotary pairs (d, d+16), d < 16: step 0 holds dims 8h2..8h2+7, step 1 holds 16+8h2..: both in this lane
    const float* cs = a.cos_t + ((long)q_tok * a.nh + head) * ROT + 8 * h2;
    const float* sn = a.sin_t + ((long)q_tok * a.nh + head) * ROT + 8 * h2;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const f32x4 cc = *reinterpret_cast<const f32x4*>(cs + 4 * u), sc = *reinterpret_cast<const f32x4*>(sn + 4 * u);
      const f32x4 x1 = qf[u], x2 = qf[2 + u];
      qf[u] = x1 * cc - x2 * sc;
      qf[2 + u] = x2 * cc + x1 * sc;
    }
  }
  bf16x8 qh[4], ql[4];
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    u32x2 h0, l0, h1, l1;
    split_in<PK>(qf[2 * st], h0, l0);
    split_in<PK>(qf[2 * st + 1], h1, l1);
    qh[st] = __builtin_bit_cast(bf16x8, u32x4{h0[0], h0[1], h1[0], h1[1]});
    ql[st] = __builtin_bit_cast(bf16x8, u32x4{l0[0], l0[1], l1[0], l1[1]});
  }

  // clamped window start (NATTEN semantics, dilation 1), relative to the halo origin; origin of this wave's 10x16 key
  // patch (at the bottom / right border the halo is pulled in, so the patch is clamped into it; the column origin is
  // rounded down to an even key so that the 8-byte V^T reads stay 4-byte aligned)
  const int wy = clampi(qy - NA_K / 2, 0, a.H - NA_K) - hy0;
  const int wx = clampi(qx - NA_K / 2, 0, a.W - NA_K) - hx0;
  const int row_lo = min(clampi(min(ty0 + 4 * wy_, a.H - 1) - NA_K / 2, 0, a.H - NA_K) - hy0, NA_HR - 10);
  const int col_lo = min(clampi(min(tx0 + 8 * wx_, a.W - 1) - NA_K / 2, 0, a.W - NA_K) - hx0, NA_HC - 14) & ~1;
  const int korg = row_lo * NA_HC + col_lo;      // halo index of the patch's key (0, 0); local key 16r + c is korg + 22r + c
  code_warm_end(warm);
  __syncthreads();

  // ---- S^T = K Q^T over the wave's 5 key tiles ----------------------------------------------------------------------
  f32x16 S[NA_KT];
#pragma unroll
  for (int t = 0; t < NA_KT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) S[t][i] = 0.f;
  // step-major, term-major: consecutive MFMAs go to different key tiles (no back-to-back chain on one accumulator)
  const int krow0 = korg + (l31 >> 4) * NA_HC + (l31 & 15);          // this lane's K row of tile 0 (tile t: + 2*22*t)
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    bf16x8 kh[NA_KT], kl[NA_KT];
#pragma unroll
    for (int t = 0; t < NA_KT; ++t) {
      const int o = na_kswz(krow0 + 2 * NA_HC * t, 2 * st + h2);
      kh[t] = *reinterpret_cast<const bf16x8*>(Khi + o);
      kl[t] = *reinterpret_cast<const bf16x8*>(Klo + o);
    }
#pragma unroll
    for (int t = 0; t < NA_KT; ++t) S[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl[t], qh[st], S[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NA_KT; ++t) S[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh[t], ql[st], S[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NA_KT; ++t) S[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh[t], qh[st], S[t], 0, 0, 0);
  }

  // ---- V halo prefetch: the first half of the rounds is requested now and consumed after the softmax (its latency
  // hides behind the mask / exp work); the second half is requested once those registers are free again ------------
  // V staging walks key PAIRS (2p, 2p+1: same halo row, 22 is even): round j handles pair 16j + rsub, 10 rounds
  constexpr int NA_VR = (NA_VT_STRIDE / 2 / 2 + 15) / 16, NA_VH = NA_VR / 2;      // 10 rounds, 5 per half
  struct PairIt {
    int ky, kx, key;                       // halo coordinates / index of the pair's even key
    __device__ __forceinline__ void next() { key += 32; kx += 10; ++ky; if (kx >= NA_HC) { kx -= NA_HC; ++ky; } }
  };
  const PairIt pair0{(2 * rsub) / NA_HC, (2 * rsub) % NA_HC, 2 * rsub};
  auto pair_tok = [&](const PairIt& h) -> int {      // token of the even key; the odd key is the next token (or invalid)
    const int ky = hy0 + h.ky, kx = hx0 + h.kx;
    if (FULL) return h.key < NA_KEYS ? ky * a.W + kx : -1;
    return (h.key < NA_KEYS && ky < a.H && kx < a.W) ? ky * a.W + kx : -1;
  };
  auto pair_odd_ok = [&](const PairIt& h) -> bool { return FULL || hx0 + h.kx + 1 < a.W; };
  f32x4 vreg[2 * NA_VH];
  PairIt hv_load = pair0, hv_use = pair0;
#pragma unroll
  for (int j = 0; j < NA_VH; ++j) {
    const int tok = pair_tok(hv_load);
    const unsigned o = (unsigned)max(tok, 0) * row_bytes;
    vreg[2 * j] = *reinterpret_cast<const f32x4*>(vbase_c + o);
    vreg[2 * j + 1] = *reinterpret_cast<const f32x4*>(vbase_c + o + (tok >= 0 && pair_odd_ok(hv_load) ? row_bytes : 0u));
    hv_load.next();
  }

  // ---- window mask + softmax over keys (the mask enters as an additive 0 / -inf bias) ----
  // validity of the wave's 160 local keys for THIS lane's query as 5 x 32-bit words (word t = patch rows 2t, 2t+1):
  // the window is 7 runs of 7 consecutive keys, one per patch row wy-row_lo .. +6, starting at column wx-col_lo
  unsigned vw[NA_KT];
#pragma unroll
  for (int t = 0; t < NA_KT; ++t) vw[t] = 0u;
  {
    const int r0 = wy - row_lo, c0 = wx - col_lo;
#pragma unroll
    for (int rr = 0; rr < NA_K; ++rr) {
      const int pr = r0 + rr;
      const unsigned run = 0x7Fu << ((pr & 1) * 16 + c0);
#pragma unroll
      for (int t = 0; t < NA_KT; ++t) vw[t] |= ((pr >> 1) == t) ? run : 0u;
    }
#pragma unroll
    for (int t = 0; t < NA_KT; ++t) vw[t] >>= 4 * h2;         // accumulator element i of this lane is key (i&3) + 8*(i>>2) + 4*h2
  }
  auto key_bias = [&](int t, int i) -> float {
    return (vw[t] & (1u << ((i & 3) + 8 * (i >> 2)))) ? 0.f : -INFINITY;
  };
  float m = -INFINITY;
#pragma unroll
  for (int t = 0; t < NA_KT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      S[t][i] += key_bias(t, i);          // additive 0 / -inf mask, folded in once
      m = fmaxf(m, S[t][i]);
    }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float l = 0.f;
#pragma unroll
  for (int t = 0; t < NA_KT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float pv = __expf(S[t][i] - m);
      S[t][i] = pv;
      l += pv;
    }
  l += __shfl_xor(l, 32, 64);

  // ---- V -> LDS, transposed: Vt[e][key], split --------------------------------------------------------------------------
  __syncthreads();                       // every wave is done reading K
  char* Vhi = na_smem;
  char* Vlo = na_smem + NA_IMG_V;
  // one pair -> 4 dwords (hi) + 4 dwords (lo): Vt[e][2p], Vt[e][2p+1] packed in one 32-bit LDS word per e
  auto stage_v = [&](f32x4 v0, f32x4 v1) {
    const int tok = pair_tok(hv_use);
    if (hv_use.key < NA_VT_STRIDE / 2) {           // keys 308..315 (read by patches at the halo's corner, always masked): zero
      if (tok < 0) v0 = f32x4{0.f, 0.f, 0.f, 0.f};
      if (tok < 0 || !pair_odd_ok(hv_use)) v1 = f32x4{0.f, 0.f, 0.f, 0.f};
      u32x2 h0, l0, h1, l1;
      split_in<PK>(v0, h0, l0);
      split_in<PK>(v1, h1, l1);
      char* ph_ = Vhi + (4 * c16) * NA_VT_STRIDE + hv_use.key * 2;
      char* pl_ = Vlo + (4 * c16) * NA_VT_STRIDE + hv_use.key * 2;
      // e = 4c16+0: low halves, +1: high halves of word 0; +2, +3: word 1   (v_perm_b32 selects)
      *reinterpret_cast<unsigned*>(ph_) = __builtin_amdgcn_perm(h1[0], h0[0], 0x05040100u);
      *reinterpret_cast<unsigned*>(ph_ + NA_VT_STRIDE) = __builtin_amdgcn_perm(h1[0], h0[0], 0x07060302u);
      *reinterpret_cast<unsigned*>(ph_ + 2 * NA_VT_STRIDE) = __builtin_amdgcn_perm(h1[1], h0[1], 0x05040100u);
      *reinterpret_cast<unsigned*>(ph_ + 3 * NA_VT_STRIDE) = __builtin_amdgcn_perm(h1[1], h0[1], 0x07060302u);
      *reinterpret_cast<unsigned*>(pl_) = __builtin_amdgcn_perm(l1[0], l0[0], 0x05040100u);
      *reinterpret_cast<unsigned*>(pl_ + NA_VT_STRIDE) = __builtin_amdgcn_perm(l1[0], l0[0], 0x07060302u);
      *reinterpret_cast<unsigned*>(pl_ + 2 * NA_VT_STRIDE) = __builtin_amdgcn_perm(l1[1], l0[1], 0x05040100u);
      *reinterpret_cast<unsigned*>(pl_ + 3 * NA_VT_STRIDE) = __builtin_amdgcn_perm(l1[1], l0[1], 0x07060302u);
    }
    hv_use.next();
  };
  {
    f32x4 vreg2[2 * (NA_VR - NA_VH)];
#pragma unroll
    for (int j = 0; j < NA_VR - NA_VH; ++j) {
      const int tok = pair_tok(hv_load);
      const unsigned o = (unsigned)max(tok, 0) * row_bytes;
      vreg2[2 * j] = *reinterpret_cast<const f32x4*>(vbase_c + o);
      vreg2[2 * j + 1] = *reinterpret_cast<const f32x4*>(vbase_c + o + (tok >= 0 && pair_odd_ok(hv_load) ? row_bytes : 0u));
      hv_load.next();
    }
#pragma unroll
    for (int j = 0; j < NA_VH; ++j) stage_v(vreg[2 * j], vreg[2 * j + 1]);
#pragma unroll
    for (int j = 0; j < NA_VR - NA_VH; ++j) stage_v(vreg2[2 * j], vreg2[2 * j + 1]);
  }
  __syncthreads();

  // ---- O^T = V^T P^T ------------------------------------------------------------------------------------------------------
  // four accumulators (head-dim half e x key-chunk parity u, summed at the end), issued term-major so that no MFMA
  // follows another on the same accumulator
  f32x16 O[2][2];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 16; ++i) O[e][u][i] = 0.f;
#pragma unroll
  for (int t = 0; t < NA_KT; ++t) {
    bf16x8 ph[2], pl[2], vh[2][2], vl[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      // B operand: this lane's 8 probabilities of k-slots (regs 8u..8u+7 = local keys 32t+16u+4h2+{0..3} and +8):
      // patch row 2t+u, columns 4h2.. and 8+4h2..
      u32x2 ph0, pl0, ph1, pl1;
      split4_bf16(f32x4{S[t][8 * u], S[t][8 * u + 1], S[t][8 * u + 2], S[t][8 * u + 3]}, ph0, pl0);
      split4_bf16(f32x4{S[t][8 * u + 4], S[t][8 * u + 5], S[t][8 * u + 6], S[t][8 * u + 7]}, ph1, pl1);
      ph[u] = __builtin_bit_cast(bf16x8, u32x4{ph0[0], ph0[1], ph1[0], ph1[1]});
      pl[u] = __builtin_bit_cast(bf16x8, u32x4{pl0[0], pl0[1], pl1[0], pl1[1]});
      const int key0 = korg + (2 * t + u) * NA_HC + 4 * h2;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int o = (32 * e + l31) * NA_VT_STRIDE + key0 * 2;
        // 4-byte aligned only (the patch origin is an even key): dword reads (ds_read2_b32), never a misaligned b64
        const unsigned* hp = reinterpret_cast<const unsigned*>(Vhi + o);
        const unsigned* lp = reinterpret_cast<const unsigned*>(Vlo + o);
        vh[e][u] = __builtin_bit_cast(bf16x8, u32x4{hp[0], hp[1], hp[4], hp[5]});
        vl[e][u] = __builtin_bit_cast(bf16x8, u32x4{lp[0], lp[1], lp[4], lp[5]});
      }
    }
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int u = 0; u < 2; ++u) O[e][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl[e][u], ph[u], O[e][u], 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int u = 0; u < 2; ++u) O[e][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh[e][u], pl[u], O[e][u], 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int u = 0; u < 2; ++u) O[e][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh[e][u], ph[u], O[e][u], 0, 0, 0);
  }

  if (q_ok) {
    const float inv = 1.0f / l;
    float* op = a.out + ((long)b * T + q_tok) * (a.nh * DH) + head * DH;
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v;
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (O[e][0][4 * g + u] + O[e][1][4 * g + u]) * inv;
        *reinterpret_cast<f32x4*>(op + e * 32 + 8 * g + 4 * h2) = v;
      }
  }
}

// ---- global core on the split-bf16 MFMA ------------------------------------------------------------------------------
// Dense softmax attention per (sample, head) over T <= 256 tokens, same 3-term bf16 split and the same operand scheme
// as the neighbourhood core (S^T = K Q^T with K rows from swizzled LDS, O^T = V^T P^T with P from registers and V staged
// transposed), without halo or window mask: NT = ceil(T/32) waves, wave w owns queries 32w..32w+31 and all NT key tiles
// (the whole score row lives in registers: no online softmax).  K and then V^T occupy the same LDS buffer.
template <int MODE, int NT, int PMODE>
__global__ __launch_bounds__(NT * 64) void attn_global_split_kernel(const DenseArgs a) {
  constexpr bool PREP = PMODE == 1, PK = PMODE == 2;
  extern __shared__ __attribute__((aligned(16))) char gs_smem[];
  const auto warm = code_warm_begin<32768>((int)blockIdx.x < a.warm && blockIdx.y == 0 && threadIdx.x < 64);   // kd_common.h: this kernel's code -> L2
  constexpr int TP = NT * 32, NTHR = NT * 64;
  constexpr int VSTR = TP * 2 + 4;                       // bytes per e-row of V^T: odd dword count -> conflict-free dword-pair reads
  constexpr int IMG_K = TP * 128, IMG_V = DH * VSTR;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, h2 = lane >> 5;
  int b, head, wi = 0, wj = 0;
  if (MODE == MODE_GLOBAL) {
    head = blockIdx.x % a.nh; b = blockIdx.x / a.nh;
  } else {                                               // one workgroup per (sample, head, window): ws^2 slots
    const int nww = a.W >> WinLog2<MODE>::v, nwh = a.H >> WinLog2<MODE>::v;
    int r = blockIdx.x;
    wj = r % nww; r /= nww; wi = r % nwh; r /= nwh; head = r % a.nh; b = r / a.nh;
  }
  const int T = MODE == MODE_GLOBAL ? a.T : (1 << (2 * WinLog2<MODE>::v));          // slots of this problem
  const long row_stride = 3L * a.nh * DH;
  const float* base = a.qkv + (long)b * a.T * row_stride + head * DH;
  const float sqrt_scale = PREP ? sqrtf(a.scale_h[head]) : 1.f;
  const int c16 = tid & 15, rsub = tid >> 4;             // staging: 16 lanes per key row, NTHR/16 rows per round, 8 rounds
  auto tok_of = [&](int slot) -> int {                   // slot (clamped into the problem) -> token index inside the sample
    return slot_token<MODE>(a, min(slot, T - 1), wi, wj);
  };

  // ---- this lane's query: 32 of its 64 dims, 8-wide chunks 2*step + h2 -------------------------------------------------
  const int q_slot = wid * 32 + l31;
  const bool q_ok = q_slot < T;
  const int q_tok = tok_of(q_slot);
  f32x4 qf[8];
  {
    const float* rp = base + (long)q_tok * row_stride;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      qf[2 * st] = *reinterpret_cast<const f32x4*>(rp + 16 * st + 8 * h2);
      qf[2 * st + 1] = *reinterpret_cast<const f32x4*>(rp + 16 * st + 8 * h2 + 4);
    }
  }

  // ---- K -> LDS (all rounds requested before any is consumed) -------------------------------------------------------------
  char* Khi = gs_smem;
  char* Klo = gs_smem + IMG_K;
  {
    f32x4 kreg[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int slot = i * (NTHR / 16) + rsub;
      kreg[i] = *reinterpret_cast<const f32x4*>(base + (long)tok_of(slot) * row_stride + a.nh * DH + 4 * c16);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int slot = i * (NTHR / 16) + rsub;
      f32x4 v = kreg[i];
      if (PREP) {
        const int tk = tok_of(slot);
        v = prep_row16(v, c16, sqrt_scale, a.cos_t + ((long)tk * a.nh + head) * ROT, a.sin_t + ((long)tk * a.nh + head) * ROT, a.eps);
      }
      if (slot >= T) v = f32x4{0.f, 0.f, 0.f, 0.f};
      u32x2 hi, lo;
      split_in<PK>(v, hi, lo);
      const int o = na_kswz(slot, c16 >> 1) + (c16 & 1) * 8;
      *reinterpret_cast<u32x2*>(Khi + o) = hi;
      *reinterpret_cast<u32x2*>(Klo + o) = lo;
    }
  }

  // ---- V prefetch as key pairs (2p, 2p+1): pair index j*(NTHR/16) + rsub, 4 rounds ------------------------------------------
  const float* vbase = base + 2 * a.nh * DH + 4 * c16;
  f32x4 vreg[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int key = 2 * (j * (NTHR / 16) + rsub);
    vreg[2 * j] = *reinterpret_cast<const f32x4*>(vbase + (long)tok_of(key) * row_stride);
    vreg[2 * j + 1] = *reinterpret_cast<const f32x4*>(vbase + (long)tok_of(key + 1) * row_stride);
  }

  // ---- q preparation + split into B-operand fragments -------------------------------------------------------------------------
  if (PREP) {
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) ss += qf[i][0] * qf[i][0] + qf[i][1] * qf[i][1] + qf[i][2] * qf[i][2] + qf[i][3] * qf[i][3];
    ss += __shfl_xor(ss, 32, 64);
    const float f = sqrt_scale * rsqrtf(ss + a.eps);
#pragma unroll
    for (int i = 0; i < 8; ++i) qf[i] = qf[i] * f;
    const float* cs = a.cos_t + ((long)q_tok * a.nh + head) * ROT + 8 * h2;
    const float* sn = a.sin_t + ((long)q_tok * a.nh + head) * ROT + 8 * h2;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const f32x4 cc = *reinterpret_cast<const f32x4*>(cs + 4 * u), sc = *reinterpret_cast<const f32x4*>(sn + 4 * u);
      const f32x4 x1 = qf[u], x2 = qf[2 + u];
      qf[u] = x1 * cc - x2 * sc;
      qf[2 + u] = x2 * cc + x1 * sc;
    }
  }
  bf16x8 qh[4], ql[4];
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    u32x2 h0, l0, h1, l1;
    split_in<PK>(qf[2 * st], h0, l0);
    split_in<PK>(qf[2 * st + 1], h1, l1);
    qh[st] = __builtin_bit_cast(bf16x8, u32x4{h0[0], h0[1], h1[0], h1[1]});
    ql[st] = __builtin_bit_cast(bf16x8, u32x4{l0[0], l0[1], l1[0], l1[1]});
  }
  code_warm_end(warm);
  __syncthreads();

  // ---- S^T = K Q^T over all NT key tiles (step-major, term-major issue order) ------------------------------------------------------
  f32x16 S[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) S[t][i] = 0.f;
#pragma unroll
  for (int st = 0; st < 4; ++st) {
#pragma unroll
    for (int t0 = 0; t0 < NT; t0 += 4) {          // groups of 4 tiles: fragment registers stay bounded at NT = 8
      bf16x8 kh[4], kl[4];
#pragma unroll
      for (int t = 0; t < 4 && t0 + t < NT; ++t) {
        const int o = na_kswz((t0 + t) * 32 + l31, 2 * st + h2);
        kh[t] = *reinterpret_cast<const bf16x8*>(Khi + o);
        kl[t] = *reinterpret_cast<const bf16x8*>(Klo + o);
      }
#pragma unroll
      for (int t = 0; t < 4 && t0 + t < NT; ++t) S[t0 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl[t], qh[st], S[t0 + t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4 && t0 + t < NT; ++t) S[t0 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh[t], ql[st], S[t0 + t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4 && t0 + t < NT; ++t) S[t0 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh[t], qh[st], S[t0 + t], 0, 0, 0);
    }
  }

  const int q_region = (MODE != MODE_GLOBAL) ? slot_region<MODE>(min(q_slot, T - 1), wi, wj, a.shift) : 0;
  // ---- softmax over keys (keys >= T masked by an additive -inf) ----------------------------------------------------------------------
  float m = -INFINITY;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == MODE_GLOBAL) {
        if (t * 32 + 32 > T) S[t][i] += (t * 32 + mfma32_row(i, lane) < T) ? 0.f : -INFINITY;
      } else {
        if ((1 << (2 * WinLog2<MODE>::v)) < TP) S[t][i] += (t * 32 + mfma32_row(i, lane) < T) ? 0.f : -INFINITY;      // 4x4 windows: 16 of 32 slots
        // shifted windows: a key counts only if it lies in the query's wrapped region (:285-316)
        if (a.shift) S[t][i] += (slot_region<MODE>(min(t * 32 + mfma32_row(i, lane), T - 1), wi, wj, a.shift) == q_region) ? 0.f : -INFINITY;
      }
      m = fmaxf(m, S[t][i]);
    }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float l = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float pv = __expf(S[t][i] - m);
      S[t][i] = pv;
      l += pv;
    }
  l += __shfl_xor(l, 32, 64);

  // ---- V -> LDS, transposed and pair-packed: Vt[e][key] --------------------------------------------------------------------------------
  __syncthreads();                       // every wave is done reading K
  char* Vhi = gs_smem;
  char* Vlo = gs_smem + IMG_V;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int key = 2 * (j * (NTHR / 16) + rsub);
    f32x4 v0 = vreg[2 * j], v1 = vreg[2 * j + 1];
    if (key >= T) v0 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (key + 1 >= T) v1 = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x2 h0, l0, h1, l1;
    split_in<PK>(v0, h0, l0);
    split_in<PK>(v1, h1, l1);
    char* ph_ = Vhi + (4 * c16) * VSTR + key * 2;
    char* pl_ = Vlo + (4 * c16) * VSTR + key * 2;
    *reinterpret_cast<unsigned*>(ph_) = __builtin_amdgcn_perm(h1[0], h0[0], 0x05040100u);
    *reinterpret_cast<unsigned*>(ph_ + VSTR) = __builtin_amdgcn_perm(h1[0], h0[0], 0x07060302u);
    *reinterpret_cast<unsigned*>(ph_ + 2 * VSTR) = __builtin_amdgcn_perm(h1[1], h0[1], 0x05040100u);
    *reinterpret_cast<unsigned*>(ph_ + 3 * VSTR) = __builtin_amdgcn_perm(h1[1], h0[1], 0x07060302u);
    *reinterpret_cast<unsigned*>(pl_) = __builtin_amdgcn_perm(l1[0], l0[0], 0x05040100u);
    *reinterpret_cast<unsigned*>(pl_ + VSTR) = __builtin_amdgcn_perm(l1[0], l0[0], 0x07060302u);
    *reinterpret_cast<unsigned*>(pl_ + 2 * VSTR) = __builtin_amdgcn_perm(l1[1], l0[1], 0x05040100u);
    *reinterpret_cast<unsigned*>(pl_ + 3 * VSTR) = __builtin_amdgcn_perm(l1[1], l0[1], 0x07060302u);
  }
  __syncthreads();

  // ---- O^T = V^T P^T --------------------------------------------------------------------------------------------------------------------
  f32x16 O[2][2];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 16; ++i) O[e][u][i] = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    bf16x8 ph[2], pl[2], vh[2][2], vl[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      u32x2 ph0, pl0, ph1, pl1;
      split4_bf16(f32x4{S[t][8 * u], S[t][8 * u + 1], S[t][8 * u + 2], S[t][8 * u + 3]}, ph0, pl0);
      split4_bf16(f32x4{S[t][8 * u + 4], S[t][8 * u + 5], S[t][8 * u + 6], S[t][8 * u + 7]}, ph1, pl1);
      ph[u] = __builtin_bit_cast(bf16x8, u32x4{ph0[0], ph0[1], ph1[0], ph1[1]});
      pl[u] = __builtin_bit_cast(bf16x8, u32x4{pl0[0], pl0[1], pl1[0], pl1[1]});
      const int key0 = t * 32 + 16 * u + 4 * h2;          // k-slots: keys key0..+3 and key0+8..+11
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int o = (32 * e + l31) * VSTR + key0 * 2;
        const unsigned* hp = reinterpret_cast<const unsigned*>(Vhi + o);
        const unsigned* lp = reinterpret_cast<const unsigned*>(Vlo + o);
        vh[e][u] = __builtin_bit_cast(bf16x8, u32x4{hp[0], hp[1], hp[4], hp[5]});
        vl[e][u] = __builtin_bit_cast(bf16x8, u32x4{lp[0], lp[1], lp[4], lp[5]});
      }
    }
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int u = 0; u < 2; ++u) O[e][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl[e][u], ph[u], O[e][u], 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int u = 0; u < 2; ++u) O[e][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh[e][u], pl[u], O[e][u], 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int u = 0; u < 2; ++u) O[e][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh[e][u], ph[u], O[e][u], 0, 0, 0);
  }

  if (q_ok) {
    const float inv = 1.0f / l;
    float* op = a.out + ((long)b * a.T + q_tok) * (a.nh * DH) + head * DH;
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v;
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (O[e][0][4 * g + u] + O[e][1][4 * g + u]) * inv;
        *reinterpret_cast<f32x4*>(op + e * 32 + 8 * g + 4 * h2) = v;
      }
  }
}

// ---- global core for T > 256: key blocks streamed through LDS, online softmax -----------------------------------------
// One workgroup per (sample, head, block of 256 queries): 8 waves x 32 queries, the query fragments and the O^T accumulators
// stay in registers while 128-key blocks of K (row-major, swizzled) and V (transposed, pair-packed) pass through LDS in the
// same split-bf16 operand scheme as above.  The next block's K / V rows are requested from HBM/L2 once the current block's
// scores are done, so their latency hides behind the softmax update and O^T += V^T P^T.
// Running max / sum per query (both lanes of a query keep partial sums; alpha is a per-lane scalar because queries are the
// MFMA column dimension of S^T and O^T).
constexpr int GL_QW = 8, GL_KB = 128, GL_NTK = GL_KB / 32, GL_THR = GL_QW * 64;
constexpr int GL_VSTR = GL_KB * 2 + 4;
constexpr int GL_IMG_K = GL_KB * 128, GL_IMG_V = DH * GL_VSTR;
constexpr int GL_LDS = 2 * GL_IMG_K + 2 * GL_IMG_V;

template <int PMODE>
__global__ __launch_bounds__(GL_THR) void attn_global_long_kernel(const DenseArgs a) {
  constexpr bool PREP = PMODE == 1, PK = PMODE == 2;
  extern __shared__ __attribute__((aligned(16))) char gl_smem[];
  (void)code_warm_begin<32768>((int)blockIdx.x < a.warm && blockIdx.y == 0 && threadIdx.x < 64);   // kd_common.h: this kernel's code -> L2
  char* Khi = gl_smem;
  char* Klo = gl_smem + GL_IMG_K;
  char* Vhi = gl_smem + 2 * GL_IMG_K;
  char* Vlo = Vhi + GL_IMG_V;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, h2 = lane >> 5;
  const int T = a.T, nqb = (T + GL_QW * 32 - 1) / (GL_QW * 32);
  int r = blockIdx.x;
  const int qb = r % nqb; r /= nqb;
  const int head = r % a.nh, b = r / a.nh;
  const long row_stride = 3L * a.nh * DH;
  const float* base = a.qkv + (long)b * T * row_stride + head * DH;
  const float sqrt_scale = PREP ? sqrtf(a.scale_h[head]) : 1.f;
  const int c16 = tid & 15, rsub = tid >> 4;             // staging: 16 lanes per key row, 32 rows per round

  // ---- this lane's query (32 of its 64 dims: 8-wide chunks 2*step + h2), prepared and split once ---------------------------
  const int q_slot = qb * (GL_QW * 32) + wid * 32 + l31;
  const bool q_ok = q_slot < T;
  const int q_tok = min(q_slot, T - 1);
  bf16x8 qh[4], ql[4];
  {
    f32x4 qf[8];
    const float* rp = base + (long)q_tok * row_stride;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      qf[2 * st] = *reinterpret_cast<const f32x4*>(rp + 16 * st + 8 * h2);
      qf[2 * st + 1] = *reinterpret_cast<const f32x4*>(rp + 16 * st + 8 * h2 + 4);
    }
    if (PREP) {
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) ss += qf[i][0] * qf[i][0] + qf[i][1] * qf[i][1] + qf[i][2] * qf[i][2] + qf[i][3] * qf[i][3];
      ss += __shfl_xor(ss, 32, 64);
      const float f = sqrt_scale * rsqrtf(ss + a.eps);
#pragma unroll
      for (int i = 0; i < 8; ++i) qf[i] = qf[i] * f;
      const float* cs = a.cos_t + ((long)q_tok * a.nh + head) * ROT + 8 * h2;
      const float* sn = a.sin_t + ((long)q_tok * a.nh + head) * ROT + 8 * h2;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const f32x4 cc = *reinterpret_cast<const f32x4*>(cs + 4 * u), sc = *reinterpret_cast<const f32x4*>(sn + 4 * u);
        const f32x4 x1 = qf[u], x2 = qf[2 + u];
        qf[u] = x1 * cc - x2 * sc;
        qf[2 + u] = x2 * cc + x1 * sc;
      }
    }
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      u32x2 h0, l0, h1, l1;
      split_in<PK>(qf[2 * st], h0, l0);
      split_in<PK>(qf[2 * st + 1], h1, l1);
      qh[st] = __builtin_bit_cast(bf16x8, u32x4{h0[0], h0[1], h1[0], h1[1]});
      ql[st] = __builtin_bit_cast(bf16x8, u32x4{l0[0], l0[1], l1[0], l1[1]});
    }
  }

  // ---- key-block staging registers: K rows i*32 + rsub (4 rounds), V key pairs 2*(j*32 + rsub) (2 rounds) -------------------
  f32x4 kreg[4], vreg[4];
  auto request = [&](int kb) {
    const int k0 = kb * GL_KB;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      kreg[i] = *reinterpret_cast<const f32x4*>(base + (long)min(k0 + i * 32 + rsub, T - 1) * row_stride + a.nh * DH + 4 * c16);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int key = k0 + 2 * (j * 32 + rsub);
      vreg[2 * j] = *reinterpret_cast<const f32x4*>(base + (long)min(key, T - 1) * row_stride + 2 * a.nh * DH + 4 * c16);
      vreg[2 * j + 1] = *reinterpret_cast<const f32x4*>(base + (long)min(key + 1, T - 1) * row_stride + 2 * a.nh * DH + 4 * c16);
    }
  };
  request(0);

  f32x16 O[2][2];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 16; ++i) O[e][u][i] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const int nkb = (T + GL_KB - 1) / GL_KB;

  for (int kb = 0; kb < nkb; ++kb) {
    const int k0 = kb * GL_KB;
    // ---- publish this block: K rows (prepared, split, swizzled), V^T (split, key pairs packed) ------------------------------
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int slot = i * 32 + rsub, key = k0 + slot;
      f32x4 v = kreg[i];
      if (PREP) {
        const int tk = min(key, T - 1);
        v = prep_row16(v, c16, sqrt_scale, a.cos_t + ((long)tk * a.nh + head) * ROT, a.sin_t + ((long)tk * a.nh + head) * ROT, a.eps);
      }
      if (key >= T) v = f32x4{0.f, 0.f, 0.f, 0.f};
      u32x2 hi, lo;
      split_in<PK>(v, hi, lo);
      const int o = na_kswz(slot, c16 >> 1) + (c16 & 1) * 8;
      *reinterpret_cast<u32x2*>(Khi + o) = hi;
      *reinterpret_cast<u32x2*>(Klo + o) = lo;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int slot = 2 * (j * 32 + rsub), key = k0 + slot;
      f32x4 v0 = vreg[2 * j], v1 = vreg[2 * j + 1];
      if (key >= T) v0 = f32x4{0.f, 0.f, 0.f, 0.f};
      if (key + 1 >= T) v1 = f32x4{0.f, 0.f, 0.f, 0.f};
      u32x2 h0, l0, h1, l1;
      split_in<PK>(v0, h0, l0);
      split_in<PK>(v1, h1, l1);
      char* ph_ = Vhi + (4 * c16) * GL_VSTR + slot * 2;
      char* pl_ = Vlo + (4 * c16) * GL_VSTR + slot * 2;
      *reinterpret_cast<unsigned*>(ph_) = __builtin_amdgcn_perm(h1[0], h0[0], 0x05040100u);
      *reinterpret_cast<unsigned*>(ph_ + GL_VSTR) = __builtin_amdgcn_perm(h1[0], h0[0], 0x07060302u);
      *reinterpret_cast<unsigned*>(ph_ + 2 * GL_VSTR) = __builtin_amdgcn_perm(h1[1], h0[1], 0x05040100u);
      *reinterpret_cast<unsigned*>(ph_ + 3 * GL_VSTR) = __builtin_amdgcn_perm(h1[1], h0[1], 0x07060302u);
      *reinterpret_cast<unsigned*>(pl_) = __builtin_amdgcn_perm(l1[0], l0[0], 0x05040100u);
      *reinterpret_cast<unsigned*>(pl_ + GL_VSTR) = __builtin_amdgcn_perm(l1[0], l0[0], 0x07060302u);
      *reinterpret_cast<unsigned*>(pl_ + 2 * GL_VSTR) = __builtin_amdgcn_perm(l1[1], l0[1], 0x05040100u);
      *reinterpret_cast<unsigned*>(pl_ + 3 * GL_VSTR) = __builtin_amdgcn_perm(l1[1], l0[1], 0x07060302u);
    }
    __syncthreads();

    // ---- S^T = K Q^T for the block's 4 key tiles ------------------------------------------------------------------------------
    f32x16 S[GL_NTK];
#pragma unroll
    for (int t = 0; t < GL_NTK; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) S[t][i] = 0.f;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      bf16x8 kh[GL_NTK], kl[GL_NTK];
#pragma unroll
      for (int t = 0; t < GL_NTK; ++t) {
        const int o = na_kswz(t * 32 + l31, 2 * st + h2);
        kh[t] = *reinterpret_cast<const bf16x8*>(Khi + o);
        kl[t] = *reinterpret_cast<const bf16x8*>(Klo + o);
      }
#pragma unroll
      for (int t = 0; t < GL_NTK; ++t) S[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl[t], qh[st], S[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < GL_NTK; ++t) S[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh[t], ql[st], S[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < GL_NTK; ++t) S[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh[t], qh[st], S[t], 0, 0, 0);
    }

    if (kb + 1 < nkb) request(kb + 1);          // in flight during the softmax update and the P V MFMAs (kept out of the
                                                // S phase, whose q fragments already fill the register budget)
    // ---- online softmax update (keys >= T carry -inf; every block holds at least one real key) --------------------------------
    float m_blk = -INFINITY;
#pragma unroll
    for (int t = 0; t < GL_NTK; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (k0 + GL_KB > T) S[t][i] += (k0 + t * 32 + mfma32_row(i, lane) < T) ? 0.f : -INFINITY;
        m_blk = fmaxf(m_blk, S[t][i]);
      }
    m_blk = fmaxf(m_blk, __shfl_xor(m_blk, 32, 64));
    const float m_new = fmaxf(m_run, m_blk);
    const float alpha = __expf(m_run - m_new);       // first block: exp(-inf) = 0
    m_run = m_new;
    float l_blk = 0.f;
#pragma unroll
    for (int t = 0; t < GL_NTK; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float pv = __expf(S[t][i] - m_new);
        S[t][i] = pv;
        l_blk += pv;
      }
    l_run = l_run * alpha + l_blk;
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) O[e][u][i] *= alpha;

    // ---- O^T += V^T P^T ----------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int t = 0; t < GL_NTK; ++t) {
      bf16x8 ph[2], pl[2], vh[2][2], vl[2][2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        u32x2 ph0, pl0, ph1, pl1;
        split4_bf16(f32x4{S[t][8 * u], S[t][8 * u + 1], S[t][8 * u + 2], S[t][8 * u + 3]}, ph0, pl0);
        split4_bf16(f32x4{S[t][8 * u + 4], S[t][8 * u + 5], S[t][8 * u + 6], S[t][8 * u + 7]}, ph1, pl1);
        ph[u] = __builtin_bit_cast(bf16x8, u32x4{ph0[0], ph0[1], ph1[0], ph1[1]});
        pl[u] = __builtin_bit_cast(bf16x8, u32x4{pl0[0], pl0[1], pl1[0], pl1[1]});
        const int key0 = t * 32 + 16 * u + 4 * h2;          // k-slots: keys key0..+3 and key0+8..+11
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int o = (32 * e + l31) * GL_VSTR + key0 * 2;
          const unsigned* hp = reinterpret_cast<const unsigned*>(Vhi + o);
          const unsigned* lp = reinterpret_cast<const unsigned*>(Vlo + o);
          vh[e][u] = __builtin_bit_cast(bf16x8, u32x4{hp[0], hp[1], hp[4], hp[5]});
          vl[e][u] = __builtin_bit_cast(bf16x8, u32x4{lp[0], lp[1], lp[4], lp[5]});
        }
      }
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int u = 0; u < 2; ++u) O[e][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl[e][u], ph[u], O[e][u], 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int u = 0; u < 2; ++u) O[e][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh[e][u], pl[u], O[e][u], 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int u = 0; u < 2; ++u) O[e][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh[e][u], ph[u], O[e][u], 0, 0, 0);
    }
    __syncthreads();                            // every wave is done with this block's images
  }

  if (q_ok) {
    const float l = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l;
    float* op = a.out + ((long)b * T + q_tok) * (a.nh * DH) + head * DH;
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v;
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (O[e][0][4 * g + u] + O[e][1][4 * g + u]) * inv;
        *reinterpret_cast<f32x4*>(op + e * 32 + 8 * g + 4 * h2) = v;
      }
  }
}

static int launch_global_long(const DenseArgs& a, int prep, hipStream_t s) {
  const long nqb = (a.T + GL_QW * 32 - 1) / (GL_QW * 32);
  const long nblocks = (long)a.batch * a.nh * nqb;
  LaunchScope prof("attn_global_bf16x3", 4.0 * (double)a.batch * a.nh * a.T * a.T * DH, 4.0 * (double)a.batch * a.T * a.nh * DH * 4.0, s);
#define KD_GL(PM)                                                                                                                                   \
  {                                                                                                                                                \
    auto k = attn_global_long_kernel<PM>;                                                                                                          \
    static LdsAttr set;                                                                                                                       \
    set.ensure(reinterpret_cast<const void*>(k), GL_LDS);   \
    hipLaunchKernelGGL(k, dim3((unsigned)nblocks), dim3(GL_THR), GL_LDS, s, a);                                                                    \
  }
  if (prep == 1) KD_GL(1) else if (prep == 2) KD_GL(2) else KD_GL(0)
#undef KD_GL
  return check_launch("kd_attn_global_f32");
}

template <int MODE, int NT>
static int launch_global_split(const DenseArgs& a, int prep, long nblocks, hipStream_t s) {
  constexpr int TP = NT * 32, VSTR = TP * 2 + 4;
  constexpr int lds = 2 * (TP * 128 > DH * VSTR ? TP * 128 : DH * VSTR);
  const int n_slots = MODE == MODE_GLOBAL ? a.T : (1 << (2 * WinLog2<MODE>::v));
  LaunchScope prof(MODE == MODE_GLOBAL ? "attn_global_bf16x3" : "attn_window_bf16x3", 4.0 * (double)nblocks * n_slots * n_slots * DH,
                   4.0 * (double)a.batch * a.T * a.nh * DH * 4.0, s);
#define KD_GS(PM)                                                                                                                                \
  {                                                                                                                                             \
    auto k = attn_global_split_kernel<MODE, NT, PM>;                                                                                            \
    static LdsAttr set;                                                                                                                    \
    set.ensure(reinterpret_cast<const void*>(k), lds);   \
    hipLaunchKernelGGL(k, dim3((unsigned)nblocks), dim3(NT * 64), lds, s, a);                                                                   \
  }
  if (prep == 1) KD_GS(1) else if (prep == 2) KD_GS(2) else KD_GS(0)
#undef KD_GS
  return check_launch("kd_attn_global_f32");
}

template <int MODE, int MAXT>
static int launch_dense(const DenseArgs& a, int prep, long nblocks, const char* name, hipStream_t s) {
  const size_t lds = (size_t)2 * MAXT * 32 * LDS_ROW * sizeof(float);
  const int n_slots = MODE == MODE_GLOBAL ? a.T : (1 << (2 * WinLog2<MODE>::v));
  const double flops = 4.0 * (double)nblocks * n_slots * n_slots * DH;
  const double bytes = 4.0 * (double)a.batch * a.T * a.nh * DH * 4.0;
  LaunchScope prof(name, flops, bytes, s);
  if (prep) {
    auto k = attn_dense_kernel<MODE, MAXT, true>;
    static LdsAttr set;
    set.ensure(reinterpret_cast<const void*>(k), (int)lds);
    hipLaunchKernelGGL(k, dim3((unsigned)nblocks), dim3(MAXT * 64), lds, s, a);
  } else {
    auto k = attn_dense_kernel<MODE, MAXT, false>;
    static LdsAttr set;
    set.ensure(reinterpret_cast<const void*>(k), (int)lds);
    hipLaunchKernelGGL(k, dim3((unsigned)nblocks), dim3(MAXT * 64), lds, s, a);
  }
  return check_launch(name);
}

}  // namespace kd

using namespace kd;

static int check_prep(int prep, const float* scale_h, const float* cos_t, const float* sin_t, const char* who) {
  if (prep < 0 || prep > 2) return fail(KD_EINVAL, "%s: prep must be 0 (prepared fp32), 1 (prepare here) or 2 (prepared, stored split)", who);
  if (prep == 1 && (!scale_h || !cos_t || !sin_t)) return fail(KD_EINVAL, "%s: prep needs scale_h, cos_t, sin_t", who);
  return KD_OK;
}

extern "C" int kd_qk_prep_f32(float* qkv, const float* scale_h, const float* cos_t, const float* sin_t,
                              int batch, int tokens_per_sample, int nh, float eps, void* stream) {
  if (!qkv || batch <= 0 || tokens_per_sample <= 0 || nh <= 0) return fail(KD_EINVAL, "kd_qk_prep_f32: bad arguments");
  if (int e = check_prep(1, scale_h, cos_t, sin_t, "kd_qk_prep_f32")) return e;
  const long rows = (long)batch * tokens_per_sample * 2 * nh;
  const long blocks = (rows * 16 + 255) / 256;
  hipStream_t s = (hipStream_t)stream;
  LaunchScope prof("qk_prep_f32", 0, (double)rows * DH * 8, s);
  hipLaunchKernelGGL(qk_prep_kernel, dim3((unsigned)blocks), dim3(256), 0, s, qkv, scale_h, cos_t, sin_t, rows, tokens_per_sample, nh, eps);
  return check_launch("kd_qk_prep_f32");
}

namespace kd { int attn_global_x3_try(const float* qkv, float* out, int batch, int T, int nh, hipStream_t s, int* rc); }   // attn_x3.hip

extern "C" int kd_attn_global_f32(const float* qkv, float* out, int batch, int T, int nh, int prep, const float* scale_h,
                                  const float* cos_t, const float* sin_t, float eps, int precision, void* stream) {
  if (!qkv || !out || batch <= 0 || nh <= 0 || T <= 0) return fail(KD_EINVAL, "kd_attn_global_f32: bad arguments");
  if (int e = check_prep(prep, scale_h, cos_t, sin_t, "kd_attn_global_f32")) return e;
  DenseArgs a{qkv, out, scale_h, cos_t, sin_t, batch, T, nh, 0, 0, 0, 0, eps, option("code_warm", KD_CODE_WARM_DEFAULT)};
  const long nb = (long)batch * nh;
  hipStream_t s = (hipStream_t)stream;
  if (precision != KD_PREC_EXACT && precision != KD_PREC_SPLIT3) return fail(KD_EINVAL, "kd_attn_global_f32: precision must be KD_PREC_EXACT or KD_PREC_SPLIT3");
  const bool exact = precision == KD_PREC_EXACT;
  if (exact && prep == 2) return fail(KD_EINVAL, "kd_attn_global_f32: split-stored qkv (prep = 2) is for the split-bf16x3 cores, not KD_PREC_EXACT");
  if (T > 256) {
    if (exact) return fail(KD_EINVAL, "kd_attn_global_f32: T=%d > 256 tokens is served by the streaming split-bf16x3 core only (KD_PREC_SPLIT3)", T);
    return launch_global_long(a, prep, s);
  }
  if (!exact) {
    if (prep == 2) {      // operands stored split by the qkv projection: the round-3 core (two workgroups per CU, LDS-DMA, transposing V reads)
      int rc = 0;
      if (!attn_global_x3_try(qkv, out, batch, T, nh, s, &rc)) return rc;
    }
    if (T <= 64) return launch_global_split<MODE_GLOBAL, 2>(a, prep, nb, s);
    if (T <= 128) return launch_global_split<MODE_GLOBAL, 4>(a, prep, nb, s);
    return launch_global_split<MODE_GLOBAL, 8>(a, prep, nb, s);
  }
  if (T <= 64) return launch_dense<MODE_GLOBAL, 2>(a, prep, nb, "attn_global_f32", s);
  if (T <= 128) return launch_dense<MODE_GLOBAL, 4>(a, prep, nb, "attn_global_f32", s);
  return launch_dense<MODE_GLOBAL, 8>(a, prep, nb, "attn_global_f32", s);
}

extern "C" int kd_attn_window_f32(const float* qkv, float* out, int batch, int H, int W, int nh, int ws, int shift, int prep,
                                  const float* scale_h, const float* cos_t, const float* sin_t, float eps, int precision, void* stream) {
  if (!qkv || !out || batch <= 0 || nh <= 0 || H <= 0 || W <= 0) return fail(KD_EINVAL, "kd_attn_window_f32: bad arguments");
  if (ws != 4 && ws != 8 && ws != 16) return fail(KD_EINVAL, "kd_attn_window_f32: window_size %d unsupported (4, 8 or 16)", ws);
  if ((H % ws) || (W % ws)) return fail(KD_EINVAL, "kd_attn_window_f32: grid %dx%d not divisible by the window", H, W);
  if (shift < 0 || shift >= ws) return fail(KD_EINVAL, "kd_attn_window_f32: bad shift %d", shift);
  if (int e = check_prep(prep, scale_h, cos_t, sin_t, "kd_attn_window_f32")) return e;
  DenseArgs a{qkv, out, scale_h, cos_t, sin_t, batch, H * W, nh, H, W, ws, shift, eps, option("code_warm", KD_CODE_WARM_DEFAULT)};
  const long nb = (long)batch * nh * (H / ws) * (W / ws);
  if (precision != KD_PREC_EXACT && precision != KD_PREC_SPLIT3) return fail(KD_EINVAL, "kd_attn_window_f32: precision must be KD_PREC_EXACT or KD_PREC_SPLIT3");
  hipStream_t s = (hipStream_t)stream;
  if (precision == KD_PREC_SPLIT3) {
    if (ws == 8) return launch_global_split<MODE_WINDOW, 2>(a, prep, nb, s);
    if (ws == 4) return launch_global_split<MODE_WINDOW4, 1>(a, prep, nb, s);
    return launch_global_split<MODE_WINDOW16, 8>(a, prep, nb, s);
  }
  if (prep == 2) return fail(KD_EINVAL, "kd_attn_window_f32: split-stored qkv (prep = 2) is for the split-bf16x3 cores, not KD_PREC_EXACT");
  if (ws == 8) return launch_dense<MODE_WINDOW, 2>(a, prep, nb, "attn_window_f32", s);
  if (ws == 4) return launch_dense<MODE_WINDOW4, 1>(a, prep, nb, "attn_window_f32", s);
  return launch_dense<MODE_WINDOW16, 8>(a, prep, nb, "attn_window_f32", s);
}

namespace kd { int attn_na2d_x3_try(const float* qkv, float* out, int batch, int H, int W, int nh, int ks, hipStream_t s, int* rc); }   // attn_x3.hip

extern "C" int kd_attn_na2d_f32(const float* qkv, float* out, int batch, int H, int W, int nh, int ks, int prep,
                                const float* scale_h, const float* cos_t, const float* sin_t, float eps, int precision, void* stream) {
  (void)precision;
  if (!qkv || !out || batch <= 0 || nh <= 0) return fail(KD_EINVAL, "kd_attn_na2d_f32: bad arguments");
  if (ks < 3 || ks > 13 || !(ks & 1)) return fail(KD_EINVAL, "kd_attn_na2d_f32: kernel_size %d unsupported (odd sizes 3 .. 13)", ks);
  if (H < ks || W < ks) return fail(KD_EINVAL, "kd_attn_na2d_f32: grid %dx%d smaller than the %dx%d neighbourhood", H, W, ks, ks);
  if (int e = check_prep(prep, scale_h, cos_t, sin_t, "kd_attn_na2d_f32")) return e;
  hipStream_t s = (hipStream_t)stream;
  if (prep == 2) {        // operands stored split by the qkv projection: the round-3 cores (LDS-DMA halo, transposing V reads), every kernel size
    int rc = 0;
    if (!attn_na2d_x3_try(qkv, out, batch, H, W, nh, ks, s, &rc)) return rc;
  }
  // the round-1 core below (fp32 operands, or split-stored ones with option "attn_x3" = 0) is built for the size the shipped configs use
  if (ks != NA_K)
    return fail(KD_EINVAL, "kd_attn_na2d_f32: kernel_size %d needs split-stored operands (prep = 2, KdGemm.qkv_packed) and option attn_x3; "
                           "fp32 operands: kernel_size 7 only", ks);
  NaArgs a{qkv, out, scale_h, cos_t, sin_t, batch, H, W, nh, eps, option("code_warm", KD_CODE_WARM_DEFAULT)};
  const long nb = (long)batch * nh * ((H + NA_TH - 1) / NA_TH) * ((W + NA_TW - 1) / NA_TW);
  char nm[64] = "attn_na2d";
  if (prof_on()) snprintf(nm, sizeof(nm), "attn_na2d %dx%d nh=%d", H, W, nh);
  LaunchScope prof(nm, 4.0 * batch * (double)H * W * nh * DH * ks * ks, 16.0 * batch * (double)H * W * nh * DH, s);
  static LdsAttr attr_set[6];
#define KD_NA_ATTR(I, PM, FU) attr_set[I].ensure(reinterpret_cast<const void*>(attn_na2d_kernel<PM, FU>), NA_LDS);
  KD_NA_ATTR(0, 0, true) KD_NA_ATTR(1, 0, false) KD_NA_ATTR(2, 1, true) KD_NA_ATTR(3, 1, false) KD_NA_ATTR(4, 2, true) KD_NA_ATTR(5, 2, false)
#undef KD_NA_ATTR
  const bool full = H >= NA_HR && W >= NA_HC;      // every halo key is inside the image
#define KD_NA(PM)                                                                                              \
  {                                                                                                           \
    if (full) hipLaunchKernelGGL((attn_na2d_kernel<PM, true>), dim3((unsigned)nb), dim3(256), NA_LDS, s, a); \
    else hipLaunchKernelGGL((attn_na2d_kernel<PM, false>), dim3((unsigned)nb), dim3(256), NA_LDS, s, a);      \
  }
  if (prep == 1) KD_NA(1) else if (prep == 2) KD_NA(2) else KD_NA(0)
#undef KD_NA
  return check_launch("kd_attn_na2d_f32");
}

KD_TEXT_PAD(attn_f32)      // last function of this code object: kd_common.h, code warm-up
