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
// The neighbourhood core is VALU + LDS: at fp32 the MFMA rate equals the VALU rate, and a dense
// MFMA tiling of a 7x7 window wastes >55% of its work on masked keys.  Four lanes share a query
// (16 head dims each, two DPP shuffles per score); the 14x14 key halo of an 8x8 query tile is staged
// in LDS, K first, then V in the same buffer.
#include "kd_common.h"

namespace kd {

constexpr int DH = 64;          // head dim (fixed: every config, k_diffusion/config.py:135-136)
constexpr int LDS_ROW = DH + 4; // padded LDS row (floats)
constexpr int ROT = 16;         // rotary angles per head: dims [0,16) pair with [16,32)

// ---- q/k row preparation, 16 lanes per 64-float row: lane c = lane & 15 owns dims [4c, 4c+4) -------
// scale_for_cosine_sim (image_transformer_v2.py:106-114) then _apply_rotary_emb_inplace (:187-199).
__device__ __forceinline__ f32x4 prep_row16(f32x4 v, int c, float sqrt_scale, const float* cs_row, const float* sn_row, float eps) {
  float ss = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  ss = wave_sum_xor(ss, 16);
  const float f = sqrt_scale * rsqrtf(ss + eps);
  v = v * f;
  f32x4 o;
#pragma unroll
  for (int u = 0; u < 4; ++u) o[u] = __shfl_xor(v[u], 4, 64);
  if (c < 8) {
    const f32x4 cs = *reinterpret_cast<const f32x4*>(cs_row + 4 * (c & 3));
    const f32x4 sn = *reinterpret_cast<const f32x4*>(sn_row + 4 * (c & 3));
    v = (c < 4) ? (v * cs - o * sn) : (v * cs + o * sn);
  }
  return v;
}

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
enum { MODE_GLOBAL = 0, MODE_WINDOW = 1 };

struct DenseArgs {
  const float* qkv; float* out;
  const float* scale_h; const float* cos_t; const float* sin_t;
  int batch, T, nh;          // T = tokens per sample
  int H, W, ws, shift;       // window mode
  float eps;
};

// slot -> token index inside the sample (or -1), for the problem owned by this block
template <int MODE>
__device__ __forceinline__ int slot_token(const DenseArgs& a, int slot, int wi, int wj) {
  if (MODE == MODE_GLOBAL) return slot < a.T ? slot : -1;
  const int ai = slot >> 3, bj = slot & 7;                 // ws == 8
  int i = wi * 8 + ai - a.shift; if (i < 0) i += a.H;      // rolled[i] = orig[(i - shift) mod H]  (:274)
  int j = wj * 8 + bj - a.shift; if (j < 0) j += a.W;
  return i * a.W + j;
}
// wrapped-region id of a window slot (make_shifted_window_masks, :285-316)
__device__ __forceinline__ int slot_region(int slot, int wi, int wj, int shift) {
  return ((wi == 0 && (slot >> 3) < shift) ? 2 : 0) + ((wj == 0 && (slot & 7) < shift) ? 1 : 0);
}

template <int MODE, int MAXT, bool PREP>
__global__ __launch_bounds__(MAXT * 64) void attn_dense_kernel(const DenseArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int TP = MAXT * 32;
  float* Ks = smem;                 // [TP][LDS_ROW]
  float* Vs = smem + TP * LDS_ROW;  // [TP][LDS_ROW]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  int b, head, wi = 0, wj = 0;
  if (MODE == MODE_GLOBAL) {
    head = blockIdx.x % a.nh; b = blockIdx.x / a.nh;
  } else {
    const int nww = a.W >> 3, nwh = a.H >> 3;
    int r = blockIdx.x;
    wj = r % nww; r /= nww; wi = r % nwh; r /= nwh; head = r % a.nh; b = r / a.nh;
  }
  const int n_slots = (MODE == MODE_GLOBAL) ? a.T : 64;
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
  const int q_region = (MODE == MODE_WINDOW) ? slot_region(q_slot, wi, wj, a.shift) : 0;
  auto key_bias = [&](int t, int r) -> float {
    const int ks = t * 32 + mfma32_row(r, lane);
    bool ok = ks < n_slots;
    if (MODE == MODE_WINDOW && a.shift) ok = ok && (slot_region(ks, wi, wj, a.shift) == q_region);
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
struct NaArgs {
  const float* qkv; float* out;
  const float* scale_h; const float* cos_t; const float* sin_t;
  int batch, H, W, nh;
  float eps;
};

constexpr int NA_K = 7, NA_TILE = 8, NA_HALO = NA_TILE + NA_K - 1;   // 14

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

template <bool PREP>
__global__ __launch_bounds__(256) void attn_na2d_kernel(const NaArgs a) {
  __shared__ __attribute__((aligned(16))) float KV[NA_HALO * NA_HALO * LDS_ROW];   // K, then V
  const int tid = threadIdx.x;
  const int tiles_x = (a.W + NA_TILE - 1) / NA_TILE, tiles_y = (a.H + NA_TILE - 1) / NA_TILE;
  int r = blockIdx.x;
  const int tx = r % tiles_x; r /= tiles_x;
  const int ty = r % tiles_y; r /= tiles_y;
  const int head = r % a.nh; const int b = r / a.nh;
  const int T = a.H * a.W;
  const long row_stride = 3L * a.nh * DH;
  const float* base = a.qkv + (long)b * T * row_stride + head * DH;
  const float sqrt_scale = PREP ? sqrtf(a.scale_h[head]) : 1.f;
  const int ty0 = ty * NA_TILE, tx0 = tx * NA_TILE;
  const int hy0 = clampi(ty0 - NA_K / 2, 0, max(0, a.H - NA_HALO));
  const int hx0 = clampi(tx0 - NA_K / 2, 0, max(0, a.W - NA_HALO));

  auto stage = [&](int which) {   // which: 1 = K (prepared), 2 = V
    const int c = tid & 15;
    for (int hr = tid >> 4; hr < NA_HALO * NA_HALO; hr += 16) {
      const int ky = hy0 + hr / NA_HALO, kx = hx0 + hr % NA_HALO;
      const bool ok = ky < a.H && kx < a.W;
      const int tok = ok ? ky * a.W + kx : 0;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) v = *reinterpret_cast<const f32x4*>(base + (long)tok * row_stride + which * a.nh * DH + 4 * c);
      if (PREP && which == 1)
        v = prep_row16(v, c, sqrt_scale, a.cos_t + ((long)tok * a.nh + head) * ROT, a.sin_t + ((long)tok * a.nh + head) * ROT, a.eps);
      *reinterpret_cast<f32x4*>(KV + hr * LDS_ROW + 4 * c) = v;
    }
  };
  stage(1);

  // ---- this lane's query quarter: query qb = tid>>2 of the 8x8 tile, dims [16*pi, 16*pi+16) ----------
  const int qb = tid >> 2, pi = tid & 3;
  const int qy = ty0 + (qb >> 3), qx = tx0 + (qb & 7);
  const bool q_ok = qy < a.H && qx < a.W;
  const int q_tok = q_ok ? qy * a.W + qx : 0;
  f32x4 q[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) q[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (q_ok) {
    const float* rp = base + (long)q_tok * row_stride + 16 * pi;
#pragma unroll
    for (int j = 0; j < 4; ++j) q[j] = *reinterpret_cast<const f32x4*>(rp + 4 * j);
  }
  if (PREP) {
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) ss += q[j][0] * q[j][0] + q[j][1] * q[j][1] + q[j][2] * q[j][2] + q[j][3] * q[j][3];
    ss = wave_sum_xor(ss, 4);
    const float f = sqrt_scale * rsqrtf(ss + a.eps);
    const float* cs = a.cos_t + ((long)q_tok * a.nh + head) * ROT;
    const float* sn = a.sin_t + ((long)q_tok * a.nh + head) * ROT;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      q[j] = q[j] * f;
      f32x4 o;
#pragma unroll
      for (int u = 0; u < 4; ++u) o[u] = __shfl_xor(q[j][u], 1, 64);   // pi 0 <-> 1 : dims d <-> d+16
      if (pi < 2) {
        const f32x4 cc = *reinterpret_cast<const f32x4*>(cs + 4 * j);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(sn + 4 * j);
        q[j] = (pi == 0) ? (q[j] * cc - o * sc) : (q[j] * cc + o * sc);
      }
    }
  }
  // clamped window start (NATTEN semantics, dilation 1), relative to the halo origin
  const int wy = clampi(qy - NA_K / 2, 0, a.H - NA_K) - hy0;
  const int wx = clampi(qx - NA_K / 2, 0, a.W - NA_K) - hx0;
  __syncthreads();

  // ---- scores over the 49 keys ---------------------------------------------------------------------
  float s[NA_K * NA_K];
  float m = -INFINITY;
  if (q_ok) {
#pragma unroll
    for (int ai = 0; ai < NA_K; ++ai) {
#pragma unroll
      for (int bj = 0; bj < NA_K; ++bj) {
        const float* kp = KV + ((wy + ai) * NA_HALO + wx + bj) * LDS_ROW + 16 * pi;
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 kf = *reinterpret_cast<const f32x4*>(kp + 4 * j);
          d += q[j][0] * kf[0] + q[j][1] * kf[1] + q[j][2] * kf[2] + q[j][3] * kf[3];
        }
        s[ai * NA_K + bj] = d;
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < NA_K * NA_K; ++i) s[i] = 0.f;
  }
  float l = 0.f;
#pragma unroll
  for (int i = 0; i < NA_K * NA_K; ++i) {
    s[i] = wave_sum_xor(s[i], 4);
    m = fmaxf(m, s[i]);
  }
#pragma unroll
  for (int i = 0; i < NA_K * NA_K; ++i) {
    s[i] = expf(s[i] - m);
    l += s[i];
  }
  __syncthreads();
  stage(2);
  __syncthreads();

  // ---- output quarter ---------------------------------------------------------------------------
  f32x4 o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (q_ok) {
#pragma unroll
    for (int ai = 0; ai < NA_K; ++ai) {
#pragma unroll
      for (int bj = 0; bj < NA_K; ++bj) {
        const float* vp = KV + ((wy + ai) * NA_HALO + wx + bj) * LDS_ROW + 16 * pi;
        const float p = s[ai * NA_K + bj];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = o[j] + *reinterpret_cast<const f32x4*>(vp + 4 * j) * p;
      }
    }
    const float inv = 1.0f / l;
    float* op = a.out + ((long)b * T + q_tok) * (a.nh * DH) + head * DH + 16 * pi;
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(op + 4 * j) = o[j] * inv;
  }
}

template <int MODE, int MAXT>
static int launch_dense(const DenseArgs& a, int prep, long nblocks, const char* name, hipStream_t s) {
  const size_t lds = (size_t)2 * MAXT * 32 * LDS_ROW * sizeof(float);
  const int n_slots = MODE == MODE_GLOBAL ? a.T : 64;
  const double flops = 4.0 * (double)nblocks * n_slots * n_slots * DH;
  const double bytes = 4.0 * (double)a.batch * a.T * a.nh * DH * 4.0;
  LaunchScope prof(name, flops, bytes, s);
  if (prep) {
    auto k = attn_dense_kernel<MODE, MAXT, true>;
    static bool set = false;
    if (!set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); set = true; }
    hipLaunchKernelGGL(k, dim3((unsigned)nblocks), dim3(MAXT * 64), lds, s, a);
  } else {
    auto k = attn_dense_kernel<MODE, MAXT, false>;
    static bool set = false;
    if (!set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); set = true; }
    hipLaunchKernelGGL(k, dim3((unsigned)nblocks), dim3(MAXT * 64), lds, s, a);
  }
  return check_launch(name);
}

}  // namespace kd

using namespace kd;

static int check_prep(int prep, const float* scale_h, const float* cos_t, const float* sin_t, const char* who) {
  if (prep && (!scale_h || !cos_t || !sin_t)) return fail(KD_EINVAL, "%s: prep needs scale_h, cos_t, sin_t", who);
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

extern "C" int kd_attn_global_f32(const float* qkv, float* out, int batch, int T, int nh, int prep, const float* scale_h,
                                  const float* cos_t, const float* sin_t, float eps, void* stream) {
  if (!qkv || !out || batch <= 0 || nh <= 0 || T <= 0) return fail(KD_EINVAL, "kd_attn_global_f32: bad arguments");
  if (T > 256) return fail(KD_EINVAL, "kd_attn_global_f32: T=%d > 256 tokens not supported by the LDS-resident core", T);
  if (int e = check_prep(prep, scale_h, cos_t, sin_t, "kd_attn_global_f32")) return e;
  DenseArgs a{qkv, out, scale_h, cos_t, sin_t, batch, T, nh, 0, 0, 0, 0, eps};
  const long nb = (long)batch * nh;
  hipStream_t s = (hipStream_t)stream;
  if (T <= 64) return launch_dense<MODE_GLOBAL, 2>(a, prep, nb, "attn_global_f32", s);
  if (T <= 128) return launch_dense<MODE_GLOBAL, 4>(a, prep, nb, "attn_global_f32", s);
  return launch_dense<MODE_GLOBAL, 8>(a, prep, nb, "attn_global_f32", s);
}

extern "C" int kd_attn_window_f32(const float* qkv, float* out, int batch, int H, int W, int nh, int ws, int shift, int prep,
                                  const float* scale_h, const float* cos_t, const float* sin_t, float eps, void* stream) {
  if (!qkv || !out || batch <= 0 || nh <= 0 || H <= 0 || W <= 0) return fail(KD_EINVAL, "kd_attn_window_f32: bad arguments");
  if (ws != 8) return fail(KD_EINVAL, "kd_attn_window_f32: window_size %d unsupported (only 8)", ws);
  if ((H % ws) || (W % ws)) return fail(KD_EINVAL, "kd_attn_window_f32: grid %dx%d not divisible by the window", H, W);
  if (shift < 0 || shift >= ws) return fail(KD_EINVAL, "kd_attn_window_f32: bad shift %d", shift);
  if (int e = check_prep(prep, scale_h, cos_t, sin_t, "kd_attn_window_f32")) return e;
  DenseArgs a{qkv, out, scale_h, cos_t, sin_t, batch, H * W, nh, H, W, ws, shift, eps};
  const long nb = (long)batch * nh * (H / ws) * (W / ws);
  return launch_dense<MODE_WINDOW, 2>(a, prep, nb, "attn_window_f32", (hipStream_t)stream);
}

extern "C" int kd_attn_na2d_f32(const float* qkv, float* out, int batch, int H, int W, int nh, int ks, int prep,
                                const float* scale_h, const float* cos_t, const float* sin_t, float eps, void* stream) {
  if (!qkv || !out || batch <= 0 || nh <= 0) return fail(KD_EINVAL, "kd_attn_na2d_f32: bad arguments");
  if (ks != NA_K) return fail(KD_EINVAL, "kd_attn_na2d_f32: kernel_size %d unsupported (only 7)", ks);
  if (H < ks || W < ks) return fail(KD_EINVAL, "kd_attn_na2d_f32: grid %dx%d smaller than the %dx%d neighbourhood", H, W, ks, ks);
  if (int e = check_prep(prep, scale_h, cos_t, sin_t, "kd_attn_na2d_f32")) return e;
  NaArgs a{qkv, out, scale_h, cos_t, sin_t, batch, H, W, nh, eps};
  const long nb = (long)batch * nh * ((H + NA_TILE - 1) / NA_TILE) * ((W + NA_TILE - 1) / NA_TILE);
  hipStream_t s = (hipStream_t)stream;
  LaunchScope prof("attn_na2d_f32", 4.0 * batch * (double)H * W * nh * DH * ks * ks, 16.0 * batch * (double)H * W * nh * DH, s);
  if (prep) hipLaunchKernelGGL(attn_na2d_kernel<true>, dim3((unsigned)nb), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(attn_na2d_kernel<false>, dim3((unsigned)nb), dim3(256), 0, s, a);
  return check_launch("kd_attn_na2d_f32");
}
