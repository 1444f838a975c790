// fp32-parity ("split3") GEMM on PRE-SPLIT activations: C = epilogue(A @ W^T) with A given as two bf16 planes, A = A_hi + A_lo
// (hi = bf16_rne(a), lo = bf16_rne(a - hi), [M, K] row-major each), W as the packed split image of kd_pack_weight_bf16x3, and every
// product as hi*hi + hi*lo + lo*hi on the bf16 MFMA with fp32 accumulation.
//
// Who produces the planes: kd_norm_split_f32 (below: AdaRMSNorm / RMSNorm of the fp32 residual stream, rounded once) for the qkv and
// up projections where a row's fragments do not fit the register file (K = 512: gemm_x3.hip), and the GEGLU epilogues (c_split) for the
// down projections.  With the split done by the producer both operands of this kernel move by LDS-DMA (global_load_lds, source-side chunk
// swizzle) and its K loop is ds_read_b128 + MFMA only: 16 reads per 24 MFMAs per wave, inside the 6-slot issue shadow of an MFMA
// (profiles/r03_issue_model.md).
//
//   tile      256 rows x 128 W rows (GEGLU: 64 outputs) per 512-thread workgroup, 8 waves as 4 (rows) x 2 (features): a wave owns
//             64 rows x 64 W rows = 2 x 2 accumulator blocks, operands swapped (D = W_frag x act_frag: a lane owns an activation row)
//   stage     32 k: A_hi and A_lo images [256 rows][64 B] (chunk c of row r at position c ^ ((r >> 2) & 3)) + the 16 KiB packed W
//             stage = 48 KiB; 3-slot ring, stage kt + 2 requested at the top of stage kt (one barrier per stage: "everyone is done
//             with stage kt - 1" and "everyone's pieces of stage kt + 1 are in" are the same rendezvous), two waves per SIMD
//   epilogues in the lane that owns the row: store | + residual | qkv (cosine-sim scale + RoPE from positions, optional split-stored
//             operands for the attention cores) | GEGLU, each to fp32 [M, N] or (c_split) to bf16 hi / lo planes for the next GEMM
#include "bf16_common.h"

namespace kd {
namespace x3t {

using b16::bf16x8;
using b16::u16;
using b16::u32x4;
using b16::pack_bf16;

constexpr int BMR = 256, AIMG = BMR * 64, WSTG = 16384, WIMG = 8192, STG = 2 * AIMG + WSTG, NSTG = 3;
constexpr int LDS_BYTES = NSTG * STG + 1024;            // ring + per-head constants

__device__ __forceinline__ int swz64(int row, int c) { return row * 64 + ((c ^ ((row >> 2) & 3)) << 4); }

struct TArgs {
  const u16* Ah; const u16* Al; const char* Wp;
  float* C; u16* Ch; u16* Cl; const float* R;
  int M, N, K, n_tiles_n, nk;
  int n_heads, rows_per_sample, qkv_packed;
  const float* qk_scale; const float* pos; const float* freq;
  float out_add, eps;
  int warm;
};

#define KD_BARRIER() asm volatile("s_barrier" ::: "memory")

__device__ __forceinline__ f32x4 pack_split4(const f32x4 v) {
  const unsigned h0 = pack_bf16(v[0], v[1]), h1 = pack_bf16(v[2], v[3]);
  const unsigned l0 = pack_bf16(v[0] - b16::bf_lo(h0), v[1] - b16::bf_hi(h0)), l1 = pack_bf16(v[2] - b16::bf_lo(h1), v[3] - b16::bf_hi(h1));
  return f32x4{__uint_as_float(h0), __uint_as_float(h1), __uint_as_float(l0), __uint_as_float(l1)};
}
// one 32-feature block of the lane's row (fp32, C-layout order) -> bf16 hi / lo planes
__device__ __forceinline__ void store_block_planes(u16* hrow, u16* lrow, const float (&v)[16], int lh, bool ok) {
  float hi[16], lo[16];
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    const unsigned h = pack_bf16(v[r], v[r + 1]);
    hi[r] = b16::bf_lo(h);
    hi[r + 1] = b16::bf_hi(h);
    lo[r] = v[r] - hi[r];
    lo[r + 1] = v[r + 1] - hi[r + 1];
  }
  b16::store_block_bf16(hrow, hi, lh, ok);        // (exact: hi is a bf16 value)
  b16::store_block_bf16(lrow, lo, lh, ok);
}

template <int EPI, bool CSPLIT>
__global__ __launch_bounds__(512, 1) void gemm_x3_tiled_kernel(const TArgs p) {
  constexpr bool GEGLU = EPI == KD_EPI_GEGLU;
  constexpr int NCOL = GEGLU ? 64 : 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const auto warm = code_warm_begin<12288>((int)blockIdx.x < p.warm && tid < 64);
  const int wc = wid & 1, wr = wid >> 1;
  int tile;
  {   // XCD-aware order, n fastest: the n-tiles of one row panel run back to back on ONE L2 (bijective for any grid)
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int nt = __builtin_amdgcn_readfirstlane(tile % p.n_tiles_n), mt = __builtin_amdgcn_readfirstlane(tile / p.n_tiles_n);
  const int m0 = mt * BMR, n0 = nt * NCOL;
  const int K = p.K, nk = p.nk;

  // ---- this wave's LDS-DMA pieces of a stage: A_hi / A_lo pieces 2 wid, 2 wid + 1 (16 rows x 64 B each), W pieces 2 wid, 2 wid + 1 ----
  const char* ah_src[2];
  const char* al_src[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 16 * (2 * wid + i) + (lane >> 2);
    const int chunk = (lane & 3) ^ ((lane >> 4) & 3);         // ((row >> 2) & 3) == (lane >> 4) & 3: pieces start at multiples of 16 rows
    const size_t off = (size_t)min(m0 + row, p.M - 1) * K * 2 + chunk * 16;
    ah_src[i] = reinterpret_cast<const char*>(p.Ah) + off;
    al_src[i] = reinterpret_cast<const char*>(p.Al) + off;
  }
  const char* w_src = p.Wp + (size_t)nt * nk * WSTG + (2 * wid) * 1024 + lane * 16;
  // stages past the end re-request the last one (into a slot nobody reads any more): 6 pieces per wave per stage, always
  auto issue = [&](int kt_) {
    const int kt = min(kt_, nk - 1);
    char* st = smem + (kt_ % NSTG) * STG;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ah_src[i] + (size_t)kt * 64),
                                       (__attribute__((address_space(3))) void*)(st + (2 * wid + i) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(al_src[i] + (size_t)kt * 64),
                                       (__attribute__((address_space(3))) void*)(st + AIMG + (2 * wid + i) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_src + (size_t)kt * WSTG + i * 1024),
                                       (__attribute__((address_space(3))) void*)(st + 2 * AIMG + (2 * wid + i) * 1024), 16, 0, 0);
    }
  };

  float py[2] = {0.f, 0.f}, px[2] = {0.f, 0.f};
  if (EPI == KD_EPI_QKV) {
    float* qk_tab = reinterpret_cast<float*>(smem + NSTG * STG);
    if (tid < p.n_heads * 8) qk_tab[tid] = p.freq[tid];
    if (tid < p.n_heads) qk_tab[128 + tid] = sqrtf(p.qk_scale[tid]);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int tok = min(m0 + wr * 64 + 32 * j + l31, p.M - 1) % p.rows_per_sample;
      py[j] = p.pos[2 * tok];
      px[j] = p.pos[2 * tok + 1];
      asm volatile("" : "+v"(py[j]), "+v"(px[j]));            // consumed here as far as the compiler knows (its wait lands before the ring)
    }
  }
  code_warm_end(warm);
  issue(0);
  issue(1);

  f32x16 acc[2][2];      // [feature block i][row block j]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int a_off = (wr * 64) * 64, w_off = 2 * AIMG + (wc * 64) * 64;
  const int o0 = swz64(l31, lh), o1 = swz64(l31, 2 + lh);
  bf16x8 ahf[2][2], alf[2][2], whf[2][2], wlf[2][2];      // [chunk buffer][block]
  auto read_frags = [&](int slot, int h, bf16x8 (&fah)[2], bf16x8 (&fal)[2], bf16x8 (&fwh)[2], bf16x8 (&fwl)[2]) {
    const char* st = smem + slot * STG + (h ? o1 : o0);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      fah[u] = *reinterpret_cast<const bf16x8*>(st + a_off + u * 32 * 64);
      fal[u] = *reinterpret_cast<const bf16x8*>(st + AIMG + a_off + u * 32 * 64);
      fwh[u] = *reinterpret_cast<const bf16x8*>(st + w_off + u * 32 * 64);
      fwl[u] = *reinterpret_cast<const bf16x8*>(st + w_off + WIMG + u * 32 * 64);
    }
  };
  auto mma = [&](int b) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlf[b][i], ahf[b][j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whf[b][i], alf[b][j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whf[b][i], ahf[b][j], acc[i][j], 0, 0, 0);
  };

  // stage 0 in; its first chunk's fragments
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  KD_BARRIER();
  read_frags(0, 0, ahf[0], alf[0], whf[0], wlf[0]);
  int slot = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const int nslot = slot == NSTG - 1 ? 0 : slot + 1;
    // ---- chunk 0 (fragments in buffer 0): one MFMA, the 8 reads of chunk 1, the other 11 MFMAs -----------------------------------------
    read_frags(slot, 1, ahf[1], alf[1], whf[1], wlf[1]);
    mma(0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 11, 0);
    __builtin_amdgcn_sched_barrier(0);
    // ---- stage kt + 1 is in (requested a whole stage ago) and everyone is done with stage kt - 1: its slot takes stage kt + 2 ----------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    KD_BARRIER();
    issue(kt + 2);
    // ---- chunk 1: one MFMA, the 8 reads of the next stage's chunk 0, the other 11 MFMAs ------------------------------------------------
    read_frags(nslot, 0, ahf[0], alf[0], whf[0], wlf[0]);
    mma(1);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 11, 0);
    __builtin_amdgcn_sched_barrier(0);
    slot = nslot;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the clamped tail requests still target this workgroup's LDS

  // ---- epilogue: lane owns rows m0 + 64 wr + 32 j + l31, features n0 + 64 wc + 32 i + 8 g + 4 lh + (0..3) -----------------------------
  const char* qkc = smem + NSTG * STG;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int gm = m0 + wr * 64 + 32 * j + l31;
    const bool ok = gm < p.M;
    const int gmc = ok ? gm : p.M - 1;
    if (GEGLU) {
      // W rows 64 wc .. + 63 = [32 value rows | 32 gate rows] of outputs n0 + 32 wc ..
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 o = geglu_pair(f32x2{acc[0][j][r], acc[0][j][r + 1]} * 0.5f, f32x2{acc[1][j][r], acc[1][j][r + 1]});
        v[r] = o.x;
        v[r + 1] = o.y;
      }
      const size_t off = (size_t)gmc * p.N + n0 + 32 * wc;
      if (CSPLIT) {
        store_block_planes(p.Ch + off, p.Cl + off, v, lh, ok);
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          if (ok) *reinterpret_cast<f32x4*>(p.C + off + 8 * g + 4 * lh) = f32x4{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
      }
      continue;
    }
    if (EPI == KD_EPI_QKV) {
      const int vec = (n0 >> 6) + wc;                        // (q | k | v, head) vector index of this wave's 64 columns
      const int which = vec >= 2 * p.n_heads ? 2 : (vec >= p.n_heads ? 1 : 0), head = vec - which * p.n_heads;
      if (which < 2) {
        const f32x4 fv = *reinterpret_cast<const f32x4*>(qkc + (head * 8 + 4 * lh) * 4);
        const float qsc = *reinterpret_cast<const float*>(qkc + 512 + head * 4);
        const float fr[4] = {fv[0], fv[1], fv[2], fv[3]};
        b16::qk_prep_blocks(acc[0][j], acc[1][j], 1.0f, qsc, p.eps, py[j], px[j], fr);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const size_t off = (size_t)gmc * p.N + n0 + wc * 64 + 32 * i;
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r];
      if (EPI == KD_EPI_RESIDUAL) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 rv = *reinterpret_cast<const f32x4*>(p.R + off + 8 * g + 4 * lh);
#pragma unroll
          for (int q = 0; q < 4; ++q) v[4 * g + q] += rv[q];
        }
      } else if (EPI == KD_EPI_STORE) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += p.out_add;
      }
      if (CSPLIT) {
        store_block_planes(p.Ch + off, p.Cl + off, v, lh, ok);
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 o = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
          if (EPI == KD_EPI_QKV && p.qkv_packed) o = pack_split4(o);
          if (ok) *reinterpret_cast<f32x4*>(p.C + off + 8 * g + 4 * lh) = o;
        }
      }
    }
  }
}

// ---- AdaRMSNorm / RMSNorm of fp32 rows -> bf16 hi / lo planes (image_transformer_v2.py:98-103, :155-166) -----------------------------------
// y = x * scale[b(m)] * rsqrt(mean x^2 + eps); hi = bf16(y), lo = bf16(y - hi).  One wave per row, 8 consecutive floats per lane per
// step (K a multiple of 8).  scale == nullptr: a plain split of x (no norm).
__global__ __launch_bounds__(256) void norm_split_kernel(const float* __restrict__ x, const float* __restrict__ scale, int scale_stride,
                                                         int rows_per_sample, u16* __restrict__ hi, u16* __restrict__ lo, int M, int K, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* xr = x + (size_t)row * K;
  const float* sr = scale ? scale + (size_t)(row / rows_per_sample) * scale_stride : nullptr;
  constexpr int MAXS = 4;                      // K <= 2048 in registers
  f32x4 v0[MAXS], v1[MAXS];
  float ssq = 0.f;
#pragma unroll
  for (int s = 0; s < MAXS; ++s) {
    const int k = s * 512 + lane * 8;
    if (k < K) {
      v0[s] = *reinterpret_cast<const f32x4*>(xr + k);
      v1[s] = *reinterpret_cast<const f32x4*>(xr + k + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) ssq = fmaf(v0[s][e], v0[s][e], fmaf(v1[s][e], v1[s][e], ssq));
    }
  }
  ssq = wave_sum_xor(ssq, 64);
  const float rs = sr ? rsqrtf(ssq / (float)K + eps) : 1.0f;
#pragma unroll
  for (int s = 0; s < MAXS; ++s) {
    const int k = s * 512 + lane * 8;
    if (k < K) {
      f32x4 a = v0[s], b = v1[s];
      if (sr) {
        // the reference's order (rms_norm :98-103): scale' = scale * rsqrt(mean_sq + eps);  y = x * scale'
        a = a * (*reinterpret_cast<const f32x4*>(sr + k) * rs);
        b = b * (*reinterpret_cast<const f32x4*>(sr + k + 4) * rs);
      }
      const unsigned h0 = pack_bf16(a[0], a[1]), h1 = pack_bf16(a[2], a[3]), h2 = pack_bf16(b[0], b[1]), h3 = pack_bf16(b[2], b[3]);
      const u32x4 l = {pack_bf16(a[0] - b16::bf_lo(h0), a[1] - b16::bf_hi(h0)), pack_bf16(a[2] - b16::bf_lo(h1), a[3] - b16::bf_hi(h1)),
                       pack_bf16(b[0] - b16::bf_lo(h2), b[1] - b16::bf_hi(h2)), pack_bf16(b[2] - b16::bf_lo(h3), b[3] - b16::bf_hi(h3))};
      *reinterpret_cast<u32x4*>(hi + (size_t)row * K + k) = u32x4{h0, h1, h2, h3};
      *reinterpret_cast<u32x4*>(lo + (size_t)row * K + k) = l;
    }
  }
}

template <int EPI, bool CSPLIT>
static int launch(const TArgs& a, const char* nm, double flops, double bytes, hipStream_t s) {
  auto kern = gemm_x3_tiled_kernel<EPI, CSPLIT>;
  static LdsAttr attr_set;
  attr_set.ensure(reinterpret_cast<const void*>(kern), LDS_BYTES);
  const long tiles = (long)((a.M + BMR - 1) / BMR) * a.n_tiles_n;
  LaunchScope prof(nm, flops, bytes, s);
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(512), LDS_BYTES, s, a);
  return check_launch("kd_gemm_f32(x3 tiled)");
}

}  // namespace x3t

// Eligibility + dispatch for descriptors with pre-split A (KdGemm.a_split).  Returns 1 if not taken (the caller then fails: there is
// no other kernel for that operand format).
int gemm_x3t_try(const GemmP& d, hipStream_t s, int* rc) {
  using namespace x3t;
  if (!d.a_split || d.precision != KD_PREC_SPLIT3 || d.a_mode != KD_A_PLAIN || d.norm || !d.Wp || !d.A || !d.A_lo) return 1;
  if (d.epi != KD_EPI_STORE && d.epi != KD_EPI_QKV && d.epi != KD_EPI_GEGLU && d.epi != KD_EPI_RESIDUAL) return 1;
  const int ncol = d.epi == KD_EPI_GEGLU ? 64 : 128;
  if ((d.K & 31) || d.N % ncol) return 1;
  if (d.c_split && (!d.C || !d.C_lo || d.epi == KD_EPI_QKV)) return 1;
  if (d.epi == KD_EPI_QKV && (!d.rope_pos || !d.rope_freq || d.n_heads > 16 || d.rows_per_sample <= 0)) return 1;
  TArgs a{};
  a.Ah = reinterpret_cast<const u16*>(d.A); a.Al = reinterpret_cast<const u16*>(d.A_lo); a.Wp = reinterpret_cast<const char*>(d.Wp);
  a.C = d.C; a.Ch = reinterpret_cast<u16*>(d.C); a.Cl = reinterpret_cast<u16*>(d.C_lo); a.R = d.R;
  a.M = d.M; a.N = d.N; a.K = d.K; a.n_tiles_n = d.N / ncol; a.nk = d.K / 32;
  a.n_heads = d.n_heads; a.rows_per_sample = d.rows_per_sample > 0 ? d.rows_per_sample : d.M; a.qkv_packed = d.qkv_packed;
  a.qk_scale = d.qk_scale; a.pos = d.rope_pos; a.freq = d.rope_freq;
  a.out_add = d.out_add; a.eps = d.eps; a.warm = d.warm;
  const double n_eff = d.epi == KD_EPI_GEGLU ? 2.0 * d.N : (double)d.N;
  const double flops = 2.0 * d.M * n_eff * d.K;
  const double bytes = 4.0 * ((double)d.M * d.K + n_eff * d.K + (double)d.M * d.N) + (d.epi == KD_EPI_RESIDUAL ? 4.0 * d.M * d.N : 0.0);
  char nm[96] = "gemm_x3_tiled";
  if (prof_on()) snprintf(nm, sizeof(nm), "gemm_x3_tiled<e%d%s> M=%d N=%d K=%d", d.epi, d.c_split ? ",planes" : "", d.M, d.N, d.K);
#define KD_T(EP, CS) if (d.epi == EP && (d.c_split != 0) == CS) { *rc = launch<EP, CS>(a, nm, flops, bytes, s); return 0; }
  KD_T(KD_EPI_STORE, false) KD_T(KD_EPI_STORE, true) KD_T(KD_EPI_RESIDUAL, false) KD_T(KD_EPI_QKV, false)
  KD_T(KD_EPI_GEGLU, false) KD_T(KD_EPI_GEGLU, true)
#undef KD_T
  return 1;
}

}  // namespace kd

using namespace kd;

extern "C" int kd_norm_split_f32(const float* x, const float* scale, int scale_stride, int rows_per_sample, void* hi, void* lo, int M, int K,
                                 float eps, void* stream) {
  if (!x || !hi || !lo || M <= 0 || K <= 0 || (K & 7) || K > 2048) return fail(KD_EINVAL, "kd_norm_split_f32: bad arguments (K %% 8 == 0, K <= 2048)");
  if (scale && (rows_per_sample <= 0 || (scale_stride & 3))) return fail(KD_EINVAL, "kd_norm_split_f32: rows_per_sample > 0, scale_stride %% 4 == 0");
  LaunchScope prof("norm_split_f32", 0.0, 8.0 * M * (double)K, (hipStream_t)stream);
  hipLaunchKernelGGL(x3t::norm_split_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, scale, scale_stride,
                     rows_per_sample > 0 ? rows_per_sample : M, reinterpret_cast<b16::u16*>(hi), reinterpret_cast<b16::u16*>(lo), M, K, eps);
  return check_launch("kd_norm_split_f32");
}

KD_TEXT_PAD(gemm_x3t)      // last function of this code object: kd_common.h, code warm-up
