// Fused fp32 GEMM for gfx950:  C = epilogue( prologue(A) @ W^T ),  exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32: bit-for-bit an fmaf chain, 157 TFLOP/s peak -- the parity mode of the
// reference's fp32 nn.Linear, image_transformer_v2.py:126-139).
//
// Tiling: 128x128 output tile per 256-thread workgroup (4 wave64s as 2x2, each wave a 64x64
// sub-tile = 2x2 MFMA 32x32 accumulators), K stepped by BK through a double-buffered LDS tile
// with register prefetch (one barrier per K-step).  LDS rows are padded to BK+4 floats so that
// the ds_read_b128 operand fetches (16 distinct rows per lane group) are bank-conflict free.
// Each lane fetches one float4 per operand per 4 MFMAs: MFMA step s of an 8-deep k-chunk
// consumes k = 4*(lane>>5) + s from both A and B (any consistent k permutation is a valid GEMM).
//
// Prologues / epilogues (all fused; see include/kdiff_hip.h):
//   A gather : plain | 2x2 token merge | NCHW patch gather * c_in(sigma)
//   norm     : per-row rsqrt(mean x^2 + eps) (accumulated while the A tile streams through) and
//              a per-(sample, k) scale applied on load  == AdaRMSNorm / RMSNorm
//   epilogue : store(+const) | + residual | GEGLU | 2x2 token split + lerp(skip) |
//              NCHW un-patch * c_out + x * c_skip
#include "kd_common.h"

namespace kd {

constexpr int BM = 128, BN = 128;

template <int BK> struct GemmCfg {
  static constexpr int S = BK + 4;                     // padded LDS row stride (floats)
  static constexpr int NLD = BM * BK / 4 / 256;        // float4 loads per thread per operand tile
  static constexpr int ROW_THREADS = BK / 4;           // threads that share one tile row
  static constexpr size_t LDS_BYTES = (size_t)(2 * BM * S + 2 * BN * S + BM) * sizeof(float);
};

__device__ __forceinline__ float karras_c_in(float sigma, float sd) { return 1.0f / sqrtf(sigma * sigma + sd * sd); }

template <int AMODE, bool NORM, int EPI, int BK>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const KdGemm p) {
  using Cfg = GemmCfg<BK>;
  constexpr int S = Cfg::S, NLD = Cfg::NLD, RT = Cfg::ROW_THREADS;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * BM * S;
  float* rs = Bs + 2 * BN * S;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
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

  // ---- per-thread load coordinates (same rows every K-step) ------------------------------------
  int a_row[NLD];        // tile row
  long a_off[NLD];       // element offset of (row, k=0) for plain / merge; -1 if row out of range
  int a_b[NLD];          // sample index of the row (norm scale / patch c_in)
  float a_cin[NLD];
  int a_h[NLD], a_w[NLD];
  const int kc = (tid % RT) * 4;
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    const int row = tid / RT + j * (256 / RT);
    a_row[j] = row;
    const int gm = m0 + row;
    a_off[j] = -1; a_b[j] = 0; a_cin[j] = 1.0f; a_h[j] = 0; a_w[j] = 0;
    if (gm < M) {
      if (AMODE == KD_A_PLAIN) {
        a_off[j] = (long)gm * K;
        a_b[j] = gm / p.rows_per_sample;
      } else {
        const int hw = p.gh * p.gw;
        const int b = gm / hw, r = gm % hw;
        a_b[j] = b; a_h[j] = r / p.gw; a_w[j] = r % p.gw; a_off[j] = 0;
        if (AMODE == KD_A_PATCH_NCHW && p.sigma) a_cin[j] = karras_c_in(p.sigma[b], p.sigma_data);
      }
    }
  }
  int b_row_g[NLD];      // global W row, -1 if out of range
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    const int r = tid / RT + j * (256 / RT);
    int wrow;
    if (EPI == KD_EPI_GEGLU) {
      const int n = n0 + (r >> 6) * 32 + (r & 31);
      wrow = (n < N) ? (((r >> 5) & 1) ? N + n : n) : -1;
    } else {
      wrow = (n0 + r < N) ? n0 + r : -1;
    }
    b_row_g[j] = wrow;
  }

  f32x4 ra[NLD], rb[NLD];
  float ssq[NLD];
#pragma unroll
  for (int j = 0; j < NLD; ++j) ssq[j] = 0.f;

  auto load_tiles = [&](int k0) {
    const int gk = k0 + kc;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (a_off[j] >= 0 && gk < K) {
        if (AMODE == KD_A_PLAIN) {
          v = *reinterpret_cast<const f32x4*>(p.A + a_off[j] + gk);
        } else if (AMODE == KD_A_MERGE2x2) {
          const int Cin = K >> 2;
          const int q = gk / Cin, e = gk - q * Cin;
          const long src = (((long)a_b[j] * (2 * p.gh) + 2 * a_h[j] + (q >> 1)) * (2 * p.gw) + 2 * a_w[j] + (q & 1)) * Cin + e;
          v = *reinterpret_cast<const f32x4*>(p.A + src);
        } else {  // NCHW patch gather: k = (nh*pw + nw)*chan + c
          const int Himg = p.gh * p.ph, Wimg = p.gw * p.pw;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int k = gk + u;
            if (k < K) {
              const int c = k % p.chan, q = k / p.chan;
              const int nh = q / p.pw, nw = q - nh * p.pw;
              const long src = (((long)a_b[j] * p.chan + c) * Himg + a_h[j] * p.ph + nh) * Wimg + a_w[j] * p.pw + nw;
              v[u] = p.A[src] * a_cin[j];
            }
          }
        }
        if (NORM) {
          ssq[j] += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
          const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale + (long)a_b[j] * p.scale_stride + gk);
          v = v * sc;
        }
      }
      ra[j] = v;
      f32x4 w = {0.f, 0.f, 0.f, 0.f};
      if (b_row_g[j] >= 0 && gk < K) w = *reinterpret_cast<const f32x4*>(p.W + (long)b_row_g[j] * K + gk);
      rb[j] = w;
    }
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int row = a_row[j];
      *reinterpret_cast<f32x4*>(As + (buf * BM + row) * S + kc) = ra[j];
      *reinterpret_cast<f32x4*>(Bs + (buf * BN + row) * S + kc) = rb[j];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (K + BK - 1) / BK;
  load_tiles(0);
  store_tiles(0);
  __syncthreads();

  const int frag_off = (lane & 31) * S + 4 * (lane >> 5);
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tiles((kt + 1) * BK);
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
    if (kt + 1 < nk) store_tiles(buf ^ 1);
    __syncthreads();
  }

  if (NORM) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const float s = wave_sum_xor(ssq[j], RT);
      if ((tid % RT) == 0) rs[a_row[j]] = rsqrtf(s / (float)K + p.eps);
    }
    __syncthreads();
  }

  // ---- epilogue -------------------------------------------------------------------------------
  const int col_l = lane & 31;
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
        if (gn < N) p.C[(long)gm * N + gn] = (acc[i][0][r] * rscale) * gelu_erf(acc[i][1][r] * rscale);
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

template <int AMODE, bool NORM, int EPI, int BK>
static int launch(const KdGemm& d, hipStream_t s) {
  using Cfg = GemmCfg<BK>;
  constexpr int NCOL = (EPI == KD_EPI_GEGLU) ? 64 : BN;
  const long tiles = (long)((d.M + BM - 1) / BM) * ((d.N + NCOL - 1) / NCOL);
  auto kern = gemm_f32_kernel<AMODE, NORM, EPI, BK>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
    attr_set = true;
  }
  const double n_eff = (EPI == KD_EPI_GEGLU) ? 2.0 * d.N : (double)d.N;
  char nm[96] = "gemm_f32";
  if (prof_on()) snprintf(nm, sizeof(nm), "gemm_f32<a%d,n%d,e%d> M=%d N=%d K=%d", AMODE, (int)NORM, EPI, d.M, d.N, d.K);
  LaunchScope prof(nm, 2.0 * d.M * n_eff * d.K, 4.0 * ((double)d.M * d.K + n_eff * d.K + (double)d.M * d.N), s);
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), Cfg::LDS_BYTES, s, d);
  return check_launch("kd_gemm_f32");
}

}  // namespace kd

using namespace kd;

extern "C" int kd_gemm_f32(const KdGemm* dp, void* stream) {
  if (!dp) return fail(KD_EINVAL, "kd_gemm_f32: null descriptor");
  const KdGemm& d = *dp;
  hipStream_t s = (hipStream_t)stream;
  if (d.M <= 0 || d.N <= 0 || d.K <= 0 || (d.K & 3)) return fail(KD_EINVAL, "kd_gemm_f32: bad M/N/K %d/%d/%d (K %% 4 != 0?)", d.M, d.N, d.K);
  if (!d.A || !d.W || !d.C) return fail(KD_EINVAL, "kd_gemm_f32: null A/W/C");
  if (d.norm && (!d.scale || d.rows_per_sample <= 0 || (d.scale_stride & 3))) return fail(KD_EINVAL, "kd_gemm_f32: norm needs scale, rows_per_sample, scale_stride%%4==0");
  if (d.a_mode != KD_A_PLAIN && (d.gh <= 0 || d.gw <= 0 || d.M % (d.gh * d.gw))) return fail(KD_EINVAL, "kd_gemm_f32: gather mode needs gh, gw with M %% (gh*gw) == 0");
  if (d.a_mode == KD_A_MERGE2x2 && (d.K & 15)) return fail(KD_EINVAL, "kd_gemm_f32: merge needs K %% 16 == 0");
  if (d.a_mode == KD_A_PATCH_NCHW && (d.ph <= 0 || d.pw <= 0 || d.chan <= 0 || d.K != d.ph * d.pw * d.chan)) return fail(KD_EINVAL, "kd_gemm_f32: patch gather needs K == ph*pw*chan");
  if (d.epi == KD_EPI_RESIDUAL && !d.R) return fail(KD_EINVAL, "kd_gemm_f32: residual needs R");
  if (d.epi == KD_EPI_SPLIT_LERP && (!d.R || !d.fac || (d.N & 3) || d.gh <= 0 || d.gw <= 0 || d.M % (d.gh * d.gw))) return fail(KD_EINVAL, "kd_gemm_f32: split needs R, fac, N%%4==0, gh, gw");
  if (d.epi == KD_EPI_UNPATCH_NCHW && (d.ph <= 0 || d.pw <= 0 || d.chan <= 0 || d.N != d.ph * d.pw * d.chan || d.gh <= 0 || d.gw <= 0 || d.M % (d.gh * d.gw) || (d.sigma && !d.R)))
    return fail(KD_EINVAL, "kd_gemm_f32: unpatch needs N == ph*pw*chan, gh, gw (and R when sigma is given)");
  if (d.norm && d.rows_per_sample <= 0) return fail(KD_EINVAL, "kd_gemm_f32: rows_per_sample");
  KdGemm e = d;
  if (e.rows_per_sample <= 0) e.rows_per_sample = e.M;

#define KD_CASE(AM, NO, EP) \
  if (e.a_mode == AM && (e.norm != 0) == NO && e.epi == EP) return launch<AM, NO, EP, 32>(e, s);
  KD_CASE(KD_A_PLAIN, true, KD_EPI_STORE)
  KD_CASE(KD_A_PLAIN, false, KD_EPI_STORE)
  KD_CASE(KD_A_PLAIN, false, KD_EPI_RESIDUAL)
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
