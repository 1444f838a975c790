// bf16-mode GEMMs of the HDiT denoiser for gfx950 (see bf16_common.h for the operand formats and the swapped product).
//
// Every kernel here is built around one measured fact of this chip: a CU takes ~10-13 bytes / clock through its vector-memory
// path, L2 hits included, i.e. the bytes a workgroup pulls INTO the CU cost about as much as HBM bytes.  At the level-0 shapes
// (131 072 rows, K = 128 / 384) the weight re-streamed per row panel was as much traffic as the activations themselves, so:
//
//   wstat  "W-stationary": a persistent workgroup per CU parks its slice of the packed weight (<= 144 KiB) in LDS ONCE; its waves
//          then walk 32-row chunks of A independently: the chunk goes HBM -> registers as MFMA B-operand fragments (RMS statistics
//          and the AdaRMSNorm scale applied on the way), every weight fragment comes from LDS, the epilogue runs in the lane that
//          owns the row (no LDS, no cross-lane traffic but one half-wave exchange) and stores 16 bytes per lane.  After the
//          initial weight copy there is NO barrier and no counted wait: 8-12 waves per CU sit in different phases, so one wave's
//          loads, another's MFMAs and a third's epilogue overlap by construction.
//   (astat / tiled forms for the deeper levels follow below.)
#include "bf16_common.h"

namespace kd {
namespace b16 {

struct GArgs {
  const u16* A; const char* Wp; u16* C; const u16* R;
  const float* scale; int scale_stride, rows_per_sample; float eps;
  int M, N;                 // N = output width (GEGLU: d_ff)
  int n_tiles;              // packed n-tiles in total
  int tiles_per_slice, n_slices;
  int n_heads; const float* qk_scale; const float* pos; const float* freq;
};

// ------------------------------------------------------------------------------------------------------------------
// wstat
template <int NC /* K / 16 */, int EPI, bool NORM, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_wstat_kernel(const GArgs p) {
  constexpr int K = NC * 16, NK = NC / 4;
  constexpr bool GEGLU = EPI == KD_EPI_GEGLU;
  constexpr int NCOL = GEGLU ? 64 : 128;
  static_assert(NC % 4 == 0, "K must be a multiple of 64");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int slice = blockIdx.x % p.n_slices, grp = blockIdx.x / p.n_slices, ngrp = gridDim.x / p.n_slices;
  const int nt0 = slice * p.tiles_per_slice;
  const int ntn = min(p.tiles_per_slice, p.n_tiles - nt0);

  // ---- park this slice of the packed weight in LDS (lane-linear copy, 1 KiB per wave-instruction) --------------------------
  {
    const char* src = p.Wp + (size_t)nt0 * NK * WBLK + lane * 16;
    const int bytes = ntn * NK * WBLK;
    for (int off = wid * 1024; off < bytes; off += NW * 1024)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off),
                                       (__attribute__((address_space(3))) void*)(smem + off), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // per-lane constants of the weight fragment reads: row l31 of a 32-row block, chunk 2*cc + lh of the 64-k step
  int off4[4];
#pragma unroll
  for (int cc = 0; cc < 4; ++cc) off4[cc] = swz128(l31, 2 * cc + lh);

  const int chunks = (p.M + 31) >> 5;
  for (int ch = grp * NW + wid; ch < chunks; ch += ngrp * NW) {
    const int row = ch * 32 + l31;
    const bool ok = row < p.M;
    const int rowc = ok ? row : p.M - 1;

    // ---- this lane's row of A -> B-operand fragments: chunk c holds k = 16c + 8lh .. +7 --------------------------------------
    bf16x8 a[NC];
    float rs = 1.0f;
    {
      const u32x4* ap = reinterpret_cast<const u32x4*>(p.A + (size_t)rowc * K + 8 * lh);
      u32x4 raw[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) raw[c] = ap[2 * c];
      if (NORM) {
        const int b = rowc / p.rows_per_sample;
        const float* sp = p.scale + (size_t)b * p.scale_stride + 8 * lh;
        float ssq = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const f32x4 s0 = *reinterpret_cast<const f32x4*>(sp + 16 * c), s1 = *reinterpret_cast<const f32x4*>(sp + 16 * c + 4);
          float x[8];
#pragma unroll
          for (int u = 0; u < 4; ++u) { x[2 * u] = bf_lo(raw[c][u]); x[2 * u + 1] = bf_hi(raw[c][u]); }
#pragma unroll
          for (int u = 0; u < 8; ++u) ssq = fmaf(x[u], x[u], ssq);
          const u32x4 o = {pack_bf16(x[0] * s0[0], x[1] * s0[1]), pack_bf16(x[2] * s0[2], x[3] * s0[3]),
                           pack_bf16(x[4] * s1[0], x[5] * s1[1]), pack_bf16(x[6] * s1[2], x[7] * s1[3])};
          a[c] = __builtin_bit_cast(bf16x8, o);
        }
        ssq += __shfl_xor(ssq, 32, 64);
        rs = rsqrtf(ssq / (float)K + p.eps);
      } else {
#pragma unroll
        for (int c = 0; c < NC; ++c) a[c] = __builtin_bit_cast(bf16x8, raw[c]);
      }
    }
    float py = 0.f, px = 0.f;
    if (EPI == KD_EPI_QKV) {
      const int tok = rowc % p.rows_per_sample;
      py = p.pos[2 * tok];
      px = p.pos[2 * tok + 1];
    }

    for (int t = 0; t < ntn; ++t) {
      const int n0 = (nt0 + t) * NCOL;
      // residual operand of this tile, requested before the products
      u32x4 rraw[4][2];
      if (EPI == KD_EPI_RESIDUAL) {
#pragma unroll
        for (int j = 0; j < 4; ++j) load_block_raw(p.R + (size_t)rowc * p.N + min(n0 + 32 * j, p.N - 32), rraw[j], lh);
      }
      f32x16 acc[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
      const char* wt = smem + (size_t)t * NK * WBLK;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const char* wk = wt + (c >> 2) * WBLK + off4[c & 3];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wk + j * 32 * 128);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, a[c], acc[j], 0, 0, 0);
        }
      }
      // ---- epilogue, in the lane that owns the row ---------------------------------------------------------------------------
      u16* crow = p.C + (size_t)rowc * p.N;
      if (GEGLU) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          float v[16];
          const float rsh = 0.5f * rs;
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const f32x2 o = geglu_pair(f32x2{acc[2 * jj][r], acc[2 * jj][r + 1]} * rsh, f32x2{acc[2 * jj + 1][r], acc[2 * jj + 1][r + 1]} * rs);
            v[r] = o.x;
            v[r + 1] = o.y;
          }
          const int nb = n0 + 32 * jj;
          store_block_bf16(crow + min(nb, p.N - 32), v, lh, ok && nb < p.N);
        }
      } else if (EPI == KD_EPI_QKV) {
#pragma unroll
        for (int vv = 0; vv < 2; ++vv) {
          const int vec = (n0 >> 6) + vv;                       // (q|k|v, head) vector index of these 64 columns
          const int which = vec / p.n_heads, head = vec - which * p.n_heads;
          if (which < 2) {
            float fr[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) fr[u] = lh ? p.freq[head * 8 + 4 + u] : p.freq[head * 8 + u];
            qk_prep_blocks(acc[2 * vv], acc[2 * vv + 1], rs, sqrtf(p.qk_scale[head]), p.eps, py, px, fr);
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[2 * vv][r] *= rs; acc[2 * vv + 1][r] *= rs; }
          }
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[2 * vv + jj][r];
            store_block_bf16(crow + n0 + 64 * vv + 32 * jj, v, lh, ok);
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int nb = n0 + 32 * j;
          float v[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = acc[j][r] * rs;
          if (EPI == KD_EPI_RESIDUAL) {
            float rr[16];
            block_from_raw(rraw[j], rr);
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] += rr[r];
          }
          store_block_bf16(crow + min(nb, p.N - 32), v, lh, ok && nb < p.N);
        }
      }
    }
  }
}

static int cu_count() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  }
  return n;
}

constexpr int WSTAT_LDS_MAX = 144 * 1024;

template <int NC, int EPI, bool NORM, int NW>
static int launch_wstat(const GArgs& a, int lds, const char* nm, double flops, double bytes, hipStream_t s) {
  auto kern = gemm_wstat_kernel<NC, EPI, NORM, NW>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, WSTAT_LDS_MAX);
    attr_set = true;
  }
  const int chunks = (a.M + 31) / 32;
  int groups = cu_count() / a.n_slices;
  if (groups < 1) groups = 1;
  const int need = (chunks + NW - 1) / NW;
  if (groups > need) groups = need;
  LaunchScope prof(nm, flops, bytes, s);
  hipLaunchKernelGGL(kern, dim3((unsigned)(groups * a.n_slices)), dim3(NW * 64), lds, s, a);
  return check_launch("kd_gemm_bf16(wstat)");
}

// Eligibility + dispatch.  Returns 1 if the descriptor was not taken.
int gemm_wstat_try(const KdGemm& d, hipStream_t s, int* rc) {
  if (d.a_mode != KD_A_PLAIN || !d.Wp || !option("wstat", 1)) return 1;
  if (d.epi != KD_EPI_STORE && d.epi != KD_EPI_QKV && d.epi != KD_EPI_GEGLU && d.epi != KD_EPI_RESIDUAL) return 1;
  if (d.K != 128 && d.K != 256 && d.K != 384) return 1;
  if ((d.N & 31) || d.M < 2048) return 1;
  if (d.epi == KD_EPI_RESIDUAL && d.norm) return 1;
  if (d.epi != KD_EPI_RESIDUAL && !d.norm && d.epi != KD_EPI_STORE) return 1;
  if (d.epi == KD_EPI_STORE && d.out_add != 0.f) return 1;
  if ((d.norm || d.epi == KD_EPI_QKV) && (d.rows_per_sample <= 0 || (d.rows_per_sample & 31))) return 1;   // a 32-row chunk = one sample
  const bool geglu = d.epi == KD_EPI_GEGLU;
  const int ncol = geglu ? 64 : 128;
  if (d.epi == KD_EPI_QKV && (d.N & 127)) return 1;
  const int n_tiles = (d.N + ncol - 1) / ncol, nk = d.K / 64;
  const int max_tiles = WSTAT_LDS_MAX / (nk * WBLK);
  if (max_tiles < 1) return 1;
  const int n_slices = (n_tiles + max_tiles - 1) / max_tiles;
  if (n_slices > 2) return 1;                                  // every slice re-reads A
  const int tps = (n_tiles + n_slices - 1) / n_slices;
  GArgs a{};
  a.A = reinterpret_cast<const u16*>(d.A); a.Wp = reinterpret_cast<const char*>(d.Wp);
  a.C = reinterpret_cast<u16*>(d.C); a.R = reinterpret_cast<const u16*>(d.R);
  a.scale = d.scale; a.scale_stride = d.scale_stride; a.rows_per_sample = d.rows_per_sample > 0 ? d.rows_per_sample : d.M; a.eps = d.eps;
  a.M = d.M; a.N = d.N; a.n_tiles = n_tiles; a.tiles_per_slice = tps; a.n_slices = n_slices;
  a.n_heads = d.n_heads; a.qk_scale = d.qk_scale; a.pos = d.rope_pos; a.freq = d.rope_freq;
  const int lds = tps * nk * WBLK;
  const double n_eff = geglu ? 2.0 * d.N : (double)d.N;
  const double flops = 2.0 * d.M * n_eff * d.K;
  const double bytes = 2.0 * ((double)d.M * d.K + n_eff * d.K + (double)d.M * d.N) + (d.epi == KD_EPI_RESIDUAL ? 2.0 * d.M * d.N : 0.0);
  char nm[96] = "gemm_wstat";
  if (prof_on()) snprintf(nm, sizeof(nm), "gemm_bf16_wstat<e%d,n%d> M=%d N=%d K=%d", d.epi, d.norm, d.M, d.N, d.K);
#define KD_WS(NCV, EP, NO, NWV) { *rc = launch_wstat<NCV, EP, NO, NWV>(a, lds, nm, flops, bytes, s); return 0; }
  const int ww = option("wstat_waves", 0);        // 0: per-shape default
  if (d.K == 128 && ww != 8) {
    if (d.epi == KD_EPI_QKV) KD_WS(8, KD_EPI_QKV, true, 12)
    if (d.epi == KD_EPI_GEGLU) KD_WS(8, KD_EPI_GEGLU, true, 12)
    if (d.epi == KD_EPI_STORE && d.norm) KD_WS(8, KD_EPI_STORE, true, 12)
    if (d.epi == KD_EPI_STORE) KD_WS(8, KD_EPI_STORE, false, 12)
    if (d.epi == KD_EPI_RESIDUAL) KD_WS(8, KD_EPI_RESIDUAL, false, 12)
  } else if (d.K == 128) {
    if (d.epi == KD_EPI_QKV) KD_WS(8, KD_EPI_QKV, true, 8)
    if (d.epi == KD_EPI_GEGLU) KD_WS(8, KD_EPI_GEGLU, true, 8)
    if (d.epi == KD_EPI_STORE && d.norm) KD_WS(8, KD_EPI_STORE, true, 8)
    if (d.epi == KD_EPI_STORE) KD_WS(8, KD_EPI_STORE, false, 8)
    if (d.epi == KD_EPI_RESIDUAL) KD_WS(8, KD_EPI_RESIDUAL, false, 8)
  } else if (d.K == 256) {
    if (d.epi == KD_EPI_QKV) KD_WS(16, KD_EPI_QKV, true, 8)
    if (d.epi == KD_EPI_GEGLU) KD_WS(16, KD_EPI_GEGLU, true, 8)
    if (d.epi == KD_EPI_STORE && d.norm) KD_WS(16, KD_EPI_STORE, true, 8)
    if (d.epi == KD_EPI_STORE) KD_WS(16, KD_EPI_STORE, false, 8)
    if (d.epi == KD_EPI_RESIDUAL) KD_WS(16, KD_EPI_RESIDUAL, false, 8)
  } else {
    if (d.epi == KD_EPI_RESIDUAL) KD_WS(24, KD_EPI_RESIDUAL, false, 8)
    if (d.epi == KD_EPI_STORE && !d.norm) KD_WS(24, KD_EPI_STORE, false, 8)
  }
#undef KD_WS
  return 1;
}

// ---- one-off weight packing -------------------------------------------------------------------------------------------
// one thread per (block, row, 16-byte chunk)
__global__ __launch_bounds__(256) void pack_weight_bf16_kernel(const float* __restrict__ W, char* __restrict__ out, int N, int K, int geglu,
                                                                int n_tiles, int nk) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)n_tiles * nk * WROWS * 8;
  if (idx >= total) return;
  const int q = idx & 7, r = (idx >> 3) & (WROWS - 1);
  const long blk = idx >> 10;
  const int ks = blk % nk, nt = blk / nk;
  const int wrow = w_row_of_tile(nt, r, N, geglu != 0);
  float v[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int k = ks * WKS + q * 8 + u;
    v[u] = (wrow >= 0 && k < K) ? W[(long)wrow * K + k] : 0.f;
  }
  *reinterpret_cast<u32x4*>(out + blk * WBLK + swz128(r, q)) =
      u32x4{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
}

}  // namespace b16
}  // namespace kd

using namespace kd;

extern "C" long long kd_packed_weight_bytes_bf16(int N, int K, int geglu) {
  if (N <= 0 || K <= 0) return 0;
  const long n_tiles = (N + (geglu ? 64 : 128) - 1) / (geglu ? 64 : 128), nk = (K + b16::WKS - 1) / b16::WKS;
  return n_tiles * nk * (long long)b16::WBLK;
}

extern "C" int kd_pack_weight_bf16(const float* W, void* out, int N, int K, int geglu, void* stream) {
  if (!W || !out || N <= 0 || K <= 0) return fail(KD_EINVAL, "kd_pack_weight_bf16: bad arguments");
  const int n_tiles = (N + (geglu ? 64 : 128) - 1) / (geglu ? 64 : 128), nk = (K + b16::WKS - 1) / b16::WKS;
  const long total = (long)n_tiles * nk * b16::WROWS * 8;
  hipLaunchKernelGGL(b16::pack_weight_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W,
                     reinterpret_cast<char*>(out), N, K, geglu, n_tiles, nk);
  return check_launch("kd_pack_weight_bf16");
}

extern "C" int kd_gemm_bf16(const KdGemm* dp, void* stream) {
  if (!dp) return fail(KD_EINVAL, "kd_gemm_bf16: null descriptor");
  const KdGemm& d = *dp;
  hipStream_t s = (hipStream_t)stream;
  if (d.precision != KD_PREC_BF16) return fail(KD_EINVAL, "kd_gemm_bf16: precision must be KD_PREC_BF16");
  if (d.M <= 0 || d.N <= 0 || d.K <= 0 || (d.K & 3)) return fail(KD_EINVAL, "kd_gemm_bf16: bad M/N/K %d/%d/%d (K %% 4 != 0?)", d.M, d.N, d.K);
  if (!d.A || !d.C || !d.Wp) return fail(KD_EINVAL, "kd_gemm_bf16: null A / C / Wp (kd_pack_weight_bf16)");
  if (d.norm && (!d.scale || d.rows_per_sample <= 0 || (d.scale_stride & 3))) return fail(KD_EINVAL, "kd_gemm_bf16: norm needs scale, rows_per_sample, scale_stride%%4==0");
  if (d.epi == KD_EPI_RESIDUAL && !d.R) return fail(KD_EINVAL, "kd_gemm_bf16: residual needs R");
  if (d.epi == KD_EPI_QKV && (d.n_heads <= 0 || d.N != 3 * d.n_heads * 64 || d.rows_per_sample <= 0 || !d.qk_scale || !d.rope_pos || !d.rope_freq))
    return fail(KD_EINVAL, "kd_gemm_bf16: qkv epilogue needs N == 3*n_heads*64, rows_per_sample, qk_scale, rope_pos, rope_freq");
  int rc = 0;
  if (!b16::gemm_wstat_try(d, s, &rc)) return rc;
  return fail(KD_EINVAL, "kd_gemm_bf16: unsupported combination a_mode=%d norm=%d epi=%d M=%d N=%d K=%d", d.a_mode, d.norm, d.epi, d.M, d.N, d.K);
}
