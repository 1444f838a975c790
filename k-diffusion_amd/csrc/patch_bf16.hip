// Image <-> token projections of the bf16 mode at patch width 4, gfx950.
//
//   patch-in   tokens[B, h, w, N] (bf16) = TokenMerge(patch) of the fp32 NCHW image (x c_in)       (image_transformer_v2.py:672,723-724)
//   patch-out  fp32 NCHW image = TokenSplitWithoutSkip(out_norm(tokens)) (x c_out + c_skip x input)            (:758-760, layers.py:90)
//
// Both are bound by HBM (84 / 59 MB at the headline shape against a 128 x 48 weight): what matters is that every access to the
// image is a 16-byte piece of a long contiguous run.  In the image the 4 pixels (nw = 0..3) of one (channel, patch row) of a token
// are 16 contiguous bytes, and the 32 tokens of a wave's chunk lie side by side: a float4 per lane is 512 contiguous bytes per
// half-wave.  The reference's feature order n = (nh * pw + nw) * chan + c scatters those 4 pixels over the feature axis, so the
// kernels RE-ORDER the weight instead of the data (the product does not care in which order k is summed or which MFMA row
// computes which feature):
//   patch-out  MFMA tile row 4 g + nw computes feature n(g, nw), g = c * ph + nh: the lane that owns a token then holds, in the
//              4 consecutive accumulator registers of a C-layout group, exactly the 4 pixels of run g -> one float4 store (and one
//              float4 load of the skip image) per run, no cross-lane traffic.  The re-ordering is the LDS row a lane reads for its
//              A-operand fragment: free.
//   patch-in   k-slot e of fragment chunk cc of lane-half lh is pixel nw = e & 3 of run 2 (2 cc + lh) + (e >> 2): a lane loads 6
//              float4 runs and has its B-operand fragments; the matching k order of the weight is written into the LDS image once
//              per (persistent) workgroup from the packed weight in HBM.
// The general path (any patch size) stays gemm_generic_bf16_kernel.
#include "bf16_common.h"

namespace kd {
namespace b16 {

struct PArgs {
  const u16* A; const float* img; const char* Wp; u16* Ct; float* Cimg; const float* R;
  const float* scale; int scale_stride, rows_per_sample; float eps;
  int M, N, K;
  int gh, gw, ph, chan;
  const float* sigma; float sigma_data;
  int warm;                  // code warm-up workgroups (kd_common.h)
};

__device__ __forceinline__ void glds16p(const void* src, void* dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

constexpr int PNW = 8;     // waves per workgroup

// ---- patch-out: N = 4 * ph * chan <= 64 features, K = 16 NC -----------------------------------------------------------------
template <int NC, bool NORM>
__global__ __launch_bounds__(PNW * 64) void unpatch4_kernel(const PArgs p) {
  constexpr int K = NC * 16, NK = NC / 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const auto warm = code_warm_begin<8192>((int)blockIdx.x < p.warm && tid < 64);                  // kd_common.h
  for (int off = wid * 1024; off < NK * WBLK; off += PNW * 1024) glds16p(p.Wp + off + lane * 16, smem + off);
  const int G = p.N >> 2;                     // runs of 4 pixels per token
  // LDS row this lane reads as MFMA row l31 of block j: the feature of tile row rho = 32 j + l31 -> run rho >> 2, pixel rho & 3
  int off4[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int rho = 32 * j + l31, g = rho >> 2, nw = rho & 3;
    const int c = g / p.ph, nh = g - c * p.ph;
    const int n = g < G ? (nh * 4 + nw) * p.chan + c : 0;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) off4[j][cc] = swz128(n, 2 * cc + lh);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  code_warm_end(warm);
  __syncthreads();

  const int chunks = (p.M + 31) >> 5;
  const int nwaves = gridDim.x * PNW, cpw = (chunks + nwaves - 1) / nwaves;
  const int ch0 = (blockIdx.x * PNW + wid) * cpw, ch1 = min(ch0 + cpw, chunks);
  const int hw = p.gh * p.gw;
  const long plane = (long)(p.gh * p.ph) * (p.gw * 4);
  for (int ch = ch0; ch < ch1; ++ch) {
    const int row = ch * 32 + l31;
    const bool ok = row < p.M;
    const int rowc = ok ? row : p.M - 1;
    u32x4 raw[NC];
    {
      const u32x4* ap = reinterpret_cast<const u32x4*>(p.A + (size_t)rowc * K + 8 * lh);
#pragma unroll
      for (int c = 0; c < NC; ++c) raw[c] = ap[2 * c];
    }
    // where this token's runs live in the image; the skip runs are requested before the products
    const int b = rowc / hw, rr = rowc - b * hw, h = rr / p.gw, w = rr - h * p.gw;
    long o[8];
    f32x4 skip[8];
    float c_out = 1.f, c_skip = 0.f;
    if (p.sigma) {
      const float sg = p.sigma[b], sd = p.sigma_data, var = sg * sg + sd * sd;
      c_out = sg * sd / sqrtf(var);
      c_skip = sd * sd / var;
    }
#pragma unroll
    for (int gi = 0; gi < 8; ++gi) {
      const int g = 2 * (gi & 3) + lh + 8 * (gi >> 2);
      if (2 * (gi & 3) + 8 * (gi >> 2) < G) {              // G is even: the same answer in both half-waves
        const int c = g / p.ph, nh = g - c * p.ph;
        o[gi] = ((long)b * p.chan + c) * plane + (long)(h * p.ph + nh) * (p.gw * 4) + w * 4;
        if (p.sigma) skip[gi] = *reinterpret_cast<const f32x4*>(p.R + o[gi]);
      }
    }
    bf16x8 a[NC];
    float rs = 1.0f;
    if (NORM) {
      const float* sp = p.scale + (size_t)(rowc / p.rows_per_sample) * p.scale_stride + 8 * lh;
      float ssq = 0.f;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(sp + 16 * c), s1 = *reinterpret_cast<const f32x4*>(sp + 16 * c + 4);
        float x[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { x[2 * e] = bf_lo(raw[c][e]); x[2 * e + 1] = bf_hi(raw[c][e]); }
#pragma unroll
        for (int e = 0; e < 8; ++e) ssq = fmaf(x[e], x[e], ssq);
        u32x4 q = {pack_bf16(x[0] * s0[0], x[1] * s0[1]), pack_bf16(x[2] * s0[2], x[3] * s0[3]),
                   pack_bf16(x[4] * s1[0], x[5] * s1[1]), pack_bf16(x[6] * s1[2], x[7] * s1[3])};
        asm volatile("" : "+v"(q));
        a[c] = __builtin_bit_cast(bf16x8, q);
      }
      ssq += __shfl_xor(ssq, 32, 64);
      rs = rsqrtf(ssq / (float)K + p.eps);
    } else {
#pragma unroll
      for (int c = 0; c < NC; ++c) a[c] = __builtin_bit_cast(bf16x8, raw[c]);
    }
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(smem + (c >> 2) * WBLK + off4[j][c & 3]);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, a[c], acc[j], 0, 0, 0);
      }
#pragma unroll
    for (int gi = 0; gi < 8; ++gi) {
      if (2 * (gi & 3) + 8 * (gi >> 2) < G) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[gi >> 2][4 * (gi & 3) + e] * rs;
          if (p.sigma) v[e] = v[e] * c_out + skip[gi][e] * c_skip;
        }
        if (ok) st16(p.Cimg + o[gi], v);
      }
    }
  }
}

// ---- patch-in: K = 4 * ph * chan <= 64 pixels of a token, N = 128 * n_tiles features --------------------------------------
__global__ __launch_bounds__(PNW * 64) void patchin4_kernel(const PArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const auto warm = code_warm_begin<6144>((int)blockIdx.x < p.warm && tid < 64);                  // kd_common.h
  const int n_tiles = p.N >> 7, G = p.K >> 2, nch = (G + 3) >> 2;
  // weight images with the k order of the fragments: 16-byte chunk q of row n = runs 2q, 2q + 1 (4 pixels each)
  for (int idx = tid; idx < n_tiles * 1024; idx += PNW * 64) {
    const int q = idx & 7, n = (idx >> 3) & 127, t = idx >> 10;
    const u16* src = reinterpret_cast<const u16*>(p.Wp + (size_t)t * WBLK);
    unsigned pk[4];
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) {
      unsigned two[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int e = 2 * e2 + u, g = 2 * q + (e >> 2), nw = e & 3;
        const int c = g / p.ph, nh = g - c * p.ph, k = (nh * 4 + nw) * p.chan + c;
        two[u] = g < G ? src[(swz128(n, k >> 3) >> 1) + (k & 7)] : 0u;
      }
      pk[e2] = two[0] | (two[1] << 16);
    }
    *reinterpret_cast<u32x4*>(smem + t * WBLK + swz128(n, q)) = u32x4{pk[0], pk[1], pk[2], pk[3]};
  }
  code_warm_end(warm);
  __syncthreads();
  int off4[4];
#pragma unroll
  for (int cc = 0; cc < 4; ++cc) off4[cc] = swz128(l31, 2 * cc + lh);

  const int chunks = (p.M + 31) >> 5;
  const int nwaves = gridDim.x * PNW, cpw = (chunks + nwaves - 1) / nwaves;
  const int ch0 = (blockIdx.x * PNW + wid) * cpw, ch1 = min(ch0 + cpw, chunks);
  const int hw = p.gh * p.gw;
  const long plane = (long)(p.gh * p.ph) * (p.gw * 4);
  for (int ch = ch0; ch < ch1; ++ch) {
    const int row = ch * 32 + l31;
    const bool ok = row < p.M;
    const int rowc = ok ? row : p.M - 1;
    const int b = rowc / hw, rr = rowc - b * hw, h = rr / p.gw, w = rr - h * p.gw;
    const float c_in = p.sigma ? 1.0f / sqrtf(p.sigma[b] * p.sigma[b] + p.sigma_data * p.sigma_data) : 1.0f;
    bf16x8 a[4];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      f32x4 v[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int g = 2 * (2 * cc + lh) + u;
        if (g < G) {
          const int c = g / p.ph, nh = g - c * p.ph;
          v[u] = *reinterpret_cast<const f32x4*>(p.img + ((long)b * p.chan + c) * plane + (long)(h * p.ph + nh) * (p.gw * 4) + w * 4);
        }
      }
      a[cc] = __builtin_bit_cast(bf16x8, u32x4{pack_bf16(v[0][0] * c_in, v[0][1] * c_in), pack_bf16(v[0][2] * c_in, v[0][3] * c_in),
                                               pack_bf16(v[1][0] * c_in, v[1][1] * c_in), pack_bf16(v[1][2] * c_in, v[1][3] * c_in)});
    }
    u16* crow = p.Ct + (size_t)rowc * p.N;
    for (int t = 0; t < n_tiles; ++t) {
      f32x16 acc[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        if (cc < nch) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bf16x8 wf = *reinterpret_cast<const bf16x8*>(smem + t * WBLK + j * 32 * 128 + off4[cc]);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, a[cc], acc[j], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = acc[j][r];
        store_block_bf16(crow + t * 128 + 32 * j, v, lh, ok);
      }
    }
  }
}


template <auto kern>                  // (the kernel is a template ARGUMENT: one LdsAttr per kernel, although all of them share one function type)
static int launch_patch(const PArgs& a, int lds, const char* nm, double flops, double bytes, hipStream_t s) {
  static LdsAttr attr_set;
  attr_set.ensure(reinterpret_cast<const void*>(kern), 64 * 1024);
  const int chunks = (a.M + 31) / 32;
  int groups = cu_count();
  const int need = (chunks + PNW - 1) / PNW;
  if (groups > need) groups = need;
  LaunchScope prof(nm, flops, bytes, s);
  hipLaunchKernelGGL(kern, dim3((unsigned)groups), dim3(PNW * 64), lds, s, a);
  return check_launch(nm);
}

// Eligibility + dispatch of both kernels.  Returns 1 if the descriptor was not taken.
int gemm_patch_try(const KdGemm& d, hipStream_t s, int* rc) {
  if (!option("patch_fast", 1) || !d.Wp || d.pw != 4 || d.ph <= 0 || d.chan <= 0 || d.gh <= 0 || d.gw <= 0) return 1;
  const int feat = 4 * d.ph * d.chan;
  PArgs a{};
  a.Wp = reinterpret_cast<const char*>(d.Wp);
  a.warm = option("code_warm", KD_CODE_WARM_DEFAULT);
  a.scale = d.scale; a.scale_stride = d.scale_stride; a.rows_per_sample = d.rows_per_sample > 0 ? d.rows_per_sample : d.M; a.eps = d.eps;
  a.M = d.M; a.N = d.N; a.K = d.K; a.gh = d.gh; a.gw = d.gw; a.ph = d.ph; a.chan = d.chan;
  a.sigma = d.sigma; a.sigma_data = d.sigma_data;
  char nm[96];
  if (d.a_mode == KD_A_PLAIN && d.epi == KD_EPI_UNPATCH_NCHW) {
    if (d.N != feat || feat > 64 || (feat & 7) || (d.K != 128 && d.K != 256) || (d.sigma && !d.R)) return 1;
    a.A = reinterpret_cast<const u16*>(d.A); a.Cimg = d.C; a.R = d.R;
    snprintf(nm, sizeof(nm), prof_on() ? "gemm_bf16_unpatch4 M=%d N=%d K=%d" : "gemm_unpatch4", d.M, d.N, d.K);
    const double flops = 2.0 * d.M * d.N * (double)d.K, bytes = 2.0 * d.M * d.K + (d.sigma ? 8.0 : 4.0) * d.M * d.N + 2.0 * d.N * d.K;
    const int lds = (d.K / 64) * WBLK;
    if (d.K == 128) *rc = d.norm ? launch_patch<unpatch4_kernel<8, true>>(a, lds, nm, flops, bytes, s) : launch_patch<unpatch4_kernel<8, false>>(a, lds, nm, flops, bytes, s);
    else *rc = d.norm ? launch_patch<unpatch4_kernel<16, true>>(a, lds, nm, flops, bytes, s) : launch_patch<unpatch4_kernel<16, false>>(a, lds, nm, flops, bytes, s);
    return 0;
  }
  if (d.a_mode == KD_A_PATCH_NCHW && d.epi == KD_EPI_STORE) {
    if (d.K != feat || feat > 64 || d.norm || (d.N & 127) || d.N > 512 || d.out_add != 0.f) return 1;
    a.img = d.A; a.Ct = reinterpret_cast<u16*>(d.C);
    snprintf(nm, sizeof(nm), prof_on() ? "gemm_bf16_patchin4 M=%d N=%d K=%d" : "gemm_patchin4", d.M, d.N, d.K);
    const double flops = 2.0 * d.M * d.N * (double)d.K, bytes = 4.0 * d.M * d.K + 2.0 * d.M * d.N + 2.0 * d.N * d.K;
    *rc = launch_patch<patchin4_kernel>(a, (d.N / 128) * WBLK, nm, flops, bytes, s);
    return 0;
  }
  return 1;
}

}  // namespace b16
}  // namespace kd

KD_TEXT_PAD(patch_bf16)      // last function of this code object: kd_common.h, code warm-up
