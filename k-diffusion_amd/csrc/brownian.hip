// Brownian-interval noise for the SDE samplers (gfx950).
//
// The reference's BrownianTreeNoiseSampler (k_diffusion/sampling.py:65-114) delegates to
// torchsde.BrownianTree: a host-side Python bisection tree that draws one full-tensor torch.randn
// per visited node.  Here every element owns a *virtual* Brownian tree that is never stored:
// W(t) is re-derived on demand by descending `depth` levels of dyadic Brownian-bridge midpoints.
// The normal deviates come from the counter-based generator Philox4x32-10 keyed by the sample's
// seed with counter (element index, tree node).  One Philox block (4 words = two Box-Muller pairs)
// serves TWO levels: the node at an even level takes r1*cos(a1); its left child r1*sin(a1), its
// right child r2*cos(a2) -- so a descent costs depth/2 + 1 Philox blocks.  Because W is a pure
// function of (seed, element, t), increments over nested / adjacent intervals are path-consistent
// by construction -- the property sample_dpmpp_sde relies on (sampling.py:572,580: (sigma_i,
// sigma_mid) then (sigma_i, sigma_{i+1})).  Below the finest level W is linearly interpolated
// (bridge mean).  The kernel is ALU-bound (Philox integer multiplies); Box-Muller uses the
// hardware v_log_f32 / v_sqrt_f32 / v_cos_f32 (argument in revolutions).
//
// End points repeat between queries (every sigma_i is an end point of 2-4 queries), so the cached
// entry point lets the caller keep W(t) tensors and skip their descents.
#include "kd_common.h"

namespace kd {

struct Philox4 { unsigned x0, x1, x2, x3; };

__device__ __forceinline__ Philox4 philox4x32_10(unsigned long long key, unsigned long long elem, unsigned long long node) {
  unsigned c0 = (unsigned)elem, c1 = (unsigned)(elem >> 32), c2 = (unsigned)node, c3 = (unsigned)(node >> 32);
  unsigned k0 = (unsigned)key, k1 = (unsigned)(key >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return {c0, c1, c2, c3};
}

// Box-Muller on hardware transcendentals: radius from a word mapped to (0, 1], cosine of `rev` revolutions
__device__ __forceinline__ float bm_radius(unsigned w) {
  const float u = (float)((w >> 8) + 1u) * 5.9604644775390625e-08f;                  // (0, 1]
  return __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u));   // sqrt(-2 ln u), v_log_f32 is log2
}
__device__ __forceinline__ float unit24(unsigned w) { return (float)(w >> 8) * 5.9604644775390625e-08f; }   // [0, 1)

__device__ __forceinline__ float brownian_w(unsigned long long key, unsigned long long elem, double t, double T0, double T1, int depth) {
  double ta = T0, tb = T1;
  const float sd0 = __builtin_amdgcn_sqrtf((float)(T1 - T0));
  const Philox4 root = philox4x32_10(key, elem, 0ull);
  float wa = 0.0f, wb = sd0 * bm_radius(root.x0) * __builtin_amdgcn_cosf(unit24(root.x1));
  float hs = 0.5f * sd0;                          // half the std of the level's interval: sqrt(len)/2
  unsigned long long node = 1ull;                 // heap index of the current even-level node
  for (int lv = 0; lv < depth; lv += 2) {
    const Philox4 x = philox4x32_10(key, elem, node);
    double tm = 0.5 * (ta + tb);
    float wm = 0.5f * (wa + wb) + hs * (bm_radius(x.x0) * __builtin_amdgcn_cosf(unit24(x.x1)));
    hs *= 0.70710678118654752f;
    const bool r0 = !(t < tm);
    if (r0) { ta = tm; wa = wm; } else { tb = tm; wb = wm; }
    if (lv + 1 >= depth) break;
    const float rad = bm_radius(r0 ? x.x2 : x.x0);
    const float rev = r0 ? unit24(x.x3) : unit24(x.x1) - 0.25f;       // sin(a) = cos(a - 1/4 turn)
    tm = 0.5 * (ta + tb);
    wm = 0.5f * (wa + wb) + hs * (rad * __builtin_amdgcn_cosf(rev));
    hs *= 0.70710678118654752f;
    const bool r1 = !(t < tm);
    if (r1) { ta = tm; wa = wm; } else { tb = tm; wb = wm; }
    node = 4ull * node + (r0 ? 2ull : 0ull) + (r1 ? 1ull : 0ull);
  }
  const float frac = (float)((t - ta) / (tb - ta));
  return wa + frac * (wb - wa);
}

// HAVE0 / HAVE1: W(t0) / W(t1) are read from w0 / w1 instead of descended; otherwise they are
// computed and, when the pointer is non-null, stored there for later queries.
template <bool HAVE0, bool HAVE1>
__global__ __launch_bounds__(256) void brownian_kernel(float* __restrict__ out, float* w0buf, float* w1buf,
                                                       const unsigned long long* __restrict__ seeds, int batch, long per_sample,
                                                       double T0, double T1, double t0, double t1, float mult, int depth) {
  const long n = (long)batch * per_sample;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long b = i / per_sample;
    const unsigned long long elem = (unsigned long long)(i - b * per_sample);
    const unsigned long long key = seeds[b];
    float w0, w1;
    if (HAVE0) w0 = w0buf[i];
    else { w0 = brownian_w(key, elem, t0, T0, T1, depth); if (w0buf) w0buf[i] = w0; }
    if (HAVE1) w1 = w1buf[i];
    else { w1 = brownian_w(key, elem, t1, T0, T1, depth); if (w1buf) w1buf[i] = w1; }
    out[i] = (w1 - w0) * mult;
  }
}

static int launch_brownian(const char* what, float* out, float* w0, float* w1, int have0, int have1, const unsigned long long* seeds, int batch,
                           long long per_sample, double T0, double T1, double t0, double t1, float mult, int depth, void* stream) {
  if (!out || !seeds || batch <= 0 || per_sample <= 0) return fail(KD_EINVAL, "%s: bad arguments", what);
  if (!(T0 < T1) || t0 < T0 || t1 > T1 || !(t0 <= t1)) return fail(KD_EINVAL, "%s: need T0 <= t0 <= t1 <= T1 (got %g %g %g %g)", what, T0, t0, t1, T1);
  if (depth < 1 || depth > 60) return fail(KD_EINVAL, "%s: depth %d out of range", what, depth);
  if ((have0 && !w0) || (have1 && !w1)) return fail(KD_EINVAL, "%s: a cached end point needs its buffer", what);
  const long n = (long)batch * per_sample;
  long blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipStream_t s = (hipStream_t)stream;
  const int traffic = 1 + (w0 ? 1 : 0) + (w1 ? 1 : 0);
  LaunchScope prof("brownian_f32", 0, 4.0 * n * traffic, s);
  const dim3 g((unsigned)blocks), b(256);
#define KD_BROWNIAN(H0, H1) hipLaunchKernelGGL((brownian_kernel<H0, H1>), g, b, 0, s, out, w0, w1, seeds, batch, (long)per_sample, T0, T1, t0, t1, mult, depth)
  if (have0 && have1) KD_BROWNIAN(true, true);
  else if (have0) KD_BROWNIAN(true, false);
  else if (have1) KD_BROWNIAN(false, true);
  else KD_BROWNIAN(false, false);
#undef KD_BROWNIAN
  return check_launch(what);
}

// Index-addressed standard normals (initial noise x0 = randn * sigma_max, sample.py:59; randn_like of the ancestral samplers,
// sampling.py:61-62): out[b, e] = scale * z(seeds[b], draw, e).  One Philox4x32-10 block per FOUR consecutive elements -- key seeds[b],
// counter (e >> 2, draw | 2^63: the top bit keeps these blocks apart from the tree's (element, node) counters under the same key) --
// turned into four normals by two Box-Muller pairs (r1 cos a1, r1 sin a1, r2 cos a2, r2 sin a2).  A sample's values depend on
// (seed, draw, element) only: not on the batch it is drawn in, the launch grid or the rank.  HBM-write bound (16 B per lane).
__global__ __launch_bounds__(256) void randn_kernel(float* __restrict__ out, const unsigned long long* __restrict__ seeds, int batch,
                                                    long per_sample, unsigned long long draw, float scale) {
  const long quads = (per_sample + 3) >> 2;
  const long n = (long)batch * quads;
  const unsigned long long node = draw | 0x8000000000000000ull;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long b = i / quads;
    const long q = i - b * quads;
    const Philox4 x = philox4x32_10(seeds[b], (unsigned long long)q, node);
    const float r1 = scale * bm_radius(x.x0), r2 = scale * bm_radius(x.x2);
    const float a1 = unit24(x.x1), a2 = unit24(x.x3);
    float4 z;
    z.x = r1 * __builtin_amdgcn_cosf(a1);
    z.y = r1 * __builtin_amdgcn_cosf(a1 - 0.25f);       // sin(a) = cos(a - 1/4 turn)
    z.z = r2 * __builtin_amdgcn_cosf(a2);
    z.w = r2 * __builtin_amdgcn_cosf(a2 - 0.25f);
    float* o = out + b * per_sample + 4 * q;
    if (4 * q + 4 <= per_sample && (per_sample & 3) == 0) {
      *reinterpret_cast<float4*>(o) = z;
    } else {
      const float v[4] = {z.x, z.y, z.z, z.w};
      for (int k = 0; k < 4 && 4 * q + k < per_sample; ++k) o[k] = v[k];
    }
  }
}

}  // namespace kd

using namespace kd;

extern "C" int kd_randn_f32(float* out, const unsigned long long* seeds, int batch, long long per_sample, unsigned long long draw, float scale,
                            void* stream) {
  if (!out || !seeds || batch <= 0 || per_sample <= 0) return fail(KD_EINVAL, "kd_randn_f32: bad arguments");
  if (draw >> 63) return fail(KD_EINVAL, "kd_randn_f32: draw must be below 2^63");
  if ((reinterpret_cast<uintptr_t>(out) & 15) != 0) return fail(KD_EINVAL, "kd_randn_f32: out must be 16-byte aligned");
  const long n = (long)batch * ((per_sample + 3) >> 2);
  long blocks = (n + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipStream_t s = (hipStream_t)stream;
  LaunchScope prof("randn_f32", 0, 4.0 * batch * per_sample, s);
  hipLaunchKernelGGL(randn_kernel, dim3((unsigned)blocks), dim3(256), 0, s, out, seeds, batch, (long)per_sample, draw, scale);
  return check_launch("kd_randn_f32");
}

extern "C" int kd_brownian_f32(float* out, const unsigned long long* seeds, int batch, long long per_sample, double T0, double T1,
                               double t0, double t1, float mult, int depth, void* stream) {
  return launch_brownian("kd_brownian_f32", out, nullptr, nullptr, 0, 0, seeds, batch, per_sample, T0, T1, t0, t1, mult, depth, stream);
}

extern "C" int kd_brownian_cached_f32(float* out, float* w0, float* w1, int have0, int have1, const unsigned long long* seeds, int batch,
                                      long long per_sample, double T0, double T1, double t0, double t1, float mult, int depth, void* stream) {
  return launch_brownian("kd_brownian_cached_f32", out, w0, w1, have0, have1, seeds, batch, per_sample, T0, T1, t0, t1, mult, depth, stream);
}
