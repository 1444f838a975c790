// Brownian-interval noise for the SDE samplers (gfx950).
//
// The reference's BrownianTreeNoiseSampler (k_diffusion/sampling.py:65-114) delegates to
// torchsde.BrownianTree: a host-side Python bisection tree that draws one full-tensor torch.randn
// per visited node.  Here every element owns a *virtual* Brownian tree that is never stored:
// W(t) is re-derived on demand by descending `depth` levels of dyadic Brownian-bridge midpoints,
// each midpoint's normal deviate coming from the counter-based generator Philox4x32-10 keyed by the
// sample's seed with counter (element index, tree node).  Because W is a pure function of
// (seed, element, t), increments over nested / adjacent intervals are path-consistent by
// construction -- the property sample_dpmpp_sde relies on (sampling.py:572,580: (sigma_i, sigma_mid)
// then (sigma_i, sigma_{i+1})).  Below the finest level W is linearly interpolated (bridge mean).
#include "kd_common.h"

namespace kd {

__device__ __forceinline__ void philox_round(unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3, unsigned k0, unsigned k1) {
  const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
  const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
  const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
  c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}

// standard normal deviate for (key, element, node): Philox4x32-10 + Box-Muller on the first two words
__device__ __forceinline__ float philox_normal(unsigned long long key, unsigned long long elem, unsigned long long node) {
  unsigned c0 = (unsigned)elem, c1 = (unsigned)(elem >> 32), c2 = (unsigned)node, c3 = (unsigned)(node >> 32);
  unsigned k0 = (unsigned)key, k1 = (unsigned)(key >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c0, c1, c2, c3, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  const float u1 = (float)((c0 >> 8) + 1u) * 5.9604644775390625e-08f;   // (0, 1]
  const float u2 = (float)(c1 >> 8) * 5.9604644775390625e-08f;          // [0, 1)
  return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}

__device__ float brownian_w(unsigned long long key, unsigned long long elem, double t, double T0, double T1, int depth) {
  double ta = T0, tb = T1;
  float wa = 0.0f, wb = sqrtf((float)(T1 - T0)) * philox_normal(key, elem, 0ull);
  unsigned long long node = 1ull;
  for (int lv = 0; lv < depth; ++lv) {
    const double tm = 0.5 * (ta + tb);
    const float wm = 0.5f * (wa + wb) + 0.5f * sqrtf((float)(tb - ta)) * philox_normal(key, elem, node);
    if (t < tm) { tb = tm; wb = wm; node = 2ull * node; }
    else { ta = tm; wa = wm; node = 2ull * node + 1ull; }
  }
  const float frac = (float)((t - ta) / (tb - ta));
  return wa + frac * (wb - wa);
}

__global__ __launch_bounds__(256) void brownian_kernel(float* out, const unsigned long long* seeds, int batch, long per_sample,
                                                       double T0, double T1, double t0, double t1, float mult, int depth) {
  const long n = (long)batch * per_sample;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long b = i / per_sample;
    const unsigned long long elem = (unsigned long long)(i - b * per_sample);
    const unsigned long long key = seeds[b];
    const float w0 = brownian_w(key, elem, t0, T0, T1, depth);
    const float w1 = brownian_w(key, elem, t1, T0, T1, depth);
    out[i] = (w1 - w0) * mult;
  }
}

}  // namespace kd

using namespace kd;

extern "C" int kd_brownian_f32(float* out, const unsigned long long* seeds, int batch, long long per_sample, double T0, double T1,
                               double t0, double t1, float mult, int depth, void* stream) {
  if (!out || !seeds || batch <= 0 || per_sample <= 0) return fail(KD_EINVAL, "kd_brownian_f32: bad arguments");
  if (!(T0 < T1) || t0 < T0 || t1 > T1 || !(t0 <= t1)) return fail(KD_EINVAL, "kd_brownian_f32: need T0 <= t0 <= t1 <= T1 (got %g %g %g %g)", T0, t0, t1, T1);
  if (depth < 1 || depth > 60) return fail(KD_EINVAL, "kd_brownian_f32: depth %d out of range", depth);
  const long n = (long)batch * per_sample;
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipStream_t s = (hipStream_t)stream;
  LaunchScope prof("brownian_f32", 0, 4.0 * n, s);
  hipLaunchKernelGGL(brownian_kernel, dim3((unsigned)blocks), dim3(256), 0, s, out, seeds, batch, (long)per_sample, T0, T1, t0, t1, mult, depth);
  return check_launch("kd_brownian_f32");
}
