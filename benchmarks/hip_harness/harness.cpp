// Stand-alone check + timing driver for the bf16-mode kernels of libkdiff_hip.so (through the C ABI, no Python / torch:
// a fresh GPU box runs it seconds after the snapshot lands).  Every case compares the HIP result with an fp64 CPU
// restatement of the same op on the same bf16-rounded inputs, on a sample of rows, and times the launch with HIP events.
//   build: make -C benchmarks/hip_harness        run: benchmarks/hip_harness/harness [case-filter]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../include/kdiff_hip.h"

#define HIPCHK(x)                                                                                  \
  do {                                                                                             \
    hipError_t e_ = (x);                                                                           \
    if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } \
  } while (0)

static uint16_t f2bf(float f) {   // round to nearest even
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0;
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

template <class T>
struct DevBuf {
  T* p = nullptr; size_t n = 0;
  explicit DevBuf(size_t n_) : n(n_) { HIPCHK(hipMalloc(&p, n * sizeof(T))); }
  ~DevBuf() { (void)hipFree(p); }
  void up(const std::vector<T>& h) { HIPCHK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); }
  std::vector<T> down() const { std::vector<T> h(n); HIPCHK(hipMemcpy(h.data(), p, n * sizeof(T), hipMemcpyDeviceToHost)); return h; }
};

static std::mt19937 rng(1234);
static std::vector<float> randn(size_t n, float s = 1.f) {
  std::normal_distribution<float> d(0.f, s);
  std::vector<float> v(n);
  for (auto& x : v) x = d(rng);
  return v;
}
static std::vector<uint16_t> to_bf(const std::vector<float>& v) {
  std::vector<uint16_t> o(v.size());
  for (size_t i = 0; i < v.size(); ++i) o[i] = f2bf(v[i]);
  return o;
}

static int g_fail = 0;
static const char* g_filter = nullptr;
static bool want(const char* name) { return !g_filter || strstr(name, g_filter); }

template <class F>
static float time_us(F&& launch, int iters = 20) {
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) launch();
  HIPCHK(hipEventRecord(e1, 0));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / iters;
}

static double gelu(double x) { return 0.5 * x * (1.0 + erf(x * 0.70710678118654752440)); }

// ---------------------------------------------------------------------------------------------------------------------
struct GemmCase {
  const char* name; int M, N, K, epi, norm, rps, nh;
  int a_mode = KD_A_PLAIN, gh = 0, gw = 0;     // merge / split: coarse token grid (M = batch * gh * gw)
};

static void run_gemm_case(const GemmCase& c) {
  if (!want(c.name)) return;
  const int M = c.M, N = c.N, K = c.K;
  const bool geglu = c.epi == KD_EPI_GEGLU;
  const int NW = geglu ? 2 * N : N;      // weight rows
  const int B = (M + c.rps - 1) / c.rps;
  auto A_f = randn((size_t)M * K);
  const bool merge = c.a_mode == KD_A_MERGE2x2, split = c.epi == KD_EPI_SPLIT_LERP;
  // merge: A is the FINE grid [B, 2gh, 2gw, K/4]; the GEMM row m = (b, h, w) of the coarse grid reads k = quadrant * cin + e
  auto a_index = [&](int m, int k) -> size_t {
    if (!merge) return (size_t)m * K + k;
    const int cin = K / 4, hw = c.gh * c.gw, b = m / hw, rr = m % hw, h = rr / c.gw, w = rr % c.gw, qd = k / cin, e = k % cin;
    return (((size_t)b * (2 * c.gh) + 2 * h + (qd >> 1)) * (2 * c.gw) + 2 * w + (qd & 1)) * cin + e;
  };
  // split: C (and the skip R) is the FINE grid [B, 2gh, 2gw, N/4]; output feature n = quadrant * cout + e
  auto c_index = [&](int m, int n) -> size_t {
    if (!split) return (size_t)m * N + n;
    const int cout = N / 4, hw = c.gh * c.gw, b = m / hw, rr = m % hw, h = rr / c.gw, w = rr % c.gw, qd = n / cout, e = n % cout;
    return (((size_t)b * (2 * c.gh) + 2 * h + (qd >> 1)) * (2 * c.gw) + 2 * w + (qd & 1)) * cout + e;
  };
  const float fac_h = 0.37f;
  auto W_f = randn((size_t)NW * K, 1.0f / sqrtf((float)K));
  auto A_h = to_bf(A_f);
  std::vector<float> scale_h((size_t)B * K), pos_h((size_t)c.rps * 2), freq_h((size_t)std::max(c.nh, 1) * 8), qks_h(std::max(c.nh, 1));
  for (auto& x : scale_h) x = 1.0f + 0.3f * std::normal_distribution<float>(0, 1)(rng);
  for (auto& x : pos_h) x = std::uniform_real_distribution<float>(-1, 1)(rng);
  for (int h = 0; h < std::max(c.nh, 1); ++h) {
    qks_h[h] = 8.0f + h;
    for (int j = 0; j < 8; ++j) freq_h[h * 8 + j] = (float)(exp(log(M_PI) + (log(10 * M_PI) - log(M_PI)) * (j * std::max(c.nh, 1) + h) / (8.0 * std::max(c.nh, 1))) / (2 * M_PI));
  }
  auto R_h = to_bf(randn((size_t)M * N));

  DevBuf<uint16_t> dA(A_h.size()), dC((size_t)M * N), dR(R_h.size());
  DevBuf<float> dW(W_f.size()), dS(scale_h.size()), dP(pos_h.size()), dF(freq_h.size()), dQ(qks_h.size()), dFac(1);
  dA.up(A_h); dR.up(R_h); dW.up(W_f); dS.up(scale_h); dP.up(pos_h); dF.up(freq_h); dQ.up(qks_h); dFac.up(std::vector<float>{fac_h});
  const long long wb = kd_packed_weight_bytes_bf16(N, K, geglu);
  DevBuf<char> dWp((size_t)wb);
  if (kd_pack_weight_bf16(dW.p, dWp.p, N, K, geglu, nullptr)) { printf("%s: pack failed: %s\n", c.name, kd_last_error()); ++g_fail; return; }
  HIPCHK(hipMemset(dC.p, 0xFF, (size_t)M * N * 2));

  KdGemm d;
  memset(&d, 0, sizeof(d));
  d.M = M; d.N = N; d.K = K; d.a_mode = c.a_mode; d.epi = c.epi; d.norm = c.norm; d.gh = c.gh; d.gw = c.gw; d.fac = dFac.p;
  d.rows_per_sample = c.rps; d.scale_stride = K; d.eps = 1e-6f;
  d.A = reinterpret_cast<const float*>(dA.p); d.C = reinterpret_cast<float*>(dC.p); d.R = reinterpret_cast<const float*>(dR.p);
  d.W = dW.p; d.Wp = dWp.p; d.scale = c.norm ? dS.p : nullptr; d.precision = KD_PREC_BF16;
  d.n_heads = c.nh; d.qk_scale = dQ.p; d.rope_pos = dP.p; d.rope_freq = dF.p;
  if (int rc = kd_gemm_bf16(&d, nullptr)) { printf("%-28s REJECTED (%d): %s\n", c.name, rc, kd_last_error()); ++g_fail; return; }
  HIPCHK(hipDeviceSynchronize());
  auto C_h = dC.down();

  // ---- fp64 restatement on sampled rows -------------------------------------------------------------------------------
  std::vector<int> rows;
  for (int i = 0; i < 40; ++i) rows.push_back(i);
  for (int i = 0; i < 40; ++i) rows.push_back(M - 1 - i);
  std::uniform_int_distribution<int> rd(0, M - 1);
  for (int i = 0; i < 240; ++i) rows.push_back(rd(rng));
  double max_err = 0, max_ref = 0;
  long bad = 0;
  std::vector<double> acc(NW), out(N);
  for (int m : rows) {
    const int b = m / c.rps;
    std::vector<double> a(K);
    double ssq = 0;
    for (int k = 0; k < K; ++k) { a[k] = bf2f(A_h[a_index(m, k)]); ssq += a[k] * a[k]; }
    double rs = 1.0;
    if (c.norm) {
      rs = 1.0 / sqrt(ssq / K + 1e-6);
      for (int k = 0; k < K; ++k) a[k] = bf2f(f2bf((float)(a[k] * scale_h[(size_t)b * K + k])));   // the kernel rounds x * scale to bf16
    }
    for (int n = 0; n < NW; ++n) {
      double s = 0;
      for (int k = 0; k < K; ++k) s += a[k] * (double)bf2f(f2bf(W_f[(size_t)n * K + k]));
      acc[n] = s * rs;
    }
    if (geglu) {
      for (int n = 0; n < N; ++n) out[n] = acc[n] * gelu(acc[N + n]);
    } else if (c.epi == KD_EPI_QKV) {
      const int tok = m % c.rps;
      const double py = pos_h[2 * tok], px = pos_h[2 * tok + 1];
      for (int vec = 0; vec < N / 64; ++vec) {
        const int which = vec / c.nh, head = vec % c.nh;
        double* v = &acc[vec * 64];
        if (which < 2) {
          double ss = 0;
          for (int e = 0; e < 64; ++e) ss += v[e] * v[e];
          const double f = sqrt((double)qks_h[head]) / sqrt(ss + 1e-6);
          for (int e = 0; e < 64; ++e) v[e] *= f;
          for (int e = 0; e < 16; ++e) {
            const double th = (e < 8 ? py : px) * freq_h[head * 8 + (e & 7)] * 2 * M_PI;
            const double x1 = v[e], x2 = v[e + 16];
            v[e] = x1 * cos(th) - x2 * sin(th);
            v[e + 16] = x2 * cos(th) + x1 * sin(th);
          }
        }
        for (int e = 0; e < 64; ++e) out[vec * 64 + e] = v[e];
      }
    } else {
      for (int n = 0; n < N; ++n) {
        if (split) {
          const double skip = bf2f(R_h[c_index(m, n)]);
          out[n] = skip + (double)fac_h * (acc[n] - skip);
        } else {
          out[n] = acc[n] + (c.epi == KD_EPI_RESIDUAL ? (double)bf2f(R_h[(size_t)m * N + n]) : 0.0);
        }
      }
    }
    for (int n = 0; n < N; ++n) {
      const double got = bf2f(C_h[c_index(m, n)]);
      const double err = fabs(got - out[n]);
      max_ref = std::max(max_ref, fabs(out[n]));
      if (!(err <= 0.01 * fabs(out[n]) + 0.02)) ++bad;          // bf16 output rounding (2^-9 rel) + accumulation slack
      if (err == err) max_err = std::max(max_err, err); else max_err = 1e30;
    }
  }
  const float us = time_us([&] { kd_gemm_bf16(&d, nullptr); });
  char clk_note[256] = "";
  if (strstr(c.name, "wstat") || strstr(c.name, "clock") || strstr(c.name, "astat") || strstr(c.name, "tiled")) {       // shader clock under this kernel's load (s_memtime vs the 100 MHz s_memrealtime)
    DevBuf<unsigned long long> dClk(8);
    HIPCHK(hipMemset(dClk.p, 0, 64));
    kd_prof_clock_buffer(dClk.p);
    for (int i = 0; i < 20; ++i) kd_gemm_bf16(&d, nullptr);
    HIPCHK(hipDeviceSynchronize());
    kd_prof_clock_buffer(nullptr);
    auto ck = dClk.down();
    if (ck[3] > ck[1]) snprintf(clk_note, sizeof(clk_note), "  clk %.2f GHz", (double)(ck[2] - ck[0]) / (double)(ck[3] - ck[1]) * 0.1);
    if (ck[3] > ck[1] && ck[4] && strstr(c.name, "tiled"))
      snprintf(clk_note + strlen(clk_note), sizeof(clk_note) - strlen(clk_note), "  wg0: %llu clk = first blocks in %llu + K loop %llu (%llu steps) + epilogue %llu",
               ck[2] - ck[0], ck[4] - ck[0], ck[5] - ck[4], ck[7], ck[2] - ck[5]);
    else if (ck[3] > ck[1] && ck[4])      // astat time line of workgroup 0 (shader clocks): whole / row prologue / first tile K loop / its epilogue; ring blocks
      snprintf(clk_note + strlen(clk_note), sizeof(clk_note) - strlen(clk_note), "  wg0: %llu clk = prologue %llu + [K loop %llu + epilogue %llu] x tiles (%llu blocks)",
               ck[2] - ck[0], ck[4] - ck[0], ck[5] - ck[4], ck[6] - ck[5], ck[7]);
  }
  const double flops = 2.0 * M * (double)NW * K;
  const double bytes = 2.0 * ((double)M * K + (double)M * N + ((c.epi == KD_EPI_RESIDUAL || split) ? (double)M * N : 0.0));
  printf("%-28s M=%6d N=%4d K=%4d  max|err|=%.4g (max|ref|=%.3g) bad=%ld  %8.1f us  %7.1f TF/s  %6.0f GB/s  %s\n", c.name, M, N, K, max_err, max_ref, bad,
         us, flops / us * 1e-6, bytes / us * 1e-3, bad ? "FAIL" : "ok");
  if (clk_note[0]) printf("%s\n", clk_note);
  if (bad) ++g_fail;
}

// ---------------------------------------------------------------------------------------------------------------------
// fp8 arithmetic mode (kd_gemm_mx8, round 6) from C++: AdaRMSNorm -> up projection + GEGLU with the hidden activation leaving as e4m3 rows +
// E8M0 block scales (c_split), then the down projection + skip add with both operands in e4m3 (a_split).  The quantiser is restated HERE in
// plain C++ (a third statement next to the kernel and oracle/hdit.py): e4m3 round-to-nearest-even, power-of-two scales 2^ceil(log2(amax / 448)).
static float e4m3_rne(float x) {        // the value of x rounded to OCP e4m3 (|x| <= 448)
  if (x == 0.f) return 0.f;
  int e;
  frexpf(fabsf(x), &e);                 // |x| = m 2^e, m in [0.5, 1): binade exponent e - 1
  const int be = std::max(e - 1, -6);   // below 2^-6: the subnormal spacing
  const float step = ldexpf(1.f, be - 3);
  return nearbyintf(x / step) * step;
}
static int mx_byte(float amax) {
  const float r = amax / 448.0f;
  unsigned b;
  memcpy(&b, &r, 4);
  b = (b + 0x7FFFFFu) >> 23;
  return (int)std::min(253u, std::max(1u, b));
}
static float e4m3_decode(uint8_t v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  const float x = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -x : x;
}

static void run_mx8_case(const char* name, int M, int K, int dff, int rps) {
  if (!want(name)) return;
  const int B = (M + rps - 1) / rps;
  auto x_h = to_bf(randn((size_t)M * K));
  std::vector<float> scale_h((size_t)B * K);
  for (auto& v : scale_h) v = 1.0f + 0.3f * std::normal_distribution<float>(0, 1)(rng);
  auto Wg = randn((size_t)2 * dff * K, 1.0f / sqrtf((float)K)), Wd = randn((size_t)K * dff, 1.0f / sqrtf((float)dff));
  DevBuf<uint16_t> dX(x_h.size()), dOut((size_t)M * K);
  DevBuf<float> dS(scale_h.size()), dWg(Wg.size()), dWd(Wd.size());
  DevBuf<uint8_t> dH((size_t)M * dff), dHs((size_t)M * dff / 32);
  dX.up(x_h); dS.up(scale_h); dWg.up(Wg); dWd.up(Wd);
  DevBuf<char> dWgp((size_t)kd_packed_weight_bytes_mx8(dff, K, 1)), dWdp((size_t)kd_packed_weight_bytes_mx8(K, dff, 0));
  if (kd_pack_weight_mx8(dWg.p, dWgp.p, dff, K, 1, nullptr) || kd_pack_weight_mx8(dWd.p, dWdp.p, K, dff, 0, nullptr)) {
    printf("%s: pack failed: %s\n", name, kd_last_error()); ++g_fail; return;
  }
  KdGemm up, dn;
  memset(&up, 0, sizeof(up)); memset(&dn, 0, sizeof(dn));
  up.M = M; up.N = dff; up.K = K; up.epi = KD_EPI_GEGLU; up.norm = 1; up.rows_per_sample = rps; up.scale_stride = K; up.eps = 1e-6f;
  up.A = reinterpret_cast<const float*>(dX.p); up.W = dWg.p; up.Wp = dWgp.p; up.scale = dS.p; up.precision = KD_PREC_BF16;
  up.c_split = 1; up.C = reinterpret_cast<float*>(dH.p); up.C_lo = dHs.p;
  dn.M = M; dn.N = K; dn.K = dff; dn.epi = KD_EPI_RESIDUAL; dn.precision = KD_PREC_BF16; dn.W = dWd.p; dn.Wp = dWdp.p;
  dn.a_split = 1; dn.A = reinterpret_cast<const float*>(dH.p); dn.A_lo = dHs.p;
  dn.C = reinterpret_cast<float*>(dOut.p); dn.R = reinterpret_cast<const float*>(dX.p);
  if (!kd_gemm_mx8_supported(M, dff, K, KD_EPI_GEGLU, 1) || !kd_gemm_mx8_supported(M, K, dff, KD_EPI_RESIDUAL, 0)) { printf("%-28s not taken\n", name); return; }
  if (int rc = kd_gemm_mx8(&up, nullptr)) { printf("%-28s up REJECTED (%d): %s\n", name, rc, kd_last_error()); ++g_fail; return; }
  if (int rc = kd_gemm_mx8(&dn, nullptr)) { printf("%-28s down REJECTED (%d): %s\n", name, rc, kd_last_error()); ++g_fail; return; }
  HIPCHK(hipDeviceSynchronize());
  auto H = dH.down(); auto Hs = dHs.down(); auto Out = dOut.down();
  // quantised weights (one power-of-two scale per output channel)
  auto quant_rows = [&](const std::vector<float>& W, int rows, int cols) {
    std::vector<float> q(W.size());
    for (int n = 0; n < rows; ++n) {
      float amax = 0;
      for (int k = 0; k < cols; ++k) amax = std::max(amax, fabsf(W[(size_t)n * cols + k]));
      const float s = ldexpf(1.f, mx_byte(amax) - 127);
      for (int k = 0; k < cols; ++k) q[(size_t)n * cols + k] = e4m3_rne(W[(size_t)n * cols + k] / s) * s;
    }
    return q;
  };
  const auto Wgq = quant_rows(Wg, 2 * dff, K), Wdq = quant_rows(Wd, K, dff);
  std::vector<int> rows;
  for (int i = 0; i < 24; ++i) { rows.push_back(i); rows.push_back(M - 1 - i); }
  std::uniform_int_distribution<int> rd(0, M - 1);
  for (int i = 0; i < 80; ++i) rows.push_back(rd(rng));
  long bad_h = 0, bad_o = 0;
  double max_eh = 0, max_eo = 0;
  for (int m : rows) {
    const int b = m / rps;
    std::vector<double> uq(K), h(dff);
    double ssq = 0;
    for (int kb = 0; kb < K / 32; ++kb) {
      float u[32], amax = 0;
      for (int j = 0; j < 32; ++j) {
        const float xv = bf2f(x_h[(size_t)m * K + kb * 32 + j]);
        ssq += (double)xv * xv;
        u[j] = xv * scale_h[(size_t)b * K + kb * 32 + j];
        amax = std::max(amax, fabsf(u[j]));
      }
      const float s = ldexpf(1.f, mx_byte(amax) - 127);
      for (int j = 0; j < 32; ++j) uq[kb * 32 + j] = (double)(e4m3_rne(u[j] / s) * s);
    }
    const double rs = 1.0 / sqrt(ssq / K + 1e-6);
    for (int n = 0; n < dff; ++n) {
      double v = 0, g = 0;
      for (int k = 0; k < K; ++k) { v += uq[k] * Wgq[(size_t)n * K + k]; g += uq[k] * Wgq[(size_t)(dff + n) * K + k]; }
      h[n] = v * rs * gelu(g * rs);
    }
    // the stored hidden row decodes to within one e4m3 rounding of the restated hidden (the kernel's fp32 GEGLU value may sit a last bit away)
    std::vector<double> hdec(dff);
    for (int kb = 0; kb < dff / 32; ++kb) {
      double amax = 0;
      for (int j = 0; j < 32; ++j) amax = std::max(amax, fabs(h[kb * 32 + j]));
      const double s = ldexp(1.0, (int)Hs[(size_t)m * (dff / 32) + kb] - 127);
      for (int j = 0; j < 32; ++j) {
        hdec[kb * 32 + j] = e4m3_decode(H[(size_t)m * dff + kb * 32 + j]) * s;
        const double err = fabs(hdec[kb * 32 + j] - h[kb * 32 + j]);
        max_eh = std::max(max_eh, err / (amax + 1e-30));
        if (!(err <= amax / 15.0 + 1e-9)) ++bad_h;
      }
      if (amax > 0 && !(amax / s <= 448.0 * 1.001 && amax / s > 448.0 / 2 * 0.97)) ++bad_h;      // the block scale is the smallest power of two that fits
    }
    for (int n = 0; n < K; ++n) {
      double o = bf2f(x_h[(size_t)m * K + n]);
      for (int k = 0; k < dff; ++k) o += hdec[k] * Wdq[(size_t)n * dff + k];
      const double got = bf2f(Out[(size_t)m * K + n]), err = fabs(got - o);
      max_eo = std::max(max_eo, err);
      if (!(err <= 0.01 * fabs(o) + 0.02)) ++bad_o;
    }
  }
  const float us_up = time_us([&] { kd_gemm_mx8(&up, nullptr); }), us_dn = time_us([&] { kd_gemm_mx8(&dn, nullptr); });
  printf("%-28s M=%6d K=%4d d_ff=%4d  hidden: max err / block max %.3g bad=%ld   down + skip: max|err|=%.4g bad=%ld   up %6.1f us (%6.1f TF/s)  down %6.1f us (%6.1f TF/s)  %s\n",
         name, M, K, dff, max_eh, bad_h, max_eo, bad_o, us_up, 4.0 * M * (double)dff * K / us_up * 1e-6, us_dn, 2.0 * M * (double)dff * K / us_dn * 1e-6,
         (bad_h || bad_o) ? "FAIL" : "ok");
  if (bad_h || bad_o) ++g_fail;
}

// ---------------------------------------------------------------------------------------------------------------------
// fused feed-forward block (kd_ffn_bf16) against an fp64 restatement; the two-kernel form (GEGLU GEMM + residual GEMM) timed beside it
static void run_ffn_case(const char* name, int M, int K, int dff, int rps) {
  if (!want(name)) return;
  const int B = (M + rps - 1) / rps;
  auto X_h = to_bf(randn((size_t)M * K));
  auto Wu_f = randn((size_t)2 * dff * K, 1.0f / sqrtf((float)K)), Wd_f = randn((size_t)K * dff, 1.0f / sqrtf((float)dff));
  std::vector<float> scale_h((size_t)B * K);
  for (auto& x : scale_h) x = 1.0f + 0.3f * std::normal_distribution<float>(0, 1)(rng);
  DevBuf<uint16_t> dX(X_h.size()), dY(X_h.size()), dH((size_t)M * dff);
  DevBuf<float> dWu(Wu_f.size()), dWd(Wd_f.size()), dS(scale_h.size());
  dX.up(X_h); dWu.up(Wu_f); dWd.up(Wd_f); dS.up(scale_h);
  DevBuf<char> dPu((size_t)kd_packed_weight_bytes_bf16(dff, K, 1)), dPd((size_t)kd_packed_weight_bytes_bf16(K, dff, 2)), dPd0((size_t)kd_packed_weight_bytes_bf16(K, dff, 0));
  if (kd_pack_weight_bf16(dWu.p, dPu.p, dff, K, 1, nullptr) || kd_pack_weight_bf16(dWd.p, dPd.p, K, dff, 2, nullptr) ||
      kd_pack_weight_bf16(dWd.p, dPd0.p, K, dff, 0, nullptr)) { printf("%s: pack failed: %s\n", name, kd_last_error()); ++g_fail; return; }
  HIPCHK(hipMemset(dY.p, 0xFF, dY.n * 2));
  KdFfn f;
  memset(&f, 0, sizeof(f));
  f.x = dX.p; f.out = dY.p; f.scale = dS.p; f.scale_stride = K; f.rows_per_sample = rps; f.eps = 1e-6f;
  f.Wp_up = dPu.p; f.Wp_down = dPd.p; f.M = M; f.K = K; f.d_ff = dff;
  if (K != 128 && K != 256) { printf("%-28s not supported by kd_ffn_bf16\n", name); return; }     // (kd_ffn_bf16_supported also asks for M >= 16384: a speed rule)
  if (int rc = kd_ffn_bf16(&f, nullptr)) { printf("%-28s REJECTED (%d): %s\n", name, rc, kd_last_error()); ++g_fail; return; }
  HIPCHK(hipDeviceSynchronize());
  auto Y_h = dY.down();
  std::vector<int> rows;
  for (int i = 0; i < 40 && i < M; ++i) { rows.push_back(i); rows.push_back(M - 1 - i); }
  std::uniform_int_distribution<int> rd(0, M - 1);
  for (int i = 0; i < 200; ++i) rows.push_back(rd(rng));
  double max_err = 0, max_ref = 0;
  long bad = 0;
  std::vector<double> a(K), hid(dff);
  for (int m : rows) {
    const int b = m / rps;
    double ssq = 0;
    for (int k = 0; k < K; ++k) { a[k] = bf2f(X_h[(size_t)m * K + k]); ssq += a[k] * a[k]; }
    const double rs = 1.0 / sqrt(ssq / K + 1e-6);
    for (int k = 0; k < K; ++k) a[k] = bf2f(f2bf((float)(a[k] * scale_h[(size_t)b * K + k])));
    for (int n = 0; n < dff; ++n) {
      double v = 0, g = 0;
      for (int k = 0; k < K; ++k) {
        v += a[k] * (double)bf2f(f2bf(Wu_f[(size_t)n * K + k]));
        g += a[k] * (double)bf2f(f2bf(Wu_f[(size_t)(dff + n) * K + k]));
      }
      hid[n] = bf2f(f2bf((float)(v * rs * gelu(g * rs))));            // the hidden activation is bf16 in both forms
    }
    for (int n = 0; n < K; ++n) {
      double o = 0;
      for (int j = 0; j < dff; ++j) o += hid[j] * (double)bf2f(f2bf(Wd_f[(size_t)n * dff + j]));
      o += bf2f(X_h[(size_t)m * K + n]);
      const double got = bf2f(Y_h[(size_t)m * K + n]);
      const double err = fabs(got - o);
      max_ref = std::max(max_ref, fabs(o));
      if (!(err <= 0.01 * fabs(o) + 0.03)) ++bad;
      if (err == err) max_err = std::max(max_err, err); else max_err = 1e30;
    }
  }
  const float us = time_us([&] { kd_ffn_bf16(&f, nullptr); });
  {   // time line of workgroup 0 (shader clocks)
    DevBuf<unsigned long long> dClk(8);
    HIPCHK(hipMemset(dClk.p, 0, 64));
    kd_prof_clock_buffer(dClk.p);
    for (int i = 0; i < 5; ++i) kd_ffn_bf16(&f, nullptr);
    HIPCHK(hipDeviceSynchronize());
    kd_prof_clock_buffer(nullptr);
    auto ck = dClk.down();
    if (ck[3] > ck[1])
      printf("  clk %.2f GHz  wg0: %llu clk = rows in + norm %llu + %llu tiles %llu + skip / store %llu\n", (double)(ck[2] - ck[0]) / (double)(ck[3] - ck[1]) * 0.1,
             ck[2] - ck[0], ck[4] - ck[0], ck[7], ck[5] - ck[4], ck[2] - ck[5]);
  }
  // the two-kernel form on the same data
  KdGemm u, d;
  memset(&u, 0, sizeof(u)); memset(&d, 0, sizeof(d));
  u.M = M; u.N = dff; u.K = K; u.epi = KD_EPI_GEGLU; u.norm = 1; u.rows_per_sample = rps; u.scale_stride = K; u.eps = 1e-6f; u.precision = KD_PREC_BF16;
  u.A = reinterpret_cast<const float*>(dX.p); u.C = reinterpret_cast<float*>(dH.p); u.Wp = dPu.p; u.scale = dS.p; u.W = dWu.p;
  d.M = M; d.N = K; d.K = dff; d.epi = KD_EPI_RESIDUAL; d.rows_per_sample = rps; d.eps = 1e-6f; d.precision = KD_PREC_BF16;
  d.A = reinterpret_cast<const float*>(dH.p); d.C = reinterpret_cast<float*>(dY.p); d.R = reinterpret_cast<const float*>(dX.p); d.Wp = dPd0.p; d.W = dWd.p;
  float us2 = 0;
  if (!kd_gemm_bf16(&u, nullptr) && !kd_gemm_bf16(&d, nullptr)) us2 = time_us([&] { kd_gemm_bf16(&u, nullptr); kd_gemm_bf16(&d, nullptr); });
  const double flops = 2.0 * M * (double)K * 3.0 * dff;
  printf("%-28s M=%6d K=%4d dff=%4d  max|err|=%.4g (max|ref|=%.3g) bad=%ld  %8.1f us  %7.1f TF/s  %6.0f GB/s   (two kernels: %.1f us)  %s\n", name, M, K, dff,
         max_err, max_ref, bad, us, flops / us * 1e-6, 4.0 * M * K / us * 1e-3, us2, bad ? "FAIL" : "ok");
  if (bad) ++g_fail;
}

// patch-in (fp32 NCHW image -> bf16 tokens, * c_in) and patch-out (RMSNorm -> projection -> fp32 NCHW image, Karras c_out / c_skip)
static void run_patch_case(const char* name, int B, int C, int gh, int gw, int ps, int width) {
  if (!want(name)) return;
  const int H = gh * ps, W = gw * ps, M = B * gh * gw, Kin = C * ps * ps;
  auto img = randn((size_t)B * C * H * W, 3.0f);
  std::vector<float> sigma(B);
  for (auto& x : sigma) x = std::uniform_real_distribution<float>(0.05f, 20.f)(rng);
  const float sd = 0.5f;
  auto img_at = [&](int b, int c, int y, int x) { return img[(((size_t)b * C + c) * H + y) * W + x]; };
  DevBuf<float> dImg(img.size()), dSig(B);
  dImg.up(img); dSig.up(sigma);
  long bad = 0; double max_err = 0;
  // ---- patch in ----
  {
    auto W_f = randn((size_t)width * Kin, 1.0f / sqrtf((float)Kin));
    DevBuf<float> dW(W_f.size());
    dW.up(W_f);
    DevBuf<char> dWp((size_t)kd_packed_weight_bytes_bf16(width, Kin, 0));
    kd_pack_weight_bf16(dW.p, dWp.p, width, Kin, 0, nullptr);
    DevBuf<uint16_t> dC((size_t)M * width);
    KdGemm d; memset(&d, 0, sizeof(d));
    d.M = M; d.N = width; d.K = Kin; d.a_mode = KD_A_PATCH_NCHW; d.epi = KD_EPI_STORE; d.gh = gh; d.gw = gw; d.ph = ps; d.pw = ps; d.chan = C;
    d.sigma_data = sd; d.sigma = dSig.p; d.A = dImg.p; d.C = reinterpret_cast<float*>(dC.p); d.Wp = dWp.p; d.precision = KD_PREC_BF16; d.eps = 1e-6f;
    if (int rc = kd_gemm_bf16(&d, nullptr)) { printf("%-28s patch-in REJECTED (%d): %s\n", name, rc, kd_last_error()); ++g_fail; return; }
    HIPCHK(hipDeviceSynchronize());
    auto C_h = dC.down();
    std::uniform_int_distribution<int> rd(0, M - 1);
    for (int s = 0; s < 200; ++s) {
      const int m = s == 0 ? 0 : (s == 1 ? M - 1 : rd(rng));
      const int b = m / (gh * gw), rr = m % (gh * gw), h = rr / gw, w = rr % gw;
      const double cin = 1.0 / sqrt((double)sigma[b] * sigma[b] + (double)sd * sd);
      for (int n = 0; n < width; ++n) {
        double acc = 0;
        for (int k = 0; k < Kin; ++k) {
          const int c = k % C, q = k / C, nh = q / ps, nw = q % ps;
          acc += (double)bf2f(f2bf((float)(img_at(b, c, h * ps + nh, w * ps + nw) * (float)cin))) * bf2f(f2bf(W_f[(size_t)n * Kin + k]));
        }
        const double err = fabs(bf2f(C_h[(size_t)m * width + n]) - acc);
        if (!(err <= 0.012 * fabs(acc) + 0.02)) ++bad;
        max_err = std::max(max_err, err == err ? err : 1e30);
      }
    }
    const float us = time_us([&] { kd_gemm_bf16(&d, nullptr); });
    printf("%-28s patch-in  M=%6d N=%4d K=%3d max|err|=%.4g bad=%ld %8.1f us %s\n", name, M, width, Kin, max_err, bad, us, bad ? "FAIL" : "ok");
  }
  // ---- patch out ----
  {
    const int N = Kin, K = width;
    auto X_h = to_bf(randn((size_t)M * K));
    auto W_f = randn((size_t)N * K, 1.0f / sqrtf((float)K));
    std::vector<float> gain(K);
    for (auto& x : gain) x = 1.0f + 0.2f * std::normal_distribution<float>(0, 1)(rng);
    DevBuf<uint16_t> dX(X_h.size());
    DevBuf<float> dW(W_f.size()), dG(K), dOut(img.size());
    dX.up(X_h); dW.up(W_f); dG.up(gain);
    DevBuf<char> dWp((size_t)kd_packed_weight_bytes_bf16(N, K, 0));
    kd_pack_weight_bf16(dW.p, dWp.p, N, K, 0, nullptr);
    KdGemm d; memset(&d, 0, sizeof(d));
    d.M = M; d.N = N; d.K = K; d.a_mode = KD_A_PLAIN; d.epi = KD_EPI_UNPATCH_NCHW; d.gh = gh; d.gw = gw; d.ph = ps; d.pw = ps; d.chan = C;
    d.norm = 1; d.scale = dG.p; d.scale_stride = 0; d.rows_per_sample = gh * gw; d.eps = 1e-6f;
    d.sigma_data = sd; d.sigma = dSig.p; d.A = reinterpret_cast<const float*>(dX.p); d.C = dOut.p; d.R = dImg.p; d.Wp = dWp.p; d.precision = KD_PREC_BF16;
    if (int rc = kd_gemm_bf16(&d, nullptr)) { printf("%-28s patch-out REJECTED (%d): %s\n", name, rc, kd_last_error()); ++g_fail; return; }
    HIPCHK(hipDeviceSynchronize());
    auto O_h = dOut.down();
    std::uniform_int_distribution<int> rd(0, M - 1);
    double max_err2 = 0;
    for (int s = 0; s < 200; ++s) {
      const int m = s == 0 ? 0 : (s == 1 ? M - 1 : rd(rng));
      const int b = m / (gh * gw), rr = m % (gh * gw), h = rr / gw, w = rr % gw;
      const double var = (double)sigma[b] * sigma[b] + (double)sd * sd, c_out = sigma[b] * sd / sqrt(var), c_skip = sd * sd / var;
      double ssq = 0;
      for (int k = 0; k < K; ++k) { const double x = bf2f(X_h[(size_t)m * K + k]); ssq += x * x; }
      const double rs = 1.0 / sqrt(ssq / K + 1e-6);
      for (int n = 0; n < N; ++n) {
        double acc = 0;
        for (int k = 0; k < K; ++k) acc += (double)bf2f(f2bf(bf2f(X_h[(size_t)m * K + k]) * gain[k])) * bf2f(f2bf(W_f[(size_t)n * K + k]));
        const int c = n % C, q = n / C, nh = q / ps, nw = q % ps;
        const size_t o = (((size_t)b * C + c) * H + h * ps + nh) * W + w * ps + nw;
        const double ref = acc * rs * c_out + img[o] * c_skip;
        const double err = fabs(O_h[o] - ref);
        if (!(err <= 2e-3 * fabs(ref) + 2e-2)) ++bad;
        max_err2 = std::max(max_err2, err == err ? err : 1e30);
      }
    }
    const float us = time_us([&] { kd_gemm_bf16(&d, nullptr); });
    printf("%-28s patch-out M=%6d N=%4d K=%3d max|err|=%.4g bad=%ld %8.1f us %s\n", name, M, N, K, max_err2, bad, us, bad ? "FAIL" : "ok");
  }
  if (bad) ++g_fail;
}

// ---------------------------------------------------------------------------------------------------------------------
// attention cores: mode 0 global, 1 window (ws, shift), 2 neighbourhood (ks)
struct AttnCase { const char* name; int mode, B, H, W, nh, p0, p1; };

static void run_attn_case(const AttnCase& c) {
  if (!want(c.name)) return;
  const int T = c.H * c.W, nh = c.nh, B = c.B;
  const size_t rowlen = (size_t)3 * nh * 64;
  std::vector<float> qkv_f((size_t)B * T * rowlen);
  {
    std::normal_distribution<float> d(0.f, 1.f);
    for (size_t t = 0; t < (size_t)B * T; ++t)
      for (int which = 0; which < 3; ++which)
        for (int h = 0; h < nh; ++h) {
          float* v = &qkv_f[t * rowlen + (size_t)(which * nh + h) * 64];
          double ss = 0;
          for (int e = 0; e < 64; ++e) { v[e] = d(rng); ss += (double)v[e] * v[e]; }
          if (which < 2) { const float f = (float)(sqrt(10.0) / sqrt(ss)); for (int e = 0; e < 64; ++e) v[e] *= f; }   // cosine-sim scaled, like the model
        }
  }
  auto qkv_h = to_bf(qkv_f);
  DevBuf<uint16_t> dQ(qkv_h.size()), dO((size_t)B * T * nh * 64);
  dQ.up(qkv_h);
  HIPCHK(hipMemset(dO.p, 0xFF, dO.n * 2));
  auto launch = [&]() -> int {
    if (c.mode == 0) return kd_attn_global_bf16(dQ.p, dO.p, B, T, nh, nullptr);
    if (c.mode == 1) return kd_attn_window_bf16(dQ.p, dO.p, B, c.H, c.W, nh, c.p0, c.p1, nullptr);
    return kd_attn_na2d_bf16(dQ.p, dO.p, B, c.H, c.W, nh, c.p0, nullptr);
  };
  if (int rc = launch()) { printf("%-28s REJECTED (%d): %s\n", c.name, rc, kd_last_error()); ++g_fail; return; }
  HIPCHK(hipDeviceSynchronize());
  auto O_h = dO.down();
  auto at = [&](int b, int tok, int which, int h, int e) -> double { return bf2f(qkv_h[((size_t)b * T + tok) * rowlen + (size_t)(which * nh + h) * 64 + e]); };
  std::uniform_int_distribution<int> rb(0, B - 1), rt(0, T - 1), rh(0, nh - 1);
  double max_err = 0;
  long bad = 0;
  const int nq = 300;
  for (int s = 0; s < nq; ++s) {
    const int b = s < 4 ? 0 : rb(rng), h = rh(rng);
    int tok = rt(rng);
    if (s == 0) tok = 0; if (s == 1) tok = T - 1; if (s == 2) tok = c.W - 1; if (s == 3) tok = T - c.W;
    const int qi = tok / c.W, qj = tok % c.W;
    std::vector<int> keys;
    if (c.mode == 0) {
      for (int k = 0; k < T; ++k) keys.push_back(k);
    } else if (c.mode == 1) {
      const int ws = c.p0, sh = c.p1;
      auto info = [&](int i, int j, int& win, int& reg) {
        const int ri = (i + sh) % c.H, rj = (j + sh) % c.W, wi = ri / ws, wj = rj / ws;
        win = wi * (c.W / ws) + wj;
        reg = sh ? (((wi == 0 && (ri % ws) < sh) ? 2 : 0) + ((wj == 0 && (rj % ws) < sh) ? 1 : 0)) : 0;
      };
      int qw, qr;
      info(qi, qj, qw, qr);
      for (int k = 0; k < T; ++k) { int kw, kr; info(k / c.W, k % c.W, kw, kr); if (kw == qw && kr == qr) keys.push_back(k); }
    } else {
      const int ks = c.p0;
      const int si = std::max(0, std::min(qi - ks / 2, c.H - ks)), sj = std::max(0, std::min(qj - ks / 2, c.W - ks));
      for (int i = si; i < si + ks; ++i) for (int j = sj; j < sj + ks; ++j) keys.push_back(i * c.W + j);
    }
    std::vector<double> sc(keys.size());
    double m = -1e300;
    for (size_t k = 0; k < keys.size(); ++k) {
      double d = 0;
      for (int e = 0; e < 64; ++e) d += at(b, tok, 0, h, e) * at(b, keys[k], 1, h, e);
      sc[k] = d; m = std::max(m, d);
    }
    double l = 0;
    for (auto& x : sc) { x = exp(x - m); l += x; }
    for (int e = 0; e < 64; ++e) {
      double o = 0;
      for (size_t k = 0; k < keys.size(); ++k) o += sc[k] * at(b, keys[k], 2, h, e);
      o /= l;
      const double got = bf2f(O_h[((size_t)b * T + tok) * (nh * 64) + h * 64 + e]);
      const double err = fabs(got - o);
      if (!(err <= 0.02 * fabs(o) + 0.02)) {
        if (bad < 6 || (bad % 64 == 0 && bad < 1024)) printf("    bad: b=%d y=%d x=%d head=%d e=%d got %.4f want %.4f\n", b, qi, qj, h, e, got, o);
        ++bad;
      }
      if (err == err) max_err = std::max(max_err, err); else max_err = 1e30;
    }
  }
  const float us = time_us([&] { launch(); });
  const double bytes = 2.0 * (double)B * T * nh * 64 * 4;
  printf("%-28s B=%3d %3dx%-3d nh=%d p=%d,%d  max|err|=%.4g bad=%ld  %8.1f us  %6.0f GB/s  %s\n", c.name, B, c.H, c.W, nh, c.p0, c.p1, max_err, bad, us,
         bytes / us * 1e-3, bad ? "FAIL" : "ok");
  if (bad) ++g_fail;
}

// ---------------------------------------------------------------------------------------------------------------------
// memory-path probes (what does a CU take in per clock, and what does a row-per-lane store pattern cost)
__global__ __launch_bounds__(1024) void probe_glds(const char* src, int bytes_per_block, int iters, int* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const char* base = src + (size_t)(blockIdx.x & 7) * bytes_per_block;        // 8 distinct L2-resident regions
  for (int it = 0; it < iters; ++it) {
    for (int off = wid * 1024; off < bytes_per_block; off += nw * 1024)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off + lane * 16),
                                       (__attribute__((address_space(3))) void*)(lds + off), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x] = lds[iters & 15];
}
__global__ __launch_bounds__(1024) void probe_vload(const uint4* src, int vec_per_block, int iters, int* sink) {
  const uint4* base = src + (size_t)(blockIdx.x & 7) * vec_per_block;
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    for (int i = threadIdx.x; i < vec_per_block; i += blockDim.x * 4) {
      uint4 a = base[i], b = base[min(i + (int)blockDim.x, vec_per_block - 1)], c = base[min(i + 2 * (int)blockDim.x, vec_per_block - 1)],
            d = base[min(i + 3 * (int)blockDim.x, vec_per_block - 1)];
      acc += a.x ^ b.y ^ c.z ^ d.w;
    }
    asm volatile("" ::: "memory");
  }
  if (acc == 0x12345u) sink[blockIdx.x] = (int)acc;
}
// HBM streaming read, every block its own region
__global__ __launch_bounds__(256) void probe_stream(const uint4* src, size_t vec_total, int* sink) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < vec_total; i += (size_t)gridDim.x * 256 * 4) {
    const size_t s = (size_t)gridDim.x * 256;
    uint4 a = src[i], b = src[std::min(i + s, vec_total - 1)], c = src[std::min(i + 2 * s, vec_total - 1)], d = src[std::min(i + 3 * s, vec_total - 1)];
    acc += a.x ^ b.y ^ c.z ^ d.w;
  }
  if (acc == 0x12345u) sink[blockIdx.x] = (int)acc;
}
// stores: MODE 0 = fully coalesced 16 B per lane; MODE 1 = "row per lane": lane l31 owns a row of `row_bytes`, the two half-waves
// write adjacent 16-byte pieces (32 contiguous bytes per row per instruction), successive instructions walk along the row
template <int MODE>
__global__ __launch_bounds__(256) void probe_store(uint4* dst, size_t rows, int row_bytes) {
  const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
  const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * 256) >> 6;
  const uint4 v = {1u, 2u, 3u, (unsigned)lane};
  const int vec_per_row = row_bytes / 16;
  for (size_t chunk = wave; chunk * 32 < rows; chunk += nwaves) {
    uint4* base = dst + chunk * 32 * vec_per_row;
    if (MODE == 0) {
      for (int i = lane; i < 32 * vec_per_row; i += 64) base[i] = v;
    } else {
      for (int i = 0; i < vec_per_row; i += 2) base[(size_t)l31 * vec_per_row + i + lh] = v;
    }
  }
}

static void run_probes() {
  if (!want("probe")) return;
  int dev = 0, cus = 256, khz = 0;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, dev);
  printf("device: %d CUs, %d MHz\n", cus, khz / 1000);
  DevBuf<int> sink(4096);
  {
    const int region = 96 * 1024;
    DevBuf<char> src((size_t)8 * region);
    HIPCHK(hipMemset(src.p, 1, (size_t)8 * region));
    for (int nw : {4, 8, 12, 16}) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(probe_glds), hipFuncAttributeMaxDynamicSharedMemorySize, region);
      const int iters = 200;
      const float us = time_us([&] { hipLaunchKernelGGL(probe_glds, dim3(cus), dim3(nw * 64), region, 0, src.p, region, iters, sink.p); }, 5);
      const double bytes = (double)cus * region * iters;
      printf("probe glds   L2-resident, %2d waves/CU: %8.1f us  %7.2f TB/s  = %5.1f B/clk/CU @2.4GHz\n", nw, us, bytes / us * 1e-6, bytes / us * 1e-3 / cus / 2.4);
    }
    for (int nw : {4, 8, 16}) {
      const int iters = 200;
      const float us = time_us([&] { hipLaunchKernelGGL(probe_vload, dim3(cus), dim3(nw * 64), 0, 0, reinterpret_cast<const uint4*>(src.p), region / 16, iters, sink.p); }, 5);
      const double bytes = (double)cus * region * iters;
      printf("probe vload  L2-resident, %2d waves/CU: %8.1f us  %7.2f TB/s  = %5.1f B/clk/CU @2.4GHz\n", nw, us, bytes / us * 1e-6, bytes / us * 1e-3 / cus / 2.4);
    }
  }
  {
    const size_t bytes = (size_t)1 << 30;
    DevBuf<char> big(bytes);
    HIPCHK(hipMemset(big.p, 1, bytes));
    for (int blocks : {1024, 2048, 8192}) {
      const float us = time_us([&] { hipLaunchKernelGGL(probe_stream, dim3(blocks), dim3(256), 0, 0, reinterpret_cast<const uint4*>(big.p), bytes / 16, sink.p); }, 5);
      printf("probe stream HBM read 1 GiB, %5d blocks: %8.1f us  %7.2f TB/s\n", blocks, us, (double)bytes / us * 1e-6);
    }
    for (int row_bytes : {256, 768}) {
      const size_t rows = bytes / 2 / row_bytes;
      const float us0 = time_us([&] { hipLaunchKernelGGL(probe_store<0>, dim3(2048), dim3(256), 0, 0, reinterpret_cast<uint4*>(big.p), rows, row_bytes); }, 5);
      const float us1 = time_us([&] { hipLaunchKernelGGL(probe_store<1>, dim3(2048), dim3(256), 0, 0, reinterpret_cast<uint4*>(big.p), rows, row_bytes); }, 5);
      printf("probe store  512 MiB, rows of %4d B: coalesced %8.1f us %6.2f TB/s | row-per-lane %8.1f us %6.2f TB/s\n", row_bytes, us0,
             (double)rows * row_bytes / us0 * 1e-6, us1, (double)rows * row_bytes / us1 * 1e-6);
    }
  }
}

int main(int argc, char** argv) {
  if (argc > 1) g_filter = argv[1];
  printf("libkdiff_hip version %d\n", kd_version());
  run_probes();
  const GemmCase cases[] = {
      {"wstat L0 qkv", 131072, 384, 128, KD_EPI_QKV, 1, 4096, 2},
      {"wstat L0 geglu", 131072, 384, 128, KD_EPI_GEGLU, 1, 4096, 0},
      {"wstat L0 out+res", 131072, 128, 128, KD_EPI_RESIDUAL, 0, 4096, 0},
      {"wstat L0 down+res", 131072, 128, 384, KD_EPI_RESIDUAL, 0, 4096, 0},
      {"wstat L1 qkv", 32768, 768, 256, KD_EPI_QKV, 1, 1024, 4},
      {"wstat L1 geglu", 32768, 768, 256, KD_EPI_GEGLU, 1, 1024, 0},
      {"wstat L1 out+res", 32768, 256, 256, KD_EPI_RESIDUAL, 0, 1024, 0},
      {"wstat L2 qkv", 8192, 1536, 512, KD_EPI_QKV, 1, 256, 8},
      {"wstat L2 geglu", 8192, 1536, 512, KD_EPI_GEGLU, 1, 256, 0},
      {"wstat L2 out+res", 8192, 512, 512, KD_EPI_RESIDUAL, 0, 256, 0},
      {"wstat ragged qkv", 4128, 384, 128, KD_EPI_QKV, 1, 96, 2},
      {"wstat ragged geglu", 4128, 96, 128, KD_EPI_GEGLU, 1, 96, 0},
      {"tiled L1 down+res", 32768, 256, 768, KD_EPI_RESIDUAL, 0, 1024, 0},
      {"tiled L2 down+res", 8192, 512, 1536, KD_EPI_RESIDUAL, 0, 256, 0},
      {"tiled merge0", 32768, 256, 512, KD_EPI_STORE, 0, 1024, 0, KD_A_MERGE2x2, 32, 32},
      {"tiled merge1", 8192, 512, 1024, KD_EPI_STORE, 0, 256, 0, KD_A_MERGE2x2, 16, 16},
      {"tiled split1", 8192, 1024, 512, KD_EPI_SPLIT_LERP, 0, 256, 0, KD_A_PLAIN, 16, 16},
      {"tiled split0", 32768, 512, 256, KD_EPI_SPLIT_LERP, 0, 1024, 0, KD_A_PLAIN, 32, 32},
      {"tiled ragged store", 1000, 96, 192, KD_EPI_STORE, 0, 1000, 0},
      {"tiled ragged res", 300, 160, 64, KD_EPI_RESIDUAL, 0, 300, 0},
  };
  for (const auto& c : cases) run_gemm_case(c);
  run_mx8_case("mx8 L1 ff", 32768, 256, 768, 1024);
  run_mx8_case("mx8 L2 ff", 8192, 512, 1536, 256);
  run_mx8_case("mx8 ragged ff", 1000, 256, 768, 50);
  if (want("astat")) {
    const GemmCase ca[] = {
        {"astat L1 qkv", 32768, 768, 256, KD_EPI_QKV, 1, 1024, 4},
        {"astat L1 geglu", 32768, 768, 256, KD_EPI_GEGLU, 1, 1024, 0},
        {"astat L1 store", 32768, 256, 256, KD_EPI_STORE, 1, 1024, 0},
        {"astat L2 qkv", 8192, 1536, 512, KD_EPI_QKV, 1, 256, 8},
        {"astat L2 geglu", 8192, 1536, 512, KD_EPI_GEGLU, 1, 256, 0},
        {"astat cifar L1 qkv", 4096, 1536, 512, KD_EPI_QKV, 1, 64, 8},
        {"astat ragged qkv", 1000, 768, 256, KD_EPI_QKV, 1, 50, 4},
        {"astat ragged geglu", 1000, 192, 256, KD_EPI_GEGLU, 1, 50, 0},
    };
    for (const auto& c : ca) run_gemm_case(c);
    for (int rows : {128, 256, 128, 256}) {
      kd_set_option("astat_rows", rows);
      printf("-- astat_rows = %d\n", rows);
      for (int i = 0; i < 5; ++i) run_gemm_case(ca[i]);
    }
    kd_set_option("astat_rows", 0);
    for (int sp : {1, 2, 3, 6}) {
      kd_set_option("astat_splits", sp);
      printf("-- astat_splits = %d\n", sp);
      const GemmCase cs[] = {
          {"astat L1 qkv", 32768, 768, 256, KD_EPI_QKV, 1, 1024, 4},
          {"astat L1 geglu", 32768, 768, 256, KD_EPI_GEGLU, 1, 1024, 0},
          {"astat L2 qkv", 8192, 1536, 512, KD_EPI_QKV, 1, 256, 8},
          {"astat L2 geglu", 8192, 1536, 512, KD_EPI_GEGLU, 1, 256, 0},
      };
      for (const auto& c : cs) run_gemm_case(c);
    }
    kd_set_option("astat_splits", 0);
  }
  if (want("tiled")) {      // the same residual shapes forced through the tiled kernel, both tile heights
    kd_set_option("wstat", 0);
    for (int bm : {128, 256}) {
      kd_set_option("tiled_bm", bm);
      printf("-- tiled_bm = %d, wstat off\n", bm);
      const GemmCase ct[] = {
          {"tiled L0 out+res", 131072, 128, 128, KD_EPI_RESIDUAL, 0, 4096, 0},
          {"tiled L0 down+res", 131072, 128, 384, KD_EPI_RESIDUAL, 0, 4096, 0},
          {"tiled L1 out+res", 32768, 256, 256, KD_EPI_RESIDUAL, 0, 1024, 0},
          {"tiled L1 down+res", 32768, 256, 768, KD_EPI_RESIDUAL, 0, 1024, 0},
          {"tiled L2 out+res", 8192, 512, 512, KD_EPI_RESIDUAL, 0, 256, 0},
          {"tiled L2 down+res", 8192, 512, 1536, KD_EPI_RESIDUAL, 0, 256, 0},
      };
      for (const auto& c : ct) run_gemm_case(c);
    }
    kd_set_option("tiled_bm", 0);
    for (int deep : {0, 1, 0, 1}) {
      kd_set_option("tiled_deep", deep);
      printf("-- tiled_deep = %d (4-slot ring for grids of at most one tile per CU)\n", deep);
      const GemmCase cd[] = {
          {"tiled L2 out+res", 8192, 512, 512, KD_EPI_RESIDUAL, 0, 256, 0},
          {"tiled L2 down+res", 8192, 512, 1536, KD_EPI_RESIDUAL, 0, 256, 0},
          {"tiled merge1", 8192, 512, 1024, KD_EPI_STORE, 0, 256, 0, KD_A_MERGE2x2, 16, 16},
          {"tiled ragged res", 300, 160, 64, KD_EPI_RESIDUAL, 0, 300, 0},
          {"tiled small K", 2000, 128, 64, KD_EPI_STORE, 0, 2000, 0},
      };
      for (const auto& c : cd) run_gemm_case(c);
    }
    kd_set_option("tiled_deep", 0);
    kd_set_option("wstat", 1);
  }
  if (want("prefetch")) {
    for (int pf : {0, 1, 2, 6}) {
      kd_set_option("wstat_prefetch", pf & 3);
      kd_set_option("wstat_waves", pf == 6 ? 4 : 0);
      printf("-- wstat_prefetch = %d\n", pf);
      const GemmCase cw[] = {
          {"prefetch L0 qkv", 131072, 384, 128, KD_EPI_QKV, 1, 4096, 2},
          {"prefetch L0 geglu", 131072, 384, 128, KD_EPI_GEGLU, 1, 4096, 0},
          {"prefetch L0 out+res", 131072, 128, 128, KD_EPI_RESIDUAL, 0, 4096, 0},
          {"prefetch L0 down+res", 131072, 128, 384, KD_EPI_RESIDUAL, 0, 4096, 0},
          {"prefetch ragged qkv", 4128, 384, 128, KD_EPI_QKV, 1, 96, 2},
      };
      for (const auto& c : cw) run_gemm_case(c);
    }
    kd_set_option("wstat_prefetch", 0);
    kd_set_option("wstat_waves", 0);
  }
  if (want("waves")) {
    for (int w : {4, 12}) {
      kd_set_option("wstat_waves", w);
      printf("-- wstat_waves = %d\n", w);
      const GemmCase cw[] = {
          {"waves L0 qkv", 131072, 384, 128, KD_EPI_QKV, 1, 4096, 2},
          {"waves L0 geglu", 131072, 384, 128, KD_EPI_GEGLU, 1, 4096, 0},
          {"waves L1 qkv", 32768, 768, 256, KD_EPI_QKV, 1, 1024, 4},
          {"waves L1 geglu", 32768, 768, 256, KD_EPI_GEGLU, 1, 1024, 0},
          {"waves L2 qkv", 8192, 1536, 512, KD_EPI_QKV, 1, 256, 8},
          {"waves L2 geglu", 8192, 1536, 512, KD_EPI_GEGLU, 1, 256, 0},
      };
      for (const auto& c : cw) run_gemm_case(c);
    }
    kd_set_option("wstat_waves", 0);
  }
  for (int v : {1, 3}) {
    kd_set_option("ffn_variant", v);
    if (want("ffn")) printf("-- ffn_variant = %d\n", v);
    run_ffn_case("ffn L0", 131072, 128, 384, 4096);
    run_ffn_case("ffn ragged", 4128 + 77, 128, 448, 96);
    run_ffn_case("ffn tiny", 37, 128, 64, 37);
    run_ffn_case("ffn L1 (K=256)", 32768, 256, 768, 1024);
    run_ffn_case("ffn K=256 ragged", 1000 + 77, 256, 192, 96);
    run_ffn_case("ffn K=256 one tile", 130, 256, 64, 130);
    run_ffn_case("ffn one tile", 300, 128, 64, 100);
    run_ffn_case("ffn two tiles", 300, 128, 128, 100);
  }
  kd_set_option("ffn_variant", 1);
  run_patch_case("patch flowers", 32, 3, 64, 64, 4, 128);
  run_patch_case("patch mnist", 4, 1, 7, 7, 4, 256);
  run_patch_case("patch cifar (generic)", 8, 3, 16, 16, 2, 256);
  run_patch_case("patch odd 5x9 c4", 3, 4, 5, 9, 4, 128);
  kd_set_option("patch_fast", 0);
  run_patch_case("generic patch flowers", 32, 3, 64, 64, 4, 128);
  kd_set_option("patch_fast", 1);
  if (want("generic")) {      // the generic kernel on every mode it serves (fast kernels switched off)
    kd_set_option("bf16_fast", 0);
    const GemmCase cg[] = {
        {"generic qkv", 8192, 768, 256, KD_EPI_QKV, 1, 1024, 4},
        {"generic qkv mnist", 196, 768, 256, KD_EPI_QKV, 1, 49, 4},
        {"generic geglu", 8192, 384, 128, KD_EPI_GEGLU, 1, 4096, 0},
        {"generic geglu mnist", 196, 768, 256, KD_EPI_GEGLU, 1, 49, 0},
        {"generic out+res", 8192, 128, 128, KD_EPI_RESIDUAL, 0, 4096, 0},
        {"generic down+res mnist", 196, 256, 768, KD_EPI_RESIDUAL, 0, 49, 0},
        {"generic store norm", 300, 96, 100, KD_EPI_STORE, 1, 50, 0},
        {"generic geglu N=400", 2880, 400, 192, KD_EPI_GEGLU, 1, 960, 0},
        {"generic down K=400", 2880, 192, 400, KD_EPI_RESIDUAL, 0, 960, 0},
        {"generic store N=72", 500, 72, 64, KD_EPI_STORE, 0, 500, 0},
        {"generic store K=52", 260, 64, 52, KD_EPI_STORE, 0, 260, 0},
        {"generic merge", 2048, 256, 512, KD_EPI_STORE, 0, 256, 0, KD_A_MERGE2x2, 16, 16},
        {"generic merge small", 72, 64, 48, KD_EPI_STORE, 0, 36, 0, KD_A_MERGE2x2, 6, 6},
        {"generic split", 2048, 512, 256, KD_EPI_SPLIT_LERP, 0, 256, 0, KD_A_PLAIN, 16, 16},
    };
    for (const auto& c : cg) run_gemm_case(c);
    kd_set_option("bf16_fast", 1);
  }
  const AttnCase acs[] = {
      {"attn global L2", 0, 32, 16, 16, 8, 0, 0},
      {"attn global 49", 0, 4, 7, 7, 4, 0, 0},
      {"attn global 64", 0, 64, 8, 8, 8, 0, 0},
      {"attn global 100", 0, 3, 10, 10, 2, 0, 0},
      {"attn global long 1024", 0, 4, 32, 32, 4, 0, 0},
      {"attn global long 900", 0, 2, 30, 30, 2, 0, 0},
      {"attn window L0 s0", 1, 32, 64, 64, 2, 8, 0},
      {"attn window L0 s4", 1, 32, 64, 64, 2, 8, 4},
      {"attn window L1 s4", 1, 32, 32, 32, 4, 8, 4},
      {"attn window4 s2", 1, 2, 16, 24, 2, 4, 2},
      {"attn window16 s8", 1, 2, 32, 48, 2, 16, 8},
      {"attn na L0 k7", 2, 32, 64, 64, 2, 7, 0},
      {"attn na L1 k7", 2, 32, 32, 32, 4, 7, 0},
      {"attn na small k7", 2, 2, 9, 11, 2, 7, 0},
      {"attn na odd k7", 2, 2, 21, 37, 2, 7, 0},
      {"attn na k3", 2, 2, 20, 33, 2, 3, 0},
      {"attn na k5", 2, 2, 20, 33, 2, 5, 0},
      {"attn na k9", 2, 2, 20, 33, 2, 9, 0},
      {"attn na k11", 2, 2, 24, 40, 2, 11, 0},
      {"attn na k13", 2, 2, 32, 29, 2, 13, 0},
      {"attn na k13 L1", 2, 32, 32, 32, 4, 13, 0},
  };
  for (const auto& c : acs) run_attn_case(c);
  if (want("attn global L2")) {
    for (int qw : {2, 4}) {
      kd_set_option("attn_global_qw", qw);
      printf("-- attn_global_qw = %d\n", qw);
      run_attn_case({"attn global L2", 0, 32, 16, 16, 8, 0, 0});
    }
    kd_set_option("attn_global_qw", 8);
  }
  printf("%s (%d failing case%s)\n", g_fail ? "HARNESS FAILED" : "HARNESS OK", g_fail, g_fail == 1 ? "" : "s");
  return g_fail ? 1 : 0;
}
