// Do the pipes of a gfx950 CU overlap, and what does the chip's power / clock management do while they run?
//
// Synthetic kernels with NO memory traffic beyond what is named, each run back to back for ~2 s while a host thread samples the
// socket power (hwmon power1_average) and the shader clock (hwmon freq1_input); every kernel also measures its own shader clock
// as s_memtime ticks per 100 MHz s_memrealtime tick.
//
//   mfma        4 independent accumulator chains of v_mfma_f32_32x32x16_bf16 per wave, 8 waves per CU
//   valu        the GEGLU-style vector mix (packed fma chains + v_exp_f32), 8 waves per CU
//   mfma+valu   the same instruction counts of both in ONE wave stream, interleaved 1 MFMA : VPM vector instructions
//   stream      HBM read + write of a 1 GiB buffer (grid-stride float4 copy)
//   mfma | stream   the two kernels concurrently on two streams
//
// If the pipes overlap and nothing else limits, t(mfma+valu) = max(t(mfma), t(valu)); if time is set by the work done (power limited),
// t(mfma+valu) -> t(mfma) + t(valu) with the clock falling under the combined load.
//
// build: hipcc --offload-arch=gfx950 -O2 -std=c++17 power_probe.cpp -o power_probe      (no library needed)
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <glob.h>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

// AGPR: the accumulators live in the AccVGPR half of the register file (inline asm, "a" constraint) instead of the ArchVGPRs
template <int DO_MFMA, int DO_VALU, int VPM, int AGPR = 0>
__global__ __launch_bounds__(1024) void work_kernel(int iters, float* sink, unsigned long long* clk) {
  const int lane = threadIdx.x & 63;
  unsigned long long t0 = 0, r0 = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (lane - i)); }
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  f32x2 v[4];
  for (int j = 0; j < 4; ++j) v[j] = f32x2{0.1f * lane + j, 0.2f * lane - j};
  const f32x2 c0 = {0.999f, 1.001f}, c1 = {1e-3f, -1e-3f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if (DO_MFMA) {
        if (AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[s & 3]) : "v"(a), "v"(b));
        else acc[s & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[s & 3], 0, 0, 0);
      }
      if (DO_VALU) {
#pragma unroll
        for (int u = 0; u < VPM; ++u) {
          if ((u & 7) == 7) { v[u & 3].x = __builtin_amdgcn_exp2f(v[u & 3].x * 1e-3f); }
          else v[u & 3] = __builtin_elementwise_fma(v[u & 3], c0, c1);
        }
      }
      if (DO_MFMA && DO_VALU && AGPR) __builtin_amdgcn_sched_barrier(0);
      if (DO_MFMA && DO_VALU && !AGPR) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x402, VPM + 2, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j) { s += acc[j][0] + acc[j][7]; s += v[j].x + v[j].y; }
  if (s == 12345.678f) sink[threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clk[0] = __builtin_amdgcn_s_memtime() - t0;
    clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}

// Which vector instruction classes serialise with the matrix pipe?  KIND: 0 v_fma_f32, 1 v_exp_f32 (transcendental), 2 v_pk_fma_f32,
// 3 integer (v_add_u32 / v_xor), 4 v_cvt_pk_bf16_f32.  8 instructions of the class per MFMA slot, independent chains.
template <int DO_MFMA, int KIND>
__global__ __launch_bounds__(512) void kind_kernel(int iters, float* sink, unsigned long long* clk) {
  const int lane = threadIdx.x & 63;
  unsigned long long t0 = 0, r0 = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (lane - i)); }
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float f[8];
  f32x2 pf[8];
  unsigned u[8];
  for (int j = 0; j < 8; ++j) { f[j] = 0.01f * lane + j; pf[j] = f32x2{f[j], -f[j]}; u[j] = lane * 77u + j; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if (DO_MFMA) acc[s & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[s & 3], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (KIND == 0) f[j] = fmaf(f[j], 0.999f, 1e-3f);
        else if (KIND == 1) f[j] = __builtin_amdgcn_exp2f(f[j]);
        else if (KIND == 2) pf[j] = __builtin_elementwise_fma(pf[j], f32x2{0.999f, 1.001f}, f32x2{1e-3f, -1e-3f});
        else if (KIND == 3) u[j] = (u[j] + 0x9E3779B9u) ^ (u[j] >> 7);
        else {
          typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
          const bf2 v = {(__bf16)f[j], (__bf16)__uint_as_float(u[j] | 0x3f000000u)};
          u[j] = __builtin_bit_cast(unsigned, v);
        }
      }
      if (DO_MFMA) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x402, 20, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float sum = 0.f;
  for (int j = 0; j < 4; ++j) sum += acc[j][0] + acc[j][7];
  for (int j = 0; j < 8; ++j) sum += f[j] + pf[j].x + pf[j].y + (float)u[j];
  if (sum == 12345.678f) sink[threadIdx.x] = sum;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clk[0] = __builtin_amdgcn_s_memtime() - t0;
    clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}

// MFMA whose A operand comes from LDS (one conflict-free ds_read_b128 per MFMA, requested 4 slots ahead), like the W fragments of
// the GEMM kernels; DO_MFMA = 0 leaves only the reads (their values are consumed by a cheap xor so they are not dead).
template <int DO_MFMA, int READS_PER_MFMA>
__global__ __launch_bounds__(512) void lds_kernel(int iters, float* sink, unsigned long long* clk) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 0.001f * i;
  __syncthreads();
  unsigned long long t0 = 0, r0 = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  bf16x8 b;
  for (int i = 0; i < 8; ++i) b[i] = (__bf16)(0.002f * (lane - i));
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
  u32x4 keep = {0, 0, 0, 0};
  const char* base = lds + lane * 16;
  u32x4 frag[8];
  for (int i = 0; i < 8; ++i) frag[i] = *reinterpret_cast<const u32x4*>(base + i * 1024);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if (DO_MFMA) acc[s & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, frag[s & 7]), b, acc[s & 3], 0, 0, 0);
      else keep ^= frag[s & 7];
#pragma unroll
      for (int u = 0; u < READS_PER_MFMA; ++u) frag[(s + 4 + u) & 7] = *reinterpret_cast<const u32x4*>(base + ((s * READS_PER_MFMA + u + it) & 31) * 1024);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float sum = 0.f;
  for (int j = 0; j < 4; ++j) sum += acc[j][0] + acc[j][7];
  if (sum == 12345.678f || keep[0] == 0x12345678u) sink[threadIdx.x] = sum + keep[1];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clk[0] = __builtin_amdgcn_s_memtime() - t0;
    clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}

__global__ __launch_bounds__(256) void stream_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float4 x = src[i];
    x.x += 1.f;
    dst[i] = x;
  }
}

// ---- hwmon sampling -----------------------------------------------------------------------------------------------------------
static std::string g_pci;      // "0000:xx:00.0" of the HIP device (a box may show the hwmon nodes of GPUs that are not ours)
static std::string find_hwmon(const char* leaf) {
  glob_t g;
  std::string pat = (g_pci.empty() ? std::string("/sys/class/drm/card*/device") : "/sys/bus/pci/devices/" + g_pci) + "/hwmon/hwmon*/" + leaf;
  std::string out;
  if (glob(pat.c_str(), 0, nullptr, &g) == 0 && g.gl_pathc > 0) out = g.gl_pathv[0];
  globfree(&g);
  return out;
}
static double read_num(const std::string& path) {
  if (path.empty()) return -1;
  FILE* f = fopen(path.c_str(), "r");
  if (!f) return -1;
  double v = -1;
  if (fscanf(f, "%lf", &v) != 1) v = -1;
  fclose(f);
  return v;
}

struct Sampler {
  std::string p_power = find_hwmon("power1_average"), p_input = find_hwmon("power1_input"), p_freq = find_hwmon("freq1_input"), p_cap = find_hwmon("power1_cap");
  std::atomic<bool> run{false};
  std::thread th;
  double sum_w = 0, sum_f = 0, max_w = 0;
  int n = 0;
  void start() {
    sum_w = sum_f = max_w = 0; n = 0; run = true;
    th = std::thread([this] {
      while (run) {
        double w = read_num(p_power);
        if (w < 0) w = read_num(p_input);
        const double f = read_num(p_freq);
        if (w >= 0) { sum_w += w * 1e-6; if (w * 1e-6 > max_w) max_w = w * 1e-6; }
        if (f >= 0) sum_f += f * 1e-9;
        ++n;
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
      }
    });
  }
  void stop() { run = false; th.join(); }
};

static const char* g_only = nullptr;                  // argv[1]: run only the measurements whose name contains it
template <class F>
static void measure(const char* name, Sampler& smp, F&& launch, double work_unit, const char* unit, unsigned long long* clk_dev) {
  if (g_only && !strstr(name, g_only)) return;
  for (int i = 0; i < 3; ++i) launch();
  CHK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  // calibrate the repeat count to ~2 s
  CHK(hipEventRecord(e0, 0));
  for (int i = 0; i < 5; ++i) launch();
  CHK(hipEventRecord(e1, 0));
  CHK(hipEventSynchronize(e1));
  float ms = 0;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  const int reps = std::max(10, (int)(2000.0 / (ms / 5)));
  smp.start();
  std::this_thread::sleep_for(std::chrono::milliseconds(100));
  CHK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) launch();
  CHK(hipEventRecord(e1, 0));
  CHK(hipEventSynchronize(e1));
  smp.stop();
  CHK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps;
  unsigned long long ck[2] = {0, 0};
  if (clk_dev) CHK(hipMemcpy(ck, clk_dev, 16, hipMemcpyDeviceToHost));
  printf("%-22s %9.1f us/launch  %8.1f %s   power avg %6.0f W (max %6.0f)  hwmon sclk %.2f GHz  in-kernel clock %.2f GHz   (%d samples)\n", name, us,
         work_unit / us, unit, smp.n ? smp.sum_w / smp.n : -1.0, smp.max_w, smp.n ? smp.sum_f / smp.n : -1.0,
         ck[1] ? (double)ck[0] / (double)ck[1] * 0.1 : 0.0, smp.n);
}

int main(int argc, char** argv) {
  if (argc > 1) g_only = argv[1];
  int dev = 0, cus = 0;
  CHK(hipGetDevice(&dev));
  CHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  {
    char id[64] = "";
    if (hipDeviceGetPCIBusId(id, sizeof(id), dev) == hipSuccess) { g_pci = id; for (auto& c : g_pci) c = (char)tolower(c); }
  }
  Sampler smp;
  printf("pci %s  ", g_pci.c_str());
  printf("CUs %d; hwmon power %s, freq %s, cap %.0f W\n", cus, smp.p_power.empty() ? smp.p_input.c_str() : smp.p_power.c_str(), smp.p_freq.c_str(),
         read_num(smp.p_cap) * 1e-6);
  float* sink;
  unsigned long long* clk;
  CHK(hipMalloc(&sink, 4096));
  CHK(hipMalloc(&clk, 64));
  CHK(hipMemset(clk, 0, 64));
  const int iters = 4000;
  const double mfma_flops = (double)cus * 8 * iters * 16 * 2.0 * 32 * 32 * 16;     // per launch
  const double valu_ops = (double)cus * 8 * 64.0 * iters * 16;                       // lane-instructions / VPM
  std::this_thread::sleep_for(std::chrono::milliseconds(300));
  {
    smp.start();
    std::this_thread::sleep_for(std::chrono::milliseconds(500));
    smp.stop();
    printf("%-22s power avg %6.0f W  hwmon sclk %.2f GHz\n", "idle", smp.n ? smp.sum_w / smp.n : -1.0, smp.n ? smp.sum_f / smp.n : -1.0);
  }
  measure("mfma", smp, [&] { hipLaunchKernelGGL((work_kernel<1, 0, 8>), dim3(cus), dim3(512), 0, 0, iters, sink, clk); }, mfma_flops * 1e-6, "TFLOP/s", clk);
  measure("valu (8 per slot)", smp, [&] { hipLaunchKernelGGL((work_kernel<0, 1, 8>), dim3(cus), dim3(512), 0, 0, iters, sink, clk); }, valu_ops * 8 * 1e-6, "T lane-op/s", clk);
  measure("mfma+valu (1:8)", smp, [&] { hipLaunchKernelGGL((work_kernel<1, 1, 8>), dim3(cus), dim3(512), 0, 0, iters, sink, clk); }, mfma_flops * 1e-6, "TFLOP/s", clk);
  measure("mfma, AGPR acc", smp, [&] { hipLaunchKernelGGL((work_kernel<1, 0, 8, 1>), dim3(cus), dim3(512), 0, 0, iters, sink, clk); }, mfma_flops * 1e-6, "TFLOP/s", clk);
  measure("mfma+valu (1:8), AGPR", smp, [&] { hipLaunchKernelGGL((work_kernel<1, 1, 8, 1>), dim3(cus), dim3(512), 0, 0, iters, sink, clk); }, mfma_flops * 1e-6, "TFLOP/s", clk);
  measure("mfma+valu (1:4), AGPR", smp, [&] { hipLaunchKernelGGL((work_kernel<1, 1, 4, 1>), dim3(cus), dim3(512), 0, 0, iters, sink, clk); }, mfma_flops * 1e-6, "TFLOP/s", clk);
  measure("mfma+valu (1:4), 4 waves", smp, [&] { hipLaunchKernelGGL((work_kernel<1, 1, 4>), dim3(cus), dim3(256), 0, 0, iters, sink, clk); }, mfma_flops * 0.5e-6, "TFLOP/s", clk);
  measure("mfma+valu (1:4), 16 waves", smp, [&] { hipLaunchKernelGGL((work_kernel<1, 1, 4>), dim3(cus), dim3(1024), 0, 0, iters, sink, clk); }, mfma_flops * 2e-6, "TFLOP/s", clk);
  // MFMA fed from LDS, on 16 CUs (full clock): do the fragment reads hide behind the MFMAs?
  CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(lds_kernel<1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(lds_kernel<0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(lds_kernel<1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(lds_kernel<0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  const double lds_bytes16 = 16.0 * 8 * 64 * 16.0 * iters * 16;      // 16 CUs x 8 waves x 1 KiB per read
  measure("lds reads only, 16 CUs", smp, [&] { hipLaunchKernelGGL((lds_kernel<0, 1>), dim3(16), dim3(512), 65536, 0, iters, sink, clk); }, lds_bytes16 * 1e-3, "GB/s", clk);
  measure("mfma + 1 read, 16 CUs", smp, [&] { hipLaunchKernelGGL((lds_kernel<1, 1>), dim3(16), dim3(512), 65536, 0, iters, sink, clk); }, mfma_flops / cus * 16e-6, "TFLOP/s", clk);
  measure("2 reads only, 16 CUs", smp, [&] { hipLaunchKernelGGL((lds_kernel<0, 2>), dim3(16), dim3(512), 65536, 0, iters, sink, clk); }, 2 * lds_bytes16 * 1e-3, "GB/s", clk);
  measure("mfma + 2 reads, 16 CUs", smp, [&] { hipLaunchKernelGGL((lds_kernel<1, 2>), dim3(16), dim3(512), 65536, 0, iters, sink, clk); }, mfma_flops / cus * 16e-6, "TFLOP/s", clk);
  // instruction classes beside the MFMA, 16 CUs (full clock): alone / with one MFMA per 8 of them
#define KIND_PAIR(K, label) \
  measure(label " alone, 16 CUs", smp, [&] { hipLaunchKernelGGL((kind_kernel<0, K>), dim3(16), dim3(512), 0, 0, iters, sink, clk); }, 0.0, "-", clk); \
  measure("mfma + " label ", 16 CUs", smp, [&] { hipLaunchKernelGGL((kind_kernel<1, K>), dim3(16), dim3(512), 0, 0, iters, sink, clk); }, mfma_flops / cus * 16e-6, "TFLOP/s", clk);
  KIND_PAIR(0, "v_fma_f32 x8")
  KIND_PAIR(1, "v_exp_f32 x8")
  KIND_PAIR(2, "v_pk_fma_f32 x8")
  KIND_PAIR(3, "int add/xor x8")
  KIND_PAIR(4, "v_cvt_pk_bf16 x8")
#undef KIND_PAIR
  // a quarter of the CUs: far below any power limit -- does the sum rule survive?
  const int q = cus / 4;
  measure("mfma, 64 CUs", smp, [&] { hipLaunchKernelGGL((work_kernel<1, 0, 8>), dim3(q), dim3(512), 0, 0, iters, sink, clk); }, mfma_flops * 0.25e-6, "TFLOP/s", clk);
  measure("valu (4), 64 CUs", smp, [&] { hipLaunchKernelGGL((work_kernel<0, 1, 4>), dim3(q), dim3(512), 0, 0, iters, sink, clk); }, valu_ops * 4 * 0.25e-6, "T lane-op/s", clk);
  measure("mfma+valu (1:4), 64 CUs", smp, [&] { hipLaunchKernelGGL((work_kernel<1, 1, 4>), dim3(q), dim3(512), 0, 0, iters, sink, clk); }, mfma_flops * 0.25e-6, "TFLOP/s", clk);
  measure("mfma, 16 CUs", smp, [&] { hipLaunchKernelGGL((work_kernel<1, 0, 8>), dim3(16), dim3(512), 0, 0, iters, sink, clk); }, mfma_flops / cus * 16e-6, "TFLOP/s", clk);
  measure("valu (4), 16 CUs", smp, [&] { hipLaunchKernelGGL((work_kernel<0, 1, 4>), dim3(16), dim3(512), 0, 0, iters, sink, clk); }, valu_ops * 4 / cus * 16e-6, "T lane-op/s", clk);
  measure("mfma+valu (1:4), 16 CUs", smp, [&] { hipLaunchKernelGGL((work_kernel<1, 1, 4>), dim3(16), dim3(512), 0, 0, iters, sink, clk); }, mfma_flops / cus * 16e-6, "TFLOP/s", clk);
  measure("valu (4 per slot)", smp, [&] { hipLaunchKernelGGL((work_kernel<0, 1, 4>), dim3(cus), dim3(512), 0, 0, iters, sink, clk); }, valu_ops * 4 * 1e-6, "T lane-op/s", clk);
  measure("mfma+valu (1:4)", smp, [&] { hipLaunchKernelGGL((work_kernel<1, 1, 4>), dim3(cus), dim3(512), 0, 0, iters, sink, clk); }, mfma_flops * 1e-6, "TFLOP/s", clk);
  // HBM stream
  const size_t bytes = (size_t)1 << 30;
  float4 *src, *dst;
  CHK(hipMalloc(&src, bytes));
  CHK(hipMalloc(&dst, bytes));
  CHK(hipMemset(src, 0, bytes));
  measure("stream (r+w 2 GiB)", smp, [&] { hipLaunchKernelGGL(stream_kernel, dim3(cus * 16), dim3(256), 0, 0, src, dst, bytes / 16); }, 2.0 * bytes * 1e-3, "GB/s", nullptr);
  hipStream_t s2;
  CHK(hipStreamCreate(&s2));
  // concurrently: the MFMA kernel on half the waves per CU (256 threads) + the stream kernel on a second stream
  measure("mfma(4 waves) alone", smp, [&] { hipLaunchKernelGGL((work_kernel<1, 0, 8>), dim3(cus), dim3(256), 0, 0, iters, sink, clk); }, mfma_flops * 0.5e-6, "TFLOP/s", clk);
  measure("mfma(4 waves) | stream", smp,
          [&] {
            hipLaunchKernelGGL((work_kernel<1, 0, 8>), dim3(cus), dim3(256), 0, 0, iters, sink, clk);
            hipLaunchKernelGGL(stream_kernel, dim3(cus * 16), dim3(256), 0, s2, src, dst, bytes / 16);
            hipStreamSynchronize(s2);
          },
          mfma_flops * 0.5e-6, "TFLOP/s", clk);
  return 0;
}
