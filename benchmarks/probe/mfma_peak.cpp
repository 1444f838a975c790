// Calibration: what does a pure v_mfma_f32_32x32x16_bf16 loop sustain on this box, and at which shader clock?
// (no memory traffic; 4 independent accumulators per wave; 1, 2 and 4 waves per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
__global__ __launch_bounds__(256) void mfma_loop(float* out, unsigned long long* clk, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(threadIdx.x * 3 + i); }
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
int main() {
  float* out; unsigned long long* clk;
  hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&clk, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wg_per_cu = 1; wg_per_cu <= 4; wg_per_cu *= 2) {
    const int iters = 20000, blocks = 256 * wg_per_cu;
    hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, out, clk, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double flops = (double)blocks * 4 /*waves*/ * iters * 16.0 * 32 * 32 * 16 * 2;
    printf("%d wave(s)/SIMD: %.3f ms  %.0f TF/s  | wave 0: %llu shader cycles over %llu x 10 ns -> %.2f GHz, %.1f cycles / MFMA / wave\n", wg_per_cu, ms,
           flops / ms / 1e9, h[0], h[1], (double)h[0] / (h[1] * 10.0), (double)h[0] / (iters * 16.0));
  }
  return 0;
}
