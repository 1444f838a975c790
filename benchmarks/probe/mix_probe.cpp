// Matrix time and power of three ways to carry one fp32-grade product block on gfx950, register-only loops on the whole chip (no memory traffic):
//   A  "split3"        per 32 k of a 32 x 32 block: 6 x v_mfma_f32_32x32x16_bf16          (hi*hi + hi*lo + lo*hi, what the fp32-parity mode executes)
//   B  "f16 + fp8"     per 32 k: 2 x v_mfma_f32_32x32x16_f16 + 1 x v_mfma_scale_f32_32x32x64_f8f6f4 (both correction terms in one doubled-k e4m3 instruction)
//   C  "bf16"          per 32 k: 2 x v_mfma_f32_32x32x16_bf16                                 (the bf16 mode's one term)
// Question: under the board's power limit, is B really 2/3 of A's time per product?
//   build: hipcc --offload-arch=gfx950 -O2 mix_probe.cpp -o mix_probe ; run: ./mix_probe [milliseconds per measurement]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>
__global__ __launch_bounds__(256) void loop(float* out, int iters) {
  const unsigned l = blockIdx.x * 256 + threadIdx.x;
  // operands with random signs / mantissas and exponents near 1 (zeros would not draw the power real data draws)
  u4 ra[2], rb[2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 4; ++j) {
      ra[i][j] = (mix(l * 16 + i * 4 + j) & 0x807f807fu) | 0x3f003f00u;                 // bf16 pairs in +-[0.5, 1)
      rb[i][j] = (mix(l * 16 + 8 + i * 4 + j) & 0x807f807fu) | 0x3f003f00u;
    }
  u4 ha[2], hb[2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 4; ++j) {
      ha[i][j] = (mix(l * 32 + i * 4 + j) & 0x83ff83ffu) | 0x38003800u;                 // f16 pairs in +-[0.5, 1)
      hb[i][j] = (mix(l * 32 + 8 + i * 4 + j) & 0x83ff83ffu) | 0x38003800u;
    }
  v8i qa, qb;
  for (int j = 0; j < 8; ++j) {
    qa[j] = (int)((mix(l * 64 + j) & 0x87878787u) | 0x30303030u);                        // e4m3 bytes in +-[0.5, 1)
    qb[j] = (int)((mix(l * 64 + 8 + j) & 0x87878787u) | 0x30303030u);
  }
  v16f acc[4];
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const int sc = 0x7f7f7f7f - 0x0b0b0b0b * 0;      // unit scales
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (MODE == 0) {
#pragma unroll
        for (int t = 0; t < 6; ++t)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, ra[t & 1]), __builtin_bit_cast(bf8, rb[(t >> 1) & 1]), acc[j], 0, 0, 0);
      } else if constexpr (MODE == 1) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, ha[t]), __builtin_bit_cast(h8, hb[t]), acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(qa, qb, acc[j], 0, 0, 0, sc, 0, sc);
      } else {
#pragma unroll
        for (int t = 0; t < 2; ++t)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, ra[t]), __builtin_bit_cast(bf8, rb[t]), acc[j], 0, 0, 0);
      }
    }
    // keep the operands alive and changing a little (no loop-invariant hoisting of the whole body, no value-dependent shortcuts)
    asm volatile("" : "+v"(ra[0]), "+v"(rb[0]), "+v"(ha[0]), "+v"(hb[0]), "+v"(qa), "+v"(qb));
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[l] = s;
}

int main(int argc, char** argv) {
  const double want_ms = argc > 1 ? atof(argv[1]) : 200.0;
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount, wgs = cus * 2;             // 8 waves per CU = 2 per SIMD
  float* out; hipMalloc(&out, sizeof(float) * wgs * 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[3] = {"A split3: 6 x bf16 32x32x16 per 32 k", "B f16 + fp8: 2 x f16 32x32x16 + 1 x scaled e4m3 32x32x64 per 32 k", "C bf16: 2 x bf16 32x32x16 per 32 k"};
  auto launch = [&](int mode, int iters) {
    if (mode == 0) hipLaunchKernelGGL(loop<0>, dim3(wgs), dim3(256), 0, 0, out, iters);
    else if (mode == 1) hipLaunchKernelGGL(loop<1>, dim3(wgs), dim3(256), 0, 0, out, iters);
    else hipLaunchKernelGGL(loop<2>, dim3(wgs), dim3(256), 0, 0, out, iters);
  };
  printf("%d CUs, %d workgroups of 4 waves (2 waves per SIMD), ~%.0f ms per measurement, 3 rounds\n", cus, wgs, want_ms);
  for (int round = 0; round < 3; ++round)
    for (int mode = 0; mode < 3; ++mode) {
      int iters = 20000;
      launch(mode, iters); hipDeviceSynchronize();
      hipEventRecord(e0); launch(mode, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      iters = (int)(iters * want_ms / ms);
      hipEventRecord(e0); launch(mode, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      // one "unit" = 32 k of one 32 x 32 block = 2 * 32 * 32 * 32 fp32-grade flops
      const double units = (double)iters * 4 * wgs * 4;
      const double tf = units * 2.0 * 32 * 32 * 32 / (ms * 1e-3) / 1e12;
      const double instr_tf = tf * (mode == 0 ? 3.0 : mode == 1 ? 3.0 : 1.0);        // executed: A 3 bf16 terms; B 1 f16 term + 2 e4m3 terms
      printf("round %d  %-70s %8.1f ms  %7.1f TFLOP/s of products (%7.1f executed)  %.3f ns per unit per wave\n", round, names[mode], ms, tf, instr_tf,
             ms * 1e6 / ((double)iters * 4));
    }
  return 0;
}
