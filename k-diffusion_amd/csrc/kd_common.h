// Shared host/device helpers for libkdiff_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/kdiff_hip.h"

namespace kd {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int WAVE = 64;

// ---- error reporting (thread-local message, C ABI returns a code) -------------------------------
char* err_buf();
int fail(int code, const char* fmt, ...);

// ---- per-launch profiling (bench.py): hipEvent pairs around launches when enabled ---------------
struct ProfRec { std::string name; hipEvent_t e0, e1; double flops, bytes; };
bool prof_on();
void prof_begin(const char* name, double flops, double bytes, hipStream_t s);
void prof_end(hipStream_t s);

struct LaunchScope {
  hipStream_t s; bool on;
  LaunchScope(const char* name, double flops, double bytes, hipStream_t st) : s(st), on(prof_on()) {
    if (on) prof_begin(name, flops, bytes, s);
  }
  ~LaunchScope() { if (on) prof_end(s); }
};

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(KD_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return KD_OK;
}

// ---- device helpers ---------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum_xor(float v, int width) {
#pragma unroll
  for (int o = width >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max_xor(float v, int width) {
#pragma unroll
  for (int o = width >> 1; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// exact (erf) GELU, F.gelu default (image_transformer_v2.py:95)
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// 32x32 MFMA C/D fragment: element `reg` of lane `lane` is C[row][col]
__device__ __forceinline__ int mfma32_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

}  // namespace kd
