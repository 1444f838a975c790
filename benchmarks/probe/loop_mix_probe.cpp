// Issue-level model of ONE K-loop stage (32 k of a 128-row x 128-column tile per wave: 4 output blocks of 32 x 32) of a projection that converts
// its fp32 activation rows inside the loop (csrc/gemm_x3r.hip), whole chip, operands from LDS, no global traffic, no barriers:
//   A  split3 as built: 20 ds_read_b128, hi / lo bf16 split of the lane's 16 floats, 24 x v_mfma_f32_32x32x16_bf16
//   B  fp16 hi + e4m3 corrections: the same 20 reads, fp16 hi (packed convert), lo = x - hi, block maxima of hi and lo (+ half-wave exchange),
//      two scale bytes, e4m3 conversion of both blocks, 8 x v_mfma_f32_32x32x16_f16 + 4 x v_mfma_scale_f32_32x32x64_f8f6f4
// Question: does B's extra vector work fit in the shadow of its shorter matrix time, i.e. what is B / A per stage?
//   build: hipcc --offload-arch=gfx950 -O2 loop_mix_probe.cpp -o loop_mix_probe ; run: ./loop_mix_probe [waves per workgroup = 4]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  b2 r = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ float bf_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
__device__ __forceinline__ int scale_byte(float amax) {       // smallest power of two >= amax / 448, as an E8M0 byte
  const int b = (int)((__builtin_bit_cast(unsigned, amax * (1.0f / 448.0f)) + 0x7fffffu) >> 23);
  return b < 12 ? 12 : (b > 253 ? 253 : b);
}

template <int MODE>
__global__ __launch_bounds__(512) void loop(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, lh = lane >> 5;
  for (int i = tid; i < 64 * 1024 / 4; i += blockDim.x)
    reinterpret_cast<float*>(smem)[i] = 0.25f + 0.001f * (float)((i * 2654435761u) >> 22);      // positive, varied, finite in every format
  __syncthreads();
  v16f acc[4];
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const char* base = smem + lane * 16 + (wid & 3) * 1024;
  for (int it = 0; it < iters; ++it) {
    const char* st = base + (it & 1) * 32768;                       // (another slot every stage: nothing is loop-invariant)
    // ---- the lane's 16 floats of the stage (2 chunks x 2 reads) and the 16 W fragment reads --------------------------------------------
    f4 x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = *reinterpret_cast<const f4*>(st + i * 4096);
    u4 w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = *reinterpret_cast<const u4*>(st + 16384 + (i & 7) * 2048 + (i >> 3) * 512);
    if constexpr (MODE == 0) {
      u4 ah[2], al[2];
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float a = x[2 * c + (q >> 1)][2 * (q & 1)], b = x[2 * c + (q >> 1)][2 * (q & 1) + 1];
          const unsigned h = pk_bf16(a, b);
          ah[c][q] = h;
          al[c][q] = pk_bf16(a - bf_lo(h), b - bf_hi(h));
        }
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, w[8 * c + 2 * j + (t == 0)]), __builtin_bit_cast(bf8, t == 1 ? al[c] : ah[c]), acc[j], 0, 0, 0);
    } else {
      // fp16 hi (clamped), lo = x - hi, block maxima, scale bytes, e4m3 blocks
      u4 ah[2];
      float lo[16], hi[16];
      float mh = 0.f, ml = 0.f;
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float a = __builtin_amdgcn_fmed3f(x[2 * c + (q >> 1)][2 * (q & 1)], -65504.f, 65504.f);
          const float b = __builtin_amdgcn_fmed3f(x[2 * c + (q >> 1)][2 * (q & 1) + 1], -65504.f, 65504.f);
          const h2 h = __builtin_convertvector(f2{a, b}, h2);
          ah[c][q] = __builtin_bit_cast(unsigned, h);
          const int e = 8 * c + 2 * q;
          hi[e] = a; hi[e + 1] = b;
          lo[e] = x[2 * c + (q >> 1)][2 * (q & 1)] - (float)h[0];
          lo[e + 1] = x[2 * c + (q >> 1)][2 * (q & 1) + 1] - (float)h[1];
          mh = __builtin_fmaxf(mh, __builtin_fmaxf(__builtin_fabsf(a), __builtin_fabsf(b)));
          ml = __builtin_fmaxf(ml, __builtin_fmaxf(__builtin_fabsf(lo[e]), __builtin_fabsf(lo[e + 1])));
        }
      mh = __builtin_fmaxf(mh, __shfl_xor(mh, 32, 64));
      ml = __builtin_fmaxf(ml, __shfl_xor(ml, 32, 64));
      const int bh = scale_byte(mh), bl = scale_byte(ml);
      const float ih = __builtin_bit_cast(float, (unsigned)(254 - bh) << 23), il = __builtin_bit_cast(float, (unsigned)(254 - bl) << 23);
      v8i x8;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int lo8 = 0, hi8 = 0;
        lo8 = __builtin_amdgcn_cvt_pk_fp8_f32(lo[4 * q] * il, lo[4 * q + 1] * il, lo8, false);
        lo8 = __builtin_amdgcn_cvt_pk_fp8_f32(lo[4 * q + 2] * il, lo[4 * q + 3] * il, lo8, true);
        hi8 = __builtin_amdgcn_cvt_pk_fp8_f32(hi[4 * q] * ih, hi[4 * q + 1] * ih, hi8, false);
        hi8 = __builtin_amdgcn_cvt_pk_fp8_f32(hi[4 * q + 2] * ih, hi[4 * q + 3] * ih, hi8, true);
        x8[q] = lo8;
        x8[4 + q] = hi8;
      }
      const int sx = lh ? bh : bl, sw = 0x7f7f7f7f;
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, w[4 * c + j]), __builtin_bit_cast(h8, ah[c]), acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v8i w8;
#pragma unroll
        for (int v = 0; v < 4; ++v) { w8[v] = (int)(w[8 + 2 * j][v] & 0x77777777u); w8[4 + v] = (int)(w[9 + 2 * j][v] & 0x77777777u); }
        acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w8, x8, acc[j], 0, 0, 0, sw, 0, sx);
      }
    }
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * blockDim.x + tid] = s;
}

int main(int argc, char** argv) {
  const int waves = argc > 1 ? atoi(argv[1]) : 4;
  hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount, wgs = cus * (waves <= 4 ? 2 : 1);           // 8 waves per CU = 2 per SIMD either way
  float* out; (void)hipMalloc(&out, sizeof(float) * wgs * waves * 64);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(loop<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(loop<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  printf("%d CUs, %d workgroups of %d waves (2 waves per SIMD), 64 KiB of LDS each\n", cus, wgs, waves);
  double per_stage[2] = {0, 0};
  for (int round = 0; round < 3; ++round)
    for (int mode = 0; mode < 2; ++mode) {
      const int iters = 200000;
      auto go = [&]() {
        if (mode == 0) hipLaunchKernelGGL(loop<0>, dim3(wgs), dim3(waves * 64), 65536, 0, out, iters);
        else hipLaunchKernelGGL(loop<1>, dim3(wgs), dim3(waves * 64), 65536, 0, out, iters);
      };
      go(); (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0); go(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      const double ns = ms * 1e6 / iters;
      per_stage[mode] = ns;
      const double tf = (double)iters * wgs * waves * 4 * 2.0 * 32 * 32 * 32 / (ms * 1e-3) / 1e12;
      printf("round %d  %-28s %8.1f ms  %7.1f ns per stage per wave  %7.1f TFLOP/s of products\n", round, mode == 0 ? "A split3 (24 bf16)" : "B fp16 + e4m3 (8 + 4)", ms, ns, tf);
    }
  printf("B / A per stage: %.3f\n", per_stage[1] / per_stage[0]);
  return 0;
}
