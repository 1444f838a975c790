// How many vector instructions does a gfx950 SIMD issue in the shadow of a v_mfma_f32_32x32x16_bf16?
//
// Round 2's power_probe.cpp left this to the compiler (sched_group_barrier): the emitted loop was runs of 5 MFMA -> 24 VALU -> 11 MFMA
// with the scalar fmas SLP-packed into v_pk_fma_f32, and its "MFMA time and VALU time add" conclusion contradicts
// /opt/skills/guides/MI355X_MICROARCH.md (one wave per SIMD hides <= 5 single-issue instructions per 32-cycle MFMA gap: 32.4 cycles /
// MFMA with 5 fillers).  Here the whole loop body is ONE inline-asm statement, so the instruction stream is exactly what is written:
//
//     8 x { v_mfma_f32_32x32x16_bf16 acc[s % 4] ; N fillers on registers no MFMA touches }         (4 accumulators in rotation)
//
// filler kinds: 0 v_fma_f32   1 v_pk_fma_f32   2 v_exp_f32   3 v_cvt_pk_bf16_f32   4 ds_read_b128 (conflict-free)   5 s_nop 0 (a pure
// issue slot)   6 v_mul_f32 + v_fma_f32 alternating with a DEPENDENT chain (every filler reads the previous one's result)
// Reported: shader cycles per MFMA slot (s_memtime around the loop, wave 0 of workgroup 0 and the slowest wave of the launch) for
// 1 and 2 waves per SIMD, on 16 CUs (full clock) and on all 256 (power-managed), plus the wall time per launch.
//
// build: hipcc --offload-arch=gfx950 -O2 -std=c++17 issue_probe.cpp -o issue_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

// filler i of a slot works on register f[i % 8] (8 independent registers: a filler's result is needed again 8 fillers later at the
// earliest), except kind 6 where every filler continues ONE chain
#define F_FMA(i)   "v_fma_f32 %[f" #i "], %[f" #i "], %[c0], %[c1]\n\t"
#define F_PK(i)    "v_pk_fma_f32 %[p" #i "], %[p" #i "], %[q0], %[q1]\n\t"
#define F_EXP(i)   "v_exp_f32 %[f" #i "], %[f" #i "]\n\t"
#define F_CVT(i)   "v_cvt_pk_bf16_f32 %[f" #i "], %[f" #i "], %[c0]\n\t"
#define F_DS(i)    "ds_read_b128 %[d" #i "], %[la]\n\t"
#define F_NOP(i)   "s_nop 0\n\t"
#define F_DEP(i)   "v_fma_f32 %[f0], %[f0], %[c0], %[c1]\n\t"

#define REP0(F)
#define REP1(F) F(0)
#define REP2(F) F(0) F(1)
#define REP3(F) F(0) F(1) F(2)
#define REP4(F) F(0) F(1) F(2) F(3)
#define REP5(F) F(0) F(1) F(2) F(3) F(4)
#define REP6(F) F(0) F(1) F(2) F(3) F(4) F(5)
#define REP8(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7)
#define REP11(F) REP8(F) F(0) F(1) F(2)
#define REP16(F) REP8(F) REP8(F)
// kinds with 4 registers in rotation (an asm statement takes at most 30 operands)
#define R4_1(F) F(0)
#define R4_2(F) F(0) F(1)
#define R4_3(F) F(0) F(1) F(2)
#define R4_4(F) F(0) F(1) F(2) F(3)
#define R4_6(F) R4_4(F) F(0) F(1)
#define R4_8(F) R4_4(F) R4_4(F)

#define MFMA(a) "v_mfma_f32_32x32x16_bf16 %[acc" #a "], %[a], %[b], %[acc" #a "]\n\t"
#define BODY(FILL) MFMA(0) FILL MFMA(1) FILL MFMA(2) FILL MFMA(3) FILL MFMA(0) FILL MFMA(1) FILL MFMA(2) FILL MFMA(3) FILL

#define OPERANDS                                                                                                                     \
  [acc0] "+v"(acc[0]), [acc1] "+v"(acc[1]), [acc2] "+v"(acc[2]), [acc3] "+v"(acc[3]), [f0] "+v"(f[0]), [f1] "+v"(f[1]),              \
      [f2] "+v"(f[2]), [f3] "+v"(f[3]), [f4] "+v"(f[4]), [f5] "+v"(f[5]), [f6] "+v"(f[6]), [f7] "+v"(f[7]), [p0] "+v"(p[0]),         \
      [p1] "+v"(p[1]), [p2] "+v"(p[2]), [p3] "+v"(p[3]), [d0] "+v"(d[0]), [d1] "+v"(d[1]), [d2] "+v"(d[2]), [d3] "+v"(d[3])          \
      : [a] "v"(a), [b] "v"(b), [c0] "v"(c0), [c1] "v"(c1), [q0] "v"(q0), [q1] "v"(q1), [la] "v"(la)

template <int KIND, int N>
__device__ __forceinline__ void body(f32x16 (&acc)[4], float (&f)[8], f32x2 (&p)[4], f32x4 (&d)[4], bf16x8 a, bf16x8 b, float c0, float c1, f32x2 q0,
                                     f32x2 q1, unsigned la) {
#define TAIL ""
#define CASE(K, NN, FM, RP) \
  if constexpr (KIND == K && N == NN) { asm volatile(BODY(RP(FM)) TAIL : OPERANDS); }
  CASE(0, 0, F_FMA, REP0) CASE(0, 1, F_FMA, REP1) CASE(0, 2, F_FMA, REP2) CASE(0, 3, F_FMA, REP3) CASE(0, 4, F_FMA, REP4) CASE(0, 5, F_FMA, REP5)
  CASE(0, 6, F_FMA, REP6) CASE(0, 8, F_FMA, REP8) CASE(0, 11, F_FMA, REP11) CASE(0, 16, F_FMA, REP16)
  CASE(1, 1, F_PK, R4_1) CASE(1, 2, F_PK, R4_2) CASE(1, 3, F_PK, R4_3) CASE(1, 4, F_PK, R4_4) CASE(1, 6, F_PK, R4_6) CASE(1, 8, F_PK, R4_8)
  CASE(2, 2, F_EXP, REP2) CASE(2, 4, F_EXP, REP4) CASE(2, 8, F_EXP, REP8)
  CASE(3, 2, F_CVT, REP2) CASE(3, 4, F_CVT, REP4) CASE(3, 8, F_CVT, REP8)
#undef TAIL
#define TAIL "s_waitcnt lgkmcnt(0)"
  CASE(4, 1, F_DS, R4_1) CASE(4, 2, F_DS, R4_2) CASE(4, 4, F_DS, R4_4)
#undef TAIL
#define TAIL ""
  CASE(5, 2, F_NOP, REP2) CASE(5, 5, F_NOP, REP5) CASE(5, 8, F_NOP, REP8)
  CASE(6, 2, F_DEP, REP2) CASE(6, 4, F_DEP, REP4) CASE(6, 5, F_DEP, REP5) CASE(6, 8, F_DEP, REP8)
#undef CASE
#undef TAIL
}

// MFMAS = 0: the fillers alone (same loop without the MFMAs is not expressible with the macro: measured as KIND with a no-op MFMA
// replaced -- instead run N fillers x 8 per iteration through kind 5's structure); kept simple: only the mixed streams and the pure MFMA.
template <int KIND, int N, int WPS /* waves per SIMD */>
__global__ __launch_bounds__(256 * WPS, 1) void probe_kernel(int iters, unsigned long long* clk, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * ((lane * 7 + i * 3) % 17) - 0.07f); b[i] = (__bf16)(0.02f * ((lane * 5 + i) % 13) - 0.1f); }
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float f[8];
  f32x2 p[4];
  f32x4 d[4];
  for (int i = 0; i < 8; ++i) f[i] = 0.1f * lane + i;
  for (int i = 0; i < 4; ++i) { p[i] = f32x2{0.3f * lane - i, 0.05f * i}; d[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const float c0 = 0.999f, c1 = 1e-3f;
  const f32x2 q0 = {0.999f, 1.001f}, q1 = {1e-3f, -1e-3f};
  // conflict-free ds_read_b128: lane i reads 16 bytes at 16 * i of its wave's 1 KiB
  for (int i = threadIdx.x; i < 256 * WPS * 4; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = 0.001f * i;
  __syncthreads();
  const unsigned la = (unsigned)(wid * 1024 + lane * 16);
  body<KIND, N>(acc, f, p, d, a, b, c0, c1, q0, q1, la);           // warm: code in the instruction cache
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) body<KIND, N>(acc, f, p, d, a, b, c0, c1, q0, q1, la);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // MFMA results -> vector reads below (the asm bodies carry no hazard padding)
  float s = 0.f;
  for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][9];
  for (int i = 0; i < 8; ++i) s += f[i];
  for (int i = 0; i < 4; ++i) s += p[i].x + p[i].y + d[i][0] + d[i][3];
  if (s == 12345.678f) sink[threadIdx.x] = s;
  if (lane == 0) clk[blockIdx.x * (4 * WPS) + wid] = t1 - t0;
}

template <int KIND, int N, int WPS>
static void run(const char* what, int cus, int iters, unsigned long long* d_clk, float* d_sink) {
  auto kern = probe_kernel<KIND, N, WPS>;
  const int lds = 96 * 1024;                          // one workgroup per CU
  CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  if (getenv("PROBE_TRACE")) printf("launch %s N=%d wps=%d cus=%d\n", what, N, WPS, cus);
  hipLaunchKernelGGL(kern, dim3(cus), dim3(256 * WPS), lds, 0, iters, d_clk, d_sink);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  hipLaunchKernelGGL(kern, dim3(cus), dim3(256 * WPS), lds, 0, iters, d_clk, d_sink);
  CHK(hipEventRecord(e1));
  CHK(hipDeviceSynchronize());
  float ms = 0.f;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> clk(cus * 4 * WPS);
  CHK(hipMemcpy(clk.data(), d_clk, clk.size() * 8, hipMemcpyDeviceToHost));
  const double slots = 8.0 * iters;
  const double first = clk[0] / slots, worst = *std::max_element(clk.begin(), clk.end()) / slots;
  double mean = 0;
  for (auto c : clk) mean += c / slots;
  mean /= clk.size();
  // per SIMD: WPS waves share it, so cycles per MFMA *of the SIMD* = cycles per slot of a wave / WPS
  const double tf = 2.0 * 32 * 32 * 16 * 8.0 * iters * cus * 4 * WPS / (ms * 1e-3) / 1e12;
  printf("%-34s %3d CUs %d wave/SIMD  N=%2d  cyc/slot/wave: first %6.1f mean %6.1f worst %6.1f  -> per SIMD %6.1f cyc/MFMA   %8.1f us  %7.1f TF/s\n", what, cus, WPS,
         N, first, mean, worst, mean / WPS, ms * 1e3, tf);
  CHK(hipEventDestroy(e0));
  CHK(hipEventDestroy(e1));
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;
  unsigned long long* d_clk;
  float* d_sink;
  CHK(hipMalloc(&d_clk, 256 * 8 * 8));
  CHK(hipMalloc(&d_sink, 4096));
  for (int cus : {16, 256}) {
    printf("---- %d workgroups (one per CU) ----\n", cus);
#define RUN(K, N, name) run<K, N, 1>(name, cus, iters, d_clk, d_sink); run<K, N, 2>(name, cus, iters, d_clk, d_sink);
    RUN(0, 0, "mfma only")
    RUN(0, 1, "v_fma_f32") RUN(0, 2, "v_fma_f32") RUN(0, 3, "v_fma_f32") RUN(0, 4, "v_fma_f32") RUN(0, 5, "v_fma_f32") RUN(0, 6, "v_fma_f32")
    RUN(0, 8, "v_fma_f32") RUN(0, 11, "v_fma_f32") RUN(0, 16, "v_fma_f32")
    RUN(6, 2, "v_fma_f32 dependent chain") RUN(6, 4, "v_fma_f32 dependent chain") RUN(6, 5, "v_fma_f32 dependent chain") RUN(6, 8, "v_fma_f32 dependent chain")
    RUN(1, 1, "v_pk_fma_f32") RUN(1, 2, "v_pk_fma_f32") RUN(1, 3, "v_pk_fma_f32") RUN(1, 4, "v_pk_fma_f32") RUN(1, 6, "v_pk_fma_f32") RUN(1, 8, "v_pk_fma_f32")
    RUN(2, 2, "v_exp_f32") RUN(2, 4, "v_exp_f32") RUN(2, 8, "v_exp_f32")
    RUN(3, 2, "v_cvt_pk_bf16_f32") RUN(3, 4, "v_cvt_pk_bf16_f32") RUN(3, 8, "v_cvt_pk_bf16_f32")
    RUN(4, 1, "ds_read_b128") RUN(4, 2, "ds_read_b128") RUN(4, 4, "ds_read_b128")
    RUN(5, 2, "s_nop 0") RUN(5, 5, "s_nop 0") RUN(5, 8, "s_nop 0")
#undef RUN
  }
  return 0;
}
