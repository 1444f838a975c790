// Operand layout and scale semantics of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 x fp8 e4m3) on gfx950, found by experiment.
//   build: hipcc --offload-arch=gfx950 -O2 mx8_probe.cpp -o mx8_probe
// Every lane l supplies 32 bytes of A (row i = l & 31), 32 bytes of B (column j = l & 31) and one scale register per operand.
// Experiments:
//   1. unit scales, random small-integer e4m3 values, "same convention" packing (slot s of lane half lh of A multiplies slot s of lane half lh
//      of B): D[i][j] in the 32x32 C layout against the CPU sum -> the instruction is used correctly at all, C layout as for bf16.
//   2. k-order: A one-hot in slot sA of every lane, B one-hot in slot sB of every lane -> which (lhA, sA) meets which (lhB, sB).
//   3. scales: per-lane random exponents; which lane's scale byte is applied to which of a lane's 32 values (hypotheses H1: the lane's own
//      byte for all 32; H2: byte of lane (i + 32 * (s / 16)) for slot s); and which byte of the register op_sel picks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int OA, int OB>
__global__ void mm(const unsigned char* A, const unsigned char* B, const int* sa, const int* sb, float* D) {
  const int l = threadIdx.x;
  v8i a, b;
  for (int v = 0; v < 8; ++v) {
    a[v] = reinterpret_cast<const int*>(A + l * 32)[v];
    b[v] = reinterpret_cast<const int*>(B + l * 32)[v];
  }
  v16f c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, OA, sa[l], OB, sb[l]);
  for (int r = 0; r < 16; ++r) D[l * 16 + r] = c[r];
}

static float e4m3(unsigned char v) {      // OCP e4m3fn
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float x = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -x : x;
}
static unsigned char enc_small_int(int n) {      // |n| <= 8, exact
  if (n == 0) return 0;
  unsigned char s = n < 0 ? 0x80 : 0;
  int a = abs(n), e = 0;
  while ((1 << (e + 1)) <= a) ++e;
  const int m = (a * 8 >> e) - 8;               // a = (1 + m/8) 2^e
  return s | ((e + 7) << 3) | m;
}

int main() {
  unsigned char hA[64 * 32], hB[64 * 32];
  int hsa[64], hsb[64];
  float hD[64 * 16];
  unsigned char *dA, *dB; int *dsa, *dsb; float* dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dsa, sizeof hsa); hipMalloc(&dsb, sizeof hsb); hipMalloc(&dD, sizeof hD);
  auto run = [&](int oa, int ob) {
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipMemcpy(dsa, hsa, sizeof hsa, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, sizeof hsb, hipMemcpyHostToDevice);
    if (oa == 0 && ob == 0) mm<0, 0><<<1, 64>>>(dA, dB, dsa, dsb, dD);
    else if (oa == 1) mm<1, 0><<<1, 64>>>(dA, dB, dsa, dsb, dD);
    else if (oa == 2) mm<2, 0><<<1, 64>>>(dA, dB, dsa, dsb, dD);
    else if (oa == 3) mm<3, 0><<<1, 64>>>(dA, dB, dsa, dsb, dD);
    else mm<0, 1><<<1, 64>>>(dA, dB, dsa, dsb, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  };
  // D[i][j] from the 32x32 C layout: lane l holds column j = l & 31, rows (r & 3) + 8 (r >> 2) + 4 (l >> 5)
  auto Dat = [&](int i, int j) {
    for (int lh = 0; lh < 2; ++lh)
      for (int r = 0; r < 16; ++r)
        if ((r & 3) + 8 * (r >> 2) + 4 * lh == i) return hD[(j + 32 * lh) * 16 + r];
    return NAN;
  };
  srand(1);
  // ---- experiment 1 ---------------------------------------------------------------------------------------------------------------------
  for (int i = 0; i < 64 * 32; ++i) { hA[i] = enc_small_int(rand() % 9 - 4); hB[i] = enc_small_int(rand() % 9 - 4); }
  for (int l = 0; l < 64; ++l) hsa[l] = hsb[l] = 0x7F7F7F7F;
  run(0, 0);
  double worst = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      double s = 0;
      for (int lh = 0; lh < 2; ++lh)
        for (int sl = 0; sl < 32; ++sl) s += e4m3(hA[(i + 32 * lh) * 32 + sl]) * e4m3(hB[(j + 32 * lh) * 32 + sl]);
      worst = fmax(worst, fabs(s - Dat(i, j)));
    }
  printf("exp1 (unit scales, slot s of lane half lh of A x the same slot of B, rows / columns = lane & 31, bf16 C layout): max |diff| = %g\n", worst);
  // ---- experiment 2: which slots meet ---------------------------------------------------------------------------------------------------
  printf("exp2: count of products D[0][0] for (A: value 1 in slot sA of lanes 0 and 32 = row 0) x (B: value 1 in slot sB of lanes 0 and 32 = column 0)\n");
  printf("      and separately per lane half; listed: sA -> the (lhB, sB) it meets when A sits in lane half lhA\n");
  for (int lhA = 0; lhA < 2; ++lhA)
    for (int sA = 0; sA < 32; sA += 1) {
      memset(hA, 0, sizeof hA);
      hA[(0 + 32 * lhA) * 32 + sA] = enc_small_int(1);
      int found = 0;
      for (int lhB = 0; lhB < 2 && !found; ++lhB)
        for (int sB = 0; sB < 32 && !found; ++sB) {
          memset(hB, 0, sizeof hB);
          hB[(0 + 32 * lhB) * 32 + sB] = enc_small_int(1);
          run(0, 0);
          if (Dat(0, 0) == 1.0f) { if (lhA != lhB || sA != sB) printf("      A(lh %d, slot %2d) meets B(lh %d, slot %2d)\n", lhA, sA, lhB, sB); found = 1; }
        }
      if (!found) printf("      A(lh %d, slot %2d) meets NOTHING\n", lhA, sA);
    }
  printf("      (pairs not listed meet the same (lane half, slot) of the other operand)\n");
  // ---- experiment 3: scales -------------------------------------------------------------------------------------------------------------
  for (int i = 0; i < 64 * 32; ++i) { hA[i] = enc_small_int(rand() % 9 - 4); hB[i] = enc_small_int(rand() % 9 - 4); }
  int ea[64], eb[64];
  for (int l = 0; l < 64; ++l) { ea[l] = 125 + rand() % 5; eb[l] = 125 + rand() % 5; hsa[l] = ea[l] * 0x01010101; hsb[l] = eb[l] * 0x01010101; }
  run(0, 0);
  double w1 = 0, w2 = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      double s1 = 0, s2 = 0;
      for (int lh = 0; lh < 2; ++lh)
        for (int sl = 0; sl < 32; ++sl) {
          const double p = e4m3(hA[(i + 32 * lh) * 32 + sl]) * e4m3(hB[(j + 32 * lh) * 32 + sl]);
          s1 += p * ldexp(1.0, ea[i + 32 * lh] - 127) * ldexp(1.0, eb[j + 32 * lh] - 127);                       // H1: the lane's own scale for all 32
          s2 += p * ldexp(1.0, ea[i + 32 * (sl / 16)] - 127) * ldexp(1.0, eb[j + 32 * (sl / 16)] - 127);           // H2: slots 0-15 lane i's, 16-31 lane i+32's
        }
      w1 = fmax(w1, fabs(s1 - Dat(i, j)));
      w2 = fmax(w2, fabs(s2 - Dat(i, j)));
    }
  printf("exp3 (random scale bytes 125..129, replicated in all four byte lanes): max |diff| under H1 (a lane's own byte scales its 32 values) = %g, "
         "under H2 (slots 0-15 use lane i's byte, slots 16-31 lane i+32's) = %g\n", w1, w2);
  // which byte does op_sel pick: bytes 0..3 of the A scale register = 127, 128, 129, 130 -> D is multiplied by 1, 2, 4, 8
  for (int i = 0; i < 64 * 32; ++i) { hA[i] = enc_small_int(1); hB[i] = enc_small_int(1); }
  for (int l = 0; l < 64; ++l) { hsa[l] = 0x82818079 + 0x06; hsb[l] = 0x7F7F7F7F; }      // bytes (low to high) 0x7F, 0x80, 0x81, 0x82
  for (int o = 0; o < 4; ++o) { run(o, 0); printf("exp3b: op_sel_a = %d with A-scale bytes (low..high) 7F 80 81 82: D[0][0] = %g (64 x 2^n)\n", o, Dat(0, 0)); }
  for (int l = 0; l < 64; ++l) { hsb[l] = 0x82818079 + 0x06; hsa[l] = 0x7F7F7F7F; }
  run(0, 1); printf("exp3b: op_sel_b = 1 with B-scale bytes 7F 80 81 82: D[0][0] = %g\n", Dat(0, 0));
  return 0;
}
