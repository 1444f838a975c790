// Probe: what do HW_ID / LDS_ALLOC look like for co-resident workgroups on gfx950?  (timing-experiment helper)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256, 2) void probe(unsigned* out, int spin) {
  extern __shared__ char smem[];
  unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);     // HW_ID, 32 bits
  unsigned lds = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 6);    // LDS_ALLOC
  unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);   // XCC_ID (gfx940+)
  unsigned long long t0 = __builtin_readcyclecounter();
  volatile char* s = smem;
  for (int i = 0; i < spin; ++i) s[threadIdx.x] = (char)i;                // stay resident for a while
  if ((threadIdx.x & 63) == 0) {
    unsigned* o = out + (blockIdx.x * 4 + (threadIdx.x >> 6)) * 6;
    o[0] = hw; o[1] = lds; o[2] = xcc; o[3] = (unsigned)t0; o[4] = (unsigned)(t0 >> 32); o[5] = blockIdx.x;
  }
}
int main() {
  const int nb = 1024;
  unsigned* d; hipMalloc(&d, nb * 4 * 6 * 4);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 74000);
  hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 74000, 0, d, 20000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(nb * 4 * 6);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  // histogram of (LDS base, TG_ID, WAVE_ID) over all workgroups + start-time spread of co-resident pairs
  int hist_base[4096] = {0}, hist_tg[16] = {0}, hist_wave[16] = {0};
  for (int b = 0; b < nb; ++b) {
    unsigned* o = &h[(b * 4) * 6];
    hist_base[o[1] & 0xFFF]++; hist_tg[(o[0] >> 16) & 15]++; hist_wave[o[0] & 15]++;
  }
  for (int i = 0; i < 4096; ++i) if (hist_base[i]) printf("LDS base %d: %d workgroups\n", i, hist_base[i]);
  for (int i = 0; i < 16; ++i) if (hist_tg[i]) printf("TG_ID %d: %d workgroups\n", i, hist_tg[i]);
  for (int i = 0; i < 16; ++i) if (hist_wave[i]) printf("WAVE_ID %d: %d workgroups (wave 0 of the group)\n", i, hist_wave[i]);
  // pairs on the same (xcc, se, cu): start time difference
  int shown = 0;
  for (int a = 0; a < nb && shown < 12; ++a)
    for (int b = a + 1; b < nb && shown < 12; ++b) {
      unsigned* x = &h[(a * 4) * 6]; unsigned* y = &h[(b * 4) * 6];
      if (x[2] == y[2] && ((x[0] >> 8) & 0xFF) == ((y[0] >> 8) & 0xFF)) {
        long long ta = ((long long)x[4] << 32) | x[3], tb = ((long long)y[4] << 32) | y[3];
        printf("same CU: wg %d (base %u tg %u wave %u) and wg %d (base %u tg %u wave %u): start delta %lld cycles\n", a, x[1] & 0xFFF, (x[0] >> 16) & 15, x[0] & 15,
               b, y[1] & 0xFFF, (y[0] >> 16) & 15, y[0] & 15, tb - ta);
        ++shown;
      }
    }
  return 0;
}
