#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out) {
  __shared__ unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = i;
  __syncthreads();
  // lane l points at 4 elements starting at element 4*l  (8-byte aligned)
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(lds + 4 * threadIdx.x));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = r[j];
  // second experiment: row-major matrix [row][64 cols]: lane l -> row (l&3) + 4*(l>>4), cols 4*((l>>2)&3)..+3
  int l = threadIdx.x;
  int row = (l & 3) + 4 * (l >> 4), col = 4 * ((l >> 2) & 3);
  v4s r2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(lds + row * 64 + col));
  for (int j = 0; j < 4; ++j) out[256 + threadIdx.x * 4 + j] = r2[j];
}
int main() {
  unsigned short* d; hipMalloc(&d, 1024);
  k<<<1, 64>>>(d);
  unsigned short h[512]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
  printf("exp1: lane -> 4 values (input element indices; lane l supplied elements 4l..4l+3)\n");
  for (int l = 0; l < 64; ++l) printf("%2d: %4d %4d %4d %4d%s", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3], (l%4==3)?"\n":"   |  ");
  printf("exp2: lane -> (row,col) pairs, lane l supplied row (l&3)+4*(l>>4), cols 4*((l>>2)&3)..\n");
  for (int l = 0; l < 64; ++l) { printf("%2d:", l); for (int j=0;j<4;++j) printf(" (%d,%d)", h[256+l*4+j]/64, h[256+l*4+j]%64); printf("%s", (l%2==1)?"\n":"   |  "); }
  return 0;
}
