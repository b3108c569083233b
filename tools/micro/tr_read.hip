// Micro-test: what ds_read_b64_tr_b16 delivers.  LDS holds element value = its index; lane i of each 16-lane group
// addresses row (i>>2), columns 4(i&3).. of a 4 x 16 block (row stride 16 elements).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ short lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int lane = threadIdx.x, grp = lane >> 4, i = lane & 15;
  const short* p = lds + grp * 64 + (i >> 2) * 16 + 4 * (i & 3);
  s16x4 w = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = w[j];
}
int main() {
  short* d;
  hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short h[256];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  return 0;
}
