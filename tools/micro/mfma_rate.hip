// Micro-benchmark: issue rate of v_mfma_f32_16x16x16_f16 vs v_mfma_f32_16x16x32_f16 on gfx950 (one wave per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x4 acc[8] = {};
  f16x4 a4 = {(_Float16)1.f, (_Float16)0.5f, (_Float16)threadIdx.x, (_Float16)2.f}, b4 = a4;
  f16x8 a8 = {a4[0], a4[1], a4[2], a4[3], a4[0], a4[1], a4[2], a4[3]}, b8 = a8;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (MODE == 0) acc[j] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[j], 0, 0, 0);
      else acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[j], 0, 0, 0);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}

int main() {
  float* out;
  hipMalloc(&out, 1024 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 100000, grid = 256;
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, iters);
      else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double mfmas = (double)iters * 8;  // per wave
      const double flop = mfmas * grid * 4 * (mode == 0 ? 8192.0 : 16384.0);
      printf("%s: %.3f ms, %.1f ns per MFMA per wave, %.1f TFLOP/s\n", mode == 0 ? "16x16x16_f16" : "16x16x32_f16", ms,
             ms * 1e6 / mfmas, flop / ms / 1e9);
    }
  }
  return 0;
}
