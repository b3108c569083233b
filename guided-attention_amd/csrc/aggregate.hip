// K2: aggregate_attention (utils/ptp_utils.py:273-289, select = 0).
// A[p][k] = (1/M) * sum over every head-map m of every listed tensor of map_m[p][k], M = total head-maps.
// The reference concatenates the tensors and sums dim 0; here each thread owns one (pixel, token)
// element and walks the head-maps in list order (fixed order: bitwise reproducible, no atomics).
// Reads are coalesced across the 64 lanes (token fastest), 40 x 19 712 elements in the SD-1.x case.
#include "ga_common.h"

using namespace ga;

namespace {

constexpr int kMaxMaps = 128;  // SDXL at 1024^2: 60 stored 32x32 cross maps (AggArgs stays under the 4 KB kernarg limit)

struct AggArgs {
  const void* maps[kMaxMaps];
  int heads[kMaxMaps];
  int n_maps;
  int total_heads;
};

template <typename T>
__global__ __launch_bounds__(256) void aggregate_kernel(AggArgs a, int n_elem, float* __restrict__ A) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n_elem) return;
  float acc = 0.f;
  // Four tensors x eight head-maps = 32 loads in flight per lane (the 5 x 8 maps of the SD-1.x case take two round
  // trips instead of five); the adds then run in list order within the batch: tensor-major, head-minor.  Tensors whose
  // head count is not a multiple of 8 finish in the scalar tail below, still in order.
  for (int m0 = 0; m0 < a.n_maps; m0 += 4) {
    int hmax = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (m0 + i < a.n_maps) hmax = max(hmax, a.heads[m0 + i] & ~7);
    for (int h0 = 0; h0 < hmax; h0 += 8) {
      T v[4][8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool live = m0 + i < a.n_maps && h0 + 8 <= (a.heads[m0 + i < a.n_maps ? m0 + i : 0] & ~7);
        const T* src = static_cast<const T*>(a.maps[live ? m0 + i : m0]) + e;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[i][j] = live ? src[(size_t)(h0 + j) * n_elem] : Traits<T>::zero();
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += Traits<T>::to_f32(v[i][j]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (m0 + i >= a.n_maps) continue;
      const T* src = static_cast<const T*>(a.maps[m0 + i]) + e;
      for (int h = a.heads[m0 + i] & ~7; h < a.heads[m0 + i]; ++h) acc += Traits<T>::to_f32(src[(size_t)h * n_elem]);
    }
  }
  A[e] = acc / (float)a.total_heads;
}

}  // namespace

extern "C" int ga_aggregate_maps(const void* const* maps, const int* heads, int n_maps, int npix, int Kt, float* A,
                                 int dtype, ga_stream_t stream) {
  if (!maps || !heads || !A) return GA_ERR_NULL;
  if (n_maps < 1 || n_maps > kMaxMaps || npix < 1 || Kt < 1) return GA_ERR_SHAPE;
  AggArgs a;
  a.n_maps = n_maps;
  a.total_heads = 0;
  for (int i = 0; i < n_maps; ++i) {
    if (!maps[i]) return GA_ERR_NULL;
    if (heads[i] < 1) return GA_ERR_SHAPE;
    a.maps[i] = maps[i];
    a.heads[i] = heads[i];
    a.total_heads += heads[i];
  }
  const int n_elem = npix * Kt;
  dim3 grid((n_elem + 255) / 256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case GA_F16:
      hipLaunchKernelGGL(aggregate_kernel<_Float16>, grid, dim3(256), 0, s, a, n_elem, A);
      break;
    case GA_BF16:
      hipLaunchKernelGGL(aggregate_kernel<bf16_t>, grid, dim3(256), 0, s, a, n_elem, A);
      break;
    case GA_F32:
      hipLaunchKernelGGL(aggregate_kernel<float>, grid, dim3(256), 0, s, a, n_elem, A);
      break;
    default:
      return GA_ERR_DTYPE;
  }
  return check_launch();
}
