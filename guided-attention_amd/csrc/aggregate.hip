// K2: aggregate_attention (utils/ptp_utils.py:273-289, select = 0).
// A[p][k] = (1/M) * sum over every head-map m of every listed tensor of map_m[p][k], M = total head-maps.
// The reference concatenates the tensors and sums dim 0; here each thread owns one (pixel, token)
// element and walks the head-maps in list order (fixed order: bitwise reproducible, no atomics).
// Reads are coalesced across the 64 lanes (token fastest), 40 x 19 712 elements in the SD-1.x case.
#include "aggregate.h"

using namespace ga;

namespace {

template <typename T>
__global__ __launch_bounds__(256) void aggregate_kernel(AggArgs a, int n_elem, float* __restrict__ A) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < n_elem) A[e] = aggregate_element<T>(a, e, n_elem);
}

}  // namespace

extern "C" int ga_aggregate_maps(const void* const* maps, const int* heads, int n_maps, int npix, int Kt, float* A,
                                 int dtype, ga_stream_t stream) {
  if (!A) return GA_ERR_NULL;
  if (npix < 1 || Kt < 1) return GA_ERR_SHAPE;
  AggArgs a;
  const int rc = fill_agg_args(a, maps, heads, n_maps);
  if (rc != GA_OK) return rc;
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
