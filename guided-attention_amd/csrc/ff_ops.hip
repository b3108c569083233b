// Element-wise epilogues of the UNet blocks the guided-attention step runs (reference: the diffusers 0.12.1 UNet
// called from pipeline_guided_attention.py:583-743): HBM-bound one-pass kernels, 16-byte accesses, math in f32.
//   ga_geglu_fwd / _bwd     : y = h * gelu(gate), [h | gate] = the two halves of the GEGLU projection's output row
//                             (PyTorch: chunk + gelu + mul = 2 launches and 52 MB of traffic on the 64x64 level where
//                             this moves 31 MB; its backward is 5 launches, here one)
//   ga_bias_residual_add    : out = y + bias[c] + residual (ResnetBlock2D: conv2 bias + skip connection in one pass
//                             instead of MIOpen's separate bias add followed by the residual add)
#include <math.h>

#include "ga_common.h"

using namespace ga;

namespace {

constexpr int kThreads = 256;

template <typename T>
struct alignas(16) Vec {
  static constexpr int N = 16 / sizeof(T);
  T v[N];
};

__device__ __forceinline__ float gelu_cdf(float g) { return 0.5f * (1.0f + erff(g * 0.70710678118654752f)); }

template <typename T>
__global__ __launch_bounds__(kThreads) void geglu_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                             long long rows, int fv) {
  constexpr int N = Vec<T>::N;
  const Vec<T>* xv = reinterpret_cast<const Vec<T>*>(x);
  Vec<T>* yv = reinterpret_cast<Vec<T>*>(y);
  const long long total = rows * fv;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < total; i += (long long)gridDim.x * kThreads) {
    const long long r = i / fv;
    const int j = (int)(i - r * fv);
    const Vec<T> h = xv[r * 2 * fv + j], gt = xv[r * 2 * fv + fv + j];
    Vec<T> o;
#pragma unroll
    for (int e = 0; e < N; ++e) {
      const float g = Traits<T>::to_f32(gt.v[e]);
      o.v[e] = Traits<T>::from_f32(Traits<T>::to_f32(h.v[e]) * (g * gelu_cdf(g)));
    }
    yv[i] = o;
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void geglu_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                             T* __restrict__ dx, long long rows, int fv) {
  constexpr int N = Vec<T>::N;
  const Vec<T>* xv = reinterpret_cast<const Vec<T>*>(x);
  const Vec<T>* dv = reinterpret_cast<const Vec<T>*>(dy);
  Vec<T>* ov = reinterpret_cast<Vec<T>*>(dx);
  const long long total = rows * fv;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < total; i += (long long)gridDim.x * kThreads) {
    const long long r = i / fv;
    const int j = (int)(i - r * fv);
    const Vec<T> h = xv[r * 2 * fv + j], gt = xv[r * 2 * fv + fv + j], d = dv[i];
    Vec<T> oh, og;
#pragma unroll
    for (int e = 0; e < N; ++e) {
      const float g = Traits<T>::to_f32(gt.v[e]), dd = Traits<T>::to_f32(d.v[e]);
      const float cdf = gelu_cdf(g);
      const float pdf = 0.3989422804014327f * __expf(-0.5f * g * g);
      oh.v[e] = Traits<T>::from_f32(dd * (g * cdf));
      og.v[e] = Traits<T>::from_f32(dd * Traits<T>::to_f32(h.v[e]) * (cdf + g * pdf));
    }
    ov[r * 2 * fv + j] = oh;
    ov[r * 2 * fv + fv + j] = og;
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void bias_residual_kernel(const T* __restrict__ y, const T* __restrict__ bias,
                                                                 const T* __restrict__ res, T* __restrict__ out,
                                                                 long long rows, int cv) {
  constexpr int N = Vec<T>::N;
  const Vec<T>* yv = reinterpret_cast<const Vec<T>*>(y);
  const Vec<T>* rv = reinterpret_cast<const Vec<T>*>(res);
  const Vec<T>* bv = reinterpret_cast<const Vec<T>*>(bias);
  Vec<T>* ov = reinterpret_cast<Vec<T>*>(out);
  const long long total = rows * cv;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < total; i += (long long)gridDim.x * kThreads) {
    const Vec<T> a = yv[i], b = rv[i];
    Vec<T> o;
    if (bias != nullptr) {
      const Vec<T> c = bv[i % cv];
#pragma unroll
      for (int e = 0; e < N; ++e)
        o.v[e] = Traits<T>::from_f32(Traits<T>::to_f32(a.v[e]) + Traits<T>::to_f32(c.v[e]) + Traits<T>::to_f32(b.v[e]));
    } else {
#pragma unroll
      for (int e = 0; e < N; ++e) o.v[e] = Traits<T>::from_f32(Traits<T>::to_f32(a.v[e]) + Traits<T>::to_f32(b.v[e]));
    }
    ov[i] = o;
  }
}

inline int grid_for(long long vectors) {
  const long long wg = (vectors + kThreads - 1) / kThreads;
  return (int)(wg < 4096 ? wg : 4096);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <typename T>
int geglu_fwd(const void* x, void* y, long long rows, int F, hipStream_t s) {
  constexpr int N = Vec<T>::N;
  if (F % N) return GA_ERR_SHAPE;
  hipLaunchKernelGGL(geglu_fwd_kernel<T>, dim3(grid_for(rows * (F / N))), dim3(kThreads), 0, s, (const T*)x, (T*)y, rows,
                     F / N);
  return check_launch();
}

template <typename T>
int geglu_bwd(const void* x, const void* dy, void* dx, long long rows, int F, hipStream_t s) {
  constexpr int N = Vec<T>::N;
  if (F % N) return GA_ERR_SHAPE;
  hipLaunchKernelGGL(geglu_bwd_kernel<T>, dim3(grid_for(rows * (F / N))), dim3(kThreads), 0, s, (const T*)x,
                     (const T*)dy, (T*)dx, rows, F / N);
  return check_launch();
}

template <typename T>
int bias_residual(const void* y, const void* bias, const void* res, void* out, long long rows, int C, hipStream_t s) {
  constexpr int N = Vec<T>::N;
  if (C % N) return GA_ERR_SHAPE;
  hipLaunchKernelGGL(bias_residual_kernel<T>, dim3(grid_for(rows * (C / N))), dim3(kThreads), 0, s, (const T*)y,
                     (const T*)bias, (const T*)res, (T*)out, rows, C / N);
  return check_launch();
}

}  // namespace

extern "C" int ga_geglu_fwd(const void* x, void* y, int64_t rows, int F, int dtype, ga_stream_t stream) {
  if (!x || !y) return GA_ERR_NULL;
  if (rows < 1 || F < 1) return GA_ERR_SHAPE;
  if (!aligned16(x) || !aligned16(y)) return GA_ERR_ALIGN;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case GA_F16: return geglu_fwd<_Float16>(x, y, rows, F, s);
    case GA_BF16: return geglu_fwd<bf16_t>(x, y, rows, F, s);
    case GA_F32: return geglu_fwd<float>(x, y, rows, F, s);
    default: return GA_ERR_DTYPE;
  }
}

extern "C" int ga_geglu_bwd(const void* x, const void* dy, void* dx, int64_t rows, int F, int dtype,
                            ga_stream_t stream) {
  if (!x || !dy || !dx) return GA_ERR_NULL;
  if (rows < 1 || F < 1) return GA_ERR_SHAPE;
  if (!aligned16(x) || !aligned16(dy) || !aligned16(dx)) return GA_ERR_ALIGN;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case GA_F16: return geglu_bwd<_Float16>(x, dy, dx, rows, F, s);
    case GA_BF16: return geglu_bwd<bf16_t>(x, dy, dx, rows, F, s);
    case GA_F32: return geglu_bwd<float>(x, dy, dx, rows, F, s);
    default: return GA_ERR_DTYPE;
  }
}

extern "C" int ga_bias_residual_add(const void* y, const void* bias, const void* residual, void* out, int64_t rows,
                                    int C, int dtype, ga_stream_t stream) {
  if (!y || !residual || !out) return GA_ERR_NULL;
  if (rows < 1 || C < 1) return GA_ERR_SHAPE;
  if (!aligned16(y) || !aligned16(residual) || !aligned16(out) || (bias && !aligned16(bias))) return GA_ERR_ALIGN;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case GA_F16: return bias_residual<_Float16>(y, bias, residual, out, rows, C, s);
    case GA_BF16: return bias_residual<bf16_t>(y, bias, residual, out, rows, C, s);
    case GA_F32: return bias_residual<float>(y, bias, residual, out, rows, C, s);
    default: return GA_ERR_DTYPE;
  }
}
