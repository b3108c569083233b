// K5/K6 and the CFG+DDIM step: element-wise latent updates (16 384 elements for SD-1.x 512^2).
//   ga_latent_axpy  : out = latents - step*grad (+ fused mean|grad|)   pipeline_guided_attention.py:466-469
//   ga_latent_axpby : out = a*x + b*y (re-noise)                       pipeline_guided_attention.py:1048-1053
//   ga_cfg_ddim_step: CFG combine + DDIM eta=0 update                  pipeline_guided_attention.py:1022-1029
// Launch-latency bound; one pass, math in f32, one rounding to T at the store.
#include "ga_common.h"

using namespace ga;

namespace {

template <typename T>
__global__ __launch_bounds__(256) void axpby_kernel(const T* __restrict__ x, const T* __restrict__ y, float a, float b,
                                                    T* __restrict__ out, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    out[i] = Traits<T>::from_f32(a * Traits<T>::to_f32(x[i]) + b * Traits<T>::to_f32(y[i]));
}

template <typename T>
__device__ __forceinline__ T axpy_elem(T x, float step, float g) {
  return Traits<T>::from_f32(Traits<T>::to_f32(x) - step * g);
}

template <typename T>
__global__ __launch_bounds__(256) void axpy_kernel(const T* __restrict__ x, const T* __restrict__ g, float step,
                                                   T* __restrict__ out, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    out[i] = axpy_elem<T>(x[i], step, Traits<T>::to_f32(g[i]));
}

// single workgroup: also reduces sum|grad| deterministically (same element formula as axpy_kernel)
template <typename T>
__global__ __launch_bounds__(1024) void axpy_absmean_kernel(const T* __restrict__ x, const T* __restrict__ g,
                                                            float step, T* __restrict__ out,
                                                            float* __restrict__ absmean, long long n) {
  __shared__ float part[16];
  float acc = 0.f;
  for (long long i = threadIdx.x; i < n; i += 1024) {
    const float gv = Traits<T>::to_f32(g[i]);
    acc += fabsf(gv);
    out[i] = axpy_elem<T>(x[i], step, gv);
  }
  acc = wave_reduce_sum(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < 16; ++w) s += part[w];
    absmean[0] = s / (float)n;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void cfg_ddim_kernel(const T* __restrict__ eu, const T* __restrict__ et, float gs,
                                                       const T* __restrict__ x, float sa_t, float s1_t, float sa_p,
                                                       float s1_p, T* __restrict__ prev, T* __restrict__ x0o,
                                                       long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float u = Traits<T>::to_f32(eu[i]);
    const float eps = u + gs * (Traits<T>::to_f32(et[i]) - u);
    const float x0 = (Traits<T>::to_f32(x[i]) - s1_t * eps) / sa_t;
    if (x0o) x0o[i] = Traits<T>::from_f32(x0);
    prev[i] = Traits<T>::from_f32(sa_p * x0 + s1_p * eps);
  }
}

int grid_for(long long n) { return (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048); }

template <typename T>
int do_axpy(const void* x, const void* g, float step, void* out, float* absmean, long long n, hipStream_t s) {
  if (absmean)
    hipLaunchKernelGGL(axpy_absmean_kernel<T>, dim3(1), dim3(1024), 0, s, (const T*)x, (const T*)g, step, (T*)out,
                       absmean, n);
  else
    hipLaunchKernelGGL(axpy_kernel<T>, dim3(grid_for(n)), dim3(256), 0, s, (const T*)x, (const T*)g, step, (T*)out, n);
  return check_launch();
}

}  // namespace

extern "C" int ga_latent_axpy(const void* latents, const void* grad, float step, void* out, float* absmean, int64_t n,
                              int dtype, ga_stream_t stream) {
  if (!latents || !grad || !out) return GA_ERR_NULL;
  if (n < 1) return GA_ERR_SHAPE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case GA_F16:
      return do_axpy<_Float16>(latents, grad, step, out, absmean, n, s);
    case GA_BF16:
      return do_axpy<bf16_t>(latents, grad, step, out, absmean, n, s);
    case GA_F32:
      return do_axpy<float>(latents, grad, step, out, absmean, n, s);
    default:
      return GA_ERR_DTYPE;
  }
}

extern "C" int ga_latent_axpby(const void* x, const void* y, float a, float b, void* out, int64_t n, int dtype,
                               ga_stream_t stream) {
  if (!x || !y || !out) return GA_ERR_NULL;
  if (n < 1) return GA_ERR_SHAPE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid(grid_for(n));
  switch (dtype) {
    case GA_F16:
      hipLaunchKernelGGL(axpby_kernel<_Float16>, grid, dim3(256), 0, s, (const _Float16*)x, (const _Float16*)y, a, b,
                         (_Float16*)out, (long long)n);
      break;
    case GA_BF16:
      hipLaunchKernelGGL(axpby_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)y, a, b,
                         (bf16_t*)out, (long long)n);
      break;
    case GA_F32:
      hipLaunchKernelGGL(axpby_kernel<float>, grid, dim3(256), 0, s, (const float*)x, (const float*)y, a, b,
                         (float*)out, (long long)n);
      break;
    default:
      return GA_ERR_DTYPE;
  }
  return check_launch();
}

extern "C" int ga_cfg_ddim_step(const void* eps_uncond, const void* eps_text, float guidance, const void* x,
                                float alpha_t, float alpha_prev, void* prev, void* x0_out, int64_t n, int dtype,
                                ga_stream_t stream) {
  if (!eps_uncond || !eps_text || !x || !prev) return GA_ERR_NULL;
  if (n < 1 || !(alpha_t > 0.f) || !(alpha_prev > 0.f) || alpha_t > 1.f || alpha_prev > 1.f) return GA_ERR_SHAPE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid(grid_for(n));
  const float sa_t = sqrtf(alpha_t), s1_t = sqrtf(1.0f - alpha_t), sa_p = sqrtf(alpha_prev), s1_p = sqrtf(1.0f - alpha_prev);
  switch (dtype) {
    case GA_F16:
      hipLaunchKernelGGL(cfg_ddim_kernel<_Float16>, grid, dim3(256), 0, s, (const _Float16*)eps_uncond,
                         (const _Float16*)eps_text, guidance, (const _Float16*)x, sa_t, s1_t, sa_p, s1_p,
                         (_Float16*)prev, (_Float16*)x0_out, (long long)n);
      break;
    case GA_BF16:
      hipLaunchKernelGGL(cfg_ddim_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)eps_uncond,
                         (const bf16_t*)eps_text, guidance, (const bf16_t*)x, sa_t, s1_t, sa_p, s1_p, (bf16_t*)prev,
                         (bf16_t*)x0_out, (long long)n);
      break;
    case GA_F32:
      hipLaunchKernelGGL(cfg_ddim_kernel<float>, grid, dim3(256), 0, s, (const float*)eps_uncond,
                         (const float*)eps_text, guidance, (const float*)x, sa_t, s1_t, sa_p, s1_p, (float*)prev,
                         (float*)x0_out, (long long)n);
      break;
    default:
      return GA_ERR_DTYPE;
  }
  return check_launch();
}

extern "C" int ga_version(void) { return GA_VERSION; }

extern "C" const char* ga_strerror(int status) {
  switch (status) {
    case GA_OK: return "ok";
    case GA_ERR_NULL: return "required pointer is NULL";
    case GA_ERR_SHAPE: return "size out of the supported range";
    case GA_ERR_DTYPE: return "unknown dtype";
    case GA_ERR_ALIGN: return "pointer not 16-byte aligned or head_dim not a multiple of 8";
    case GA_ERR_LAUNCH: return "kernel launch failed";
    case GA_ERR_UNSUPPORTED: return "not implemented in this build";
    default: return "unknown status";
  }
}
