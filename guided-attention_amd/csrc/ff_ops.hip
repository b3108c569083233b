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

// out[m][0 .. C1) = a[m][:], out[m][C1 .. C1 + C2) = b[m][:] in 16-byte vectors: the UpBlock's channel concatenation of the
// running activation with the skip connection on channels-last tensors (diffusers 0.12.1 CrossAttnUpBlock2D / UpBlock2D:
// `torch.cat([hidden_states, res_hidden_states], dim=1)`, run by the reference inside pipeline_guided_attention.py:583-743).
// Each source is walked densely (vector i of a source is element i of its array: coalesced, no address select), four vectors
// per thread with all four loads issued before the first store; only the destination row needs a division (multiply-high).
constexpr int kCatPerThread = 4;
// row = i / vp without a branch: multiply-high by the host's reciprocal, plus i itself where vp == 1 (reciprocal 0, mask ~0).
// fdiv()'s `m != 0 ? ... : x` compiles to a (uniform) branch per use — four basic blocks the loads were then sunk into.
struct CatDiv {
  unsigned m, mask;
};
__device__ __forceinline__ void cat_part(const uint4* __restrict__ src, uint4* __restrict__ out, long long n, int vp, int off,
                                         int vt, CatDiv d_vp, long long base, long long stride) {
  auto dest = [&](long long i) -> uint4* {
    const int row = (int)(__umulhi((unsigned)i, d_vp.m) + ((unsigned)i & d_vp.mask)), col = (int)i - row * vp;
    return out + ((long long)row * vt + off + col);
  };
  if (base + (kCatPerThread - 1) * stride < n) {
    // all four in range (every thread but the last few): nothing is conditional, the four loads go out together — with each
    // store under `if (i < n)` the compiler moved its load in there too: load, wait, store, four times in a row
    uint4 v[kCatPerThread];
#pragma unroll
    for (int k = 0; k < kCatPerThread; ++k) v[k] = src[base + k * stride];
    __builtin_amdgcn_sched_barrier(0);   // left alone the scheduler pairs each load with its store: load, wait, store, ...
#pragma unroll
    for (int k = 0; k < kCatPerThread; ++k) *dest(base + k * stride) = v[k];
  } else {
    for (int k = 0; k < kCatPerThread; ++k) {
      const long long i = base + k * stride;
      if (i < n) *dest(i) = src[i];
    }
  }
}

__global__ __launch_bounds__(kThreads) void cat_rows_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b,
                                                            uint4* __restrict__ out, long long rows, int v1, int v2,
                                                            CatDiv d_v1, CatDiv d_v2) {
  const long long na = rows * v1, nb = rows * v2, nmax = na > nb ? na : nb;
  const long long base = (long long)blockIdx.x * kThreads + threadIdx.x;
  // == gridDim.x * kThreads, from the (preloaded) arguments instead of the hidden grid-size argument
  const long long stride = ((nmax + kThreads * kCatPerThread - 1) / (kThreads * kCatPerThread)) * kThreads;
  cat_part(a, out, na, v1, 0, v1 + v2, d_v1, base, stride);
  cat_part(b, out, nb, v2, v1, v1 + v2, d_v2, base, stride);
}

}  // namespace

extern "C" int ga_cat_channels(const void* a, const void* b, void* out, int64_t rows, int C1, int C2, int elem_bytes,
                               ga_stream_t stream) {
  if (!a || !b || !out) return GA_ERR_NULL;
  if (rows < 1 || C1 < 1 || C2 < 1 || (elem_bytes != 2 && elem_bytes != 4)) return GA_ERR_SHAPE;
  const int per = 16 / elem_bytes;
  if (C1 % per != 0 || C2 % per != 0) return GA_ERR_SHAPE;
  if (!aligned16(a) || !aligned16(b) || !aligned16(out)) return GA_ERR_ALIGN;
  const int v1 = C1 / per, v2 = C2 / per;
  const long long na = (long long)rows * v1, nb = (long long)rows * v2, nmax = na > nb ? na : nb;
  if (na >= (1LL << 31) || nb >= (1LL << 31)) return GA_ERR_SHAPE;   // 32-bit vector index in the row division
  bool ok = true;
  const FastDiv d1 = make_fastdiv(v1, (unsigned long long)na, ok), d2 = make_fastdiv(v2, (unsigned long long)nb, ok);
  if (!ok) return GA_ERR_SHAPE;
  const long long per_wg = (long long)kThreads * kCatPerThread;
  const unsigned grid = (unsigned)((nmax + per_wg - 1) / per_wg);
  const CatDiv c1{d1.m, d1.m ? 0u : ~0u}, c2{d2.m, d2.m ? 0u : ~0u};
  hipLaunchKernelGGL(cat_rows_kernel, dim3(grid), dim3(kThreads), 0, static_cast<hipStream_t>(stream), (const uint4*)a,
                     (const uint4*)b, (uint4*)out, (long long)rows, v1, v2, c1, c2);
  return check_launch();
}

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
