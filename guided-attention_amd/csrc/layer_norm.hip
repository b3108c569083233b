// Residual add + LayerNorm in one pass, forward and backward to the input — the `x = attn(...) + x; LayerNorm(x)`
// pairs of the transformer blocks (diffusers 0.12.1 BasicTransformerBlock.forward, run by the reference inside
// pipeline_guided_attention.py:583-743).  PyTorch runs the add and the norm as two kernels forward, and in the
// backward a LayerNorm-backward plus the autograd accumulation add of the two branches that meet at x; here both
// directions are one launch.
//
// gfx950 mapping: one 64-lane wave per token row (C = 320 / 640 / 1280 -> 1-3 16-byte vectors per lane, the row
// lives in registers between the statistics and the normalisation), 4 rows per workgroup, no LDS, no barriers;
// mean and variance are two in-register passes (no E[x^2] - E[x]^2 cancellation).  HBM-bound: forward moves
// 4 x rows x C elements (a, x in; x_new, y out), backward 4 x (x_new, dy, g_res in; dx out).
//
// The sum a + x is rounded to T first and the statistics are taken from the rounded row, exactly what PyTorch's
// LayerNorm sees when it reads the 16-bit tensor the add kernel wrote; the backward recomputes x_hat from the same
// stored row.
#include "ga_common.h"

using namespace ga;

namespace {

constexpr int kThreads = 256;
constexpr int kRowsPerWg = kThreads / 64;

template <typename T>
struct alignas(16) Vec {
  static constexpr int N = 16 / sizeof(T);
  T v[N];
};

template <typename T, int NV>
__global__ __launch_bounds__(kThreads) void add_ln_fwd_kernel(const T* __restrict__ a, const T* __restrict__ x,
                                                              const T* __restrict__ gamma, const T* __restrict__ beta,
                                                              T* __restrict__ xnew, T* __restrict__ y,
                                                              float* __restrict__ stats, long long rows, int C,
                                                              float eps) {
  constexpr int N = Vec<T>::N;
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * kRowsPerWg + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int cv = C / N;
  const Vec<T>* xr = reinterpret_cast<const Vec<T>*>(x + row * C);
  const Vec<T>* ar = a ? reinterpret_cast<const Vec<T>*>(a + row * C) : nullptr;
  float f[NV][N];
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int j = lane + 64 * k;
    if (j < cv) {
      Vec<T> v = xr[j];
      if (ar != nullptr) {
        const Vec<T> w = ar[j];
#pragma unroll
        for (int e = 0; e < N; ++e) v.v[e] = Traits<T>::from_f32(Traits<T>::to_f32(v.v[e]) + Traits<T>::to_f32(w.v[e]));
        reinterpret_cast<Vec<T>*>(xnew + row * C)[j] = v;
      }
#pragma unroll
      for (int e = 0; e < N; ++e) {
        f[k][e] = Traits<T>::to_f32(v.v[e]);
        sum += f[k][e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < N; ++e) f[k][e] = 0.f;
    }
  }
  const float mean = wave_sum(sum) / (float)C;   // (DPP adds, ga_common.h)
  float sq = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k)
    if (lane + 64 * k < cv) {
#pragma unroll
      for (int e = 0; e < N; ++e) {
        const float d = f[k][e] - mean;
        sq += d * d;
      }
    }
  const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
  if (stats != nullptr && lane == 0) {
    stats[row * 2] = mean;
    stats[row * 2 + 1] = rstd;
  }
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int j = lane + 64 * k;
    if (j < cv) {
      const Vec<T> gm = reinterpret_cast<const Vec<T>*>(gamma)[j], bt = reinterpret_cast<const Vec<T>*>(beta)[j];
      Vec<T> o;
#pragma unroll
      for (int e = 0; e < N; ++e)
        o.v[e] = Traits<T>::from_f32((f[k][e] - mean) * rstd * Traits<T>::to_f32(gm.v[e]) + Traits<T>::to_f32(bt.v[e]));
      reinterpret_cast<Vec<T>*>(y + row * C)[j] = o;
    }
  }
}

// dx = rstd * (dyh - mean(dyh) - x_hat * mean(dyh * x_hat)) + g_res,   dyh = dy * gamma
template <typename T, int NV>
__global__ __launch_bounds__(kThreads) void add_ln_bwd_kernel(const T* __restrict__ x, const float* __restrict__ stats,
                                                              const T* __restrict__ gamma, const T* __restrict__ dy,
                                                              const T* __restrict__ gres, T* __restrict__ dx,
                                                              long long rows, int C) {
  constexpr int N = Vec<T>::N;
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * kRowsPerWg + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int cv = C / N;
  // every operand of the row is requested in one batch, from vector indices clamped into the row (the statistics last: what
  // consumes them then waits for everything, and everything is on its way).  Under `if (j < cv)` each vector's loads were
  // issued and waited for in a block of their own, behind the statistics, and the skip connection's gradient in the second
  // pass: five dependent round trips at 1280 channels.
  Vec<T> xv[NV], dv[NV], gv[NV], rv[NV];
  const T* rsrc = gres != nullptr ? gres : dy;   // stand-in, dropped below
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int j = min(lane + 64 * k, cv - 1);
    xv[k] = reinterpret_cast<const Vec<T>*>(x + row * C)[j];
    dv[k] = reinterpret_cast<const Vec<T>*>(dy + row * C)[j];
    gv[k] = reinterpret_cast<const Vec<T>*>(gamma)[j];
    rv[k] = reinterpret_cast<const Vec<T>*>(rsrc + row * C)[j];
  }
  const float2 mr = *reinterpret_cast<const float2*>(stats + row * 2);
  __builtin_amdgcn_sched_barrier(0);   // the loads above are issued before anything below is scheduled
  const float mean = mr.x, rstd = mr.y;
  float xh[NV][N], dh[NV][N];
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const bool in = lane + 64 * k < cv;
#pragma unroll
    for (int e = 0; e < N; ++e) {
      xh[k][e] = (Traits<T>::to_f32(xv[k].v[e]) - mean) * rstd;
      dh[k][e] = Traits<T>::to_f32(dv[k].v[e]) * Traits<T>::to_f32(gv[k].v[e]);
      s0 += in ? dh[k][e] : 0.f;
      s1 += in ? dh[k][e] * xh[k][e] : 0.f;
    }
  }
  const float m0 = wave_sum(s0) / (float)C, m1 = wave_sum(s1) / (float)C;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int j = lane + 64 * k;
    // computed by every lane, stored by those inside the row: with the arithmetic under the condition the compiler moved the
    // skip gradient's loads into that block — behind the row reduction, where they are a dependent round trip again
    Vec<T> o;
#pragma unroll
    for (int e = 0; e < N; ++e)
      o.v[e] = Traits<T>::from_f32(rstd * (dh[k][e] - m0 - xh[k][e] * m1) + (gres != nullptr ? Traits<T>::to_f32(rv[k].v[e]) : 0.f));
    if (j < cv) reinterpret_cast<Vec<T>*>(dx + row * C)[j] = o;
  }
}

// NV <= 8 vectors per lane: C <= 8 * 64 * (16 / sizeof(T)) = 4096 for 16-bit types, 2048 for f32

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <typename T, int NV>
int fwd_launch(const void* a, const void* x, const void* gamma, const void* beta, void* xnew, void* y, float* stats,
               long long rows, int C, float eps, hipStream_t s) {
  const unsigned grid = (unsigned)((rows + kRowsPerWg - 1) / kRowsPerWg);
  hipLaunchKernelGGL((add_ln_fwd_kernel<T, NV>), dim3(grid), dim3(kThreads), 0, s, (const T*)a, (const T*)x,
                     (const T*)gamma, (const T*)beta, (T*)xnew, (T*)y, stats, rows, C, eps);
  return check_launch();
}

template <typename T, int NV>
int bwd_launch(const void* x, const float* stats, const void* gamma, const void* dy, const void* gres, void* dx,
               long long rows, int C, hipStream_t s) {
  const unsigned grid = (unsigned)((rows + kRowsPerWg - 1) / kRowsPerWg);
  hipLaunchKernelGGL((add_ln_bwd_kernel<T, NV>), dim3(grid), dim3(kThreads), 0, s, (const T*)x, stats, (const T*)gamma,
                     (const T*)dy, (const T*)gres, (T*)dx, rows, C);
  return check_launch();
}

#define GA_NV_SWITCH(nv, CALL)  \
  switch (nv) {                 \
    case 1: return CALL(1);     \
    case 2: return CALL(2);     \
    case 3: return CALL(3);     \
    case 4: return CALL(4);     \
    case 5: return CALL(5);     \
    case 6: return CALL(6);     \
    case 7: return CALL(7);     \
    case 8: return CALL(8);     \
    default: return GA_ERR_SHAPE; \
  }

template <typename T>
int fwd_dtype(const void* a, const void* x, const void* gamma, const void* beta, void* xnew, void* y, float* stats,
              long long rows, int C, float eps, hipStream_t s) {
  constexpr int N = Vec<T>::N;
  if (C % N) return GA_ERR_SHAPE;
  const int nv = (C / N + 63) / 64;
#define GA_CALL(NV) fwd_launch<T, NV>(a, x, gamma, beta, xnew, y, stats, rows, C, eps, s)
  GA_NV_SWITCH(nv, GA_CALL)
#undef GA_CALL
}

template <typename T>
int bwd_dtype(const void* x, const float* stats, const void* gamma, const void* dy, const void* gres, void* dx,
              long long rows, int C, hipStream_t s) {
  constexpr int N = Vec<T>::N;
  if (C % N) return GA_ERR_SHAPE;
  const int nv = (C / N + 63) / 64;
#define GA_CALL(NV) bwd_launch<T, NV>(x, stats, gamma, dy, gres, dx, rows, C, s)
  GA_NV_SWITCH(nv, GA_CALL)
#undef GA_CALL
}

}  // namespace

extern "C" int ga_add_layer_norm_fwd(const void* a, const void* x, const void* gamma, const void* beta, void* xnew,
                                     void* y, float* stats, int64_t rows, int C, float eps, int dtype,
                                     ga_stream_t stream) {
  if (!x || !gamma || !beta || !y || (a && !xnew)) return GA_ERR_NULL;
  if (rows < 1 || C < 1) return GA_ERR_SHAPE;
  if (!aligned16(x) || !aligned16(gamma) || !aligned16(beta) || !aligned16(y) || (a && (!aligned16(a) || !aligned16(xnew))))
    return GA_ERR_ALIGN;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case GA_F16: return fwd_dtype<_Float16>(a, x, gamma, beta, xnew, y, stats, rows, C, eps, s);
    case GA_BF16: return fwd_dtype<bf16_t>(a, x, gamma, beta, xnew, y, stats, rows, C, eps, s);
    case GA_F32: return fwd_dtype<float>(a, x, gamma, beta, xnew, y, stats, rows, C, eps, s);
    default: return GA_ERR_DTYPE;
  }
}

extern "C" int ga_add_layer_norm_bwd(const void* x, const float* stats, const void* gamma, const void* dy,
                                     const void* g_res, void* dx, int64_t rows, int C, int dtype, ga_stream_t stream) {
  if (!x || !stats || !gamma || !dy || !dx) return GA_ERR_NULL;
  if (rows < 1 || C < 1) return GA_ERR_SHAPE;
  if (!aligned16(x) || !aligned16(gamma) || !aligned16(dy) || !aligned16(dx) || (g_res && !aligned16(g_res)))
    return GA_ERR_ALIGN;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case GA_F16: return bwd_dtype<_Float16>(x, stats, gamma, dy, g_res, dx, rows, C, s);
    case GA_BF16: return bwd_dtype<bf16_t>(x, stats, gamma, dy, g_res, dx, rows, C, s);
    case GA_F32: return bwd_dtype<float>(x, stats, gamma, dy, g_res, dx, rows, C, s);
    default: return GA_ERR_DTYPE;
  }
}
