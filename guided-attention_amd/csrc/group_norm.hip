// GroupNorm (+ fused SiLU) for channels-last activations, forward and backward-to-input.
//
// Host-side UNet helper (reference: the diffusers ResnetBlock2D / Transformer2DModel GroupNorm + SiLU pairs
// inside the UNet forward, pipeline_guided_attention.py:583-743).  PyTorch's GroupNorm runs three kernels plus a
// separate SiLU, and returns an NCHW tensor even for a channels-last input, which makes MIOpen transpose
// around every following convolution.  This pair of kernels keeps NHWC end to end:
//   stats : one workgroup per (image, pixel block): coalesced row reads (4 B = 2 channels per lane; every
//           channel pair lies in one group because C/G is even), per-thread sums, fixed-order LDS reduction
//           to per-group partial (sum, sum of squares)                       -> workspace [B][NB][G][2], NB <= 64
//   apply : every workgroup first folds the NB partials of all G groups itself (8 lanes per group, fixed order: the
//           same (mean, rstd) bits in every workgroup) — there is no separate finalize launch — then
//           y = silu((x - mean) * rstd * gamma + beta); workgroup 0 also leaves (mean, rstd) for the backward
// Backward recomputes the normalised value and the SiLU derivative, reduces (sum dyhat, sum dyhat*yhat) the same
// way and writes dx; gamma/beta gradients are not produced (the UNet weights are frozen on this path).
// HBM-bound: forward moves 2 reads + 1 write of the tensor (the second read hits L2/MALL at these sizes).
#include "attn_common.h"

using namespace ga;

namespace {

constexpr int kThreads = 256;
constexpr int kMaxNPT = 5;  // channel pairs per thread: C <= 2560
constexpr int kU = 8;       // pixels per lane whose loads are issued together (16 in the apply kernels)

// W consecutive channels handled by one lane per pixel: 2 (a 4-byte access for 16-bit types) when the group size
// is even, which is every SD layer; 1 for odd group sizes (reduced-width test models).
template <typename T, int W>
struct Item {
  T v[W];
};

// A channel PAIR as one access (even channel index: 4-byte aligned for the 16-bit types).  Read through Item<T, 2>, whose
// alignment is that of T, the pair becomes two 2-byte loads the scheduler places apart — and the later one decides where the
// wait for "the pair" sits.
template <typename T>
__device__ __forceinline__ Item<T, 2> load_pair(const T* p) {
  if constexpr (sizeof(T) == 2) return __builtin_bit_cast(Item<T, 2>, *reinterpret_cast<const unsigned*>(p));
  else return __builtin_bit_cast(Item<T, 2>, *reinterpret_cast<const unsigned long long*>(p));
}

// v_rcp_f32 (1 ulp), not an IEEE division (ten instructions per element in every GroupNorm + SiLU pass)
__device__ __forceinline__ float sigmoidf_(float z) { return __builtin_amdgcn_rcpf(1.0f + __expf(-z)); }

// optional per-(image, channel) bias added to x on load (the ResnetBlock's time-embedding term): normalising
// x + bias[c] without a separate broadcast-add pass over the tensor
template <typename T, int W>
__device__ __forceinline__ void load_chan_bias(const T* __restrict__ cbias, int b, int CP, int cp, float (&cb)[W]) {
#pragma unroll
  for (int j = 0; j < W; ++j) cb[j] = 0.f;
  if (cbias != nullptr) {
    const Item<T, W> it = reinterpret_cast<const Item<T, W>*>(cbias)[(size_t)b * CP + cp];
#pragma unroll
    for (int j = 0; j < W; ++j) cb[j] = Traits<T>::to_f32(it.v[j]);
  }
}

// sums over this workgroup's pixels of (v0, v1) per channel pair, folded to groups in fixed order
template <int NPT>
__device__ __forceinline__ void fold_to_groups(const float (&s0)[NPT], const float (&s1)[NPT], int CP, int cpg, int G,
                                               float* lds, float* out) {
#pragma unroll
  for (int k = 0; k < NPT; ++k) {
    const int cp = threadIdx.x + k * kThreads;
    if (cp < CP) {
      lds[2 * cp] = s0[k];
      lds[2 * cp + 1] = s1[k];
    }
  }
  __syncthreads();
  for (int g = threadIdx.x; g < G; g += kThreads) {
    float a = 0.f, b = 0.f;
    for (int cp = g * cpg; cp < (g + 1) * cpg; ++cp) {
      a += lds[2 * cp];
      b += lds[2 * cp + 1];
    }
    out[2 * g] = a;
    out[2 * g + 1] = b;
  }
}

template <typename T, int NPT, int W>
__global__ __launch_bounds__(kThreads) void gn_stats_kernel(const T* __restrict__ x, const T* __restrict__ cbias,
                                                            float* __restrict__ partial, int HW, int C, int G, int PB) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int b = blockIdx.y, nb = blockIdx.x, CP = C / W, cpg = (C / G) / W;
  const int p0 = nb * PB, p1 = min(HW, p0 + PB);
  float s0[NPT], s1[NPT];
#pragma unroll
  for (int k = 0; k < NPT; ++k) s0[k] = s1[k] = 0.f;
  const Item<T, W>* xb = reinterpret_cast<const Item<T, W>*>(x) + (size_t)b * HW * CP;
#pragma unroll
  for (int k = 0; k < NPT; ++k) {
    const int cp = threadIdx.x + k * kThreads;
    if (cp >= CP) continue;
    float cb[W];
    load_chan_bias<T, W>(cbias, b, CP, cp, cb);
    for (int p = p0; p < p1; p += kU) {  // kU independent loads in flight per lane, then the fixed-order adds
      Item<T, W> v[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u)
        if (p + u < p1) v[u] = xb[(size_t)(p + u) * CP + cp];
#pragma unroll
      for (int u = 0; u < kU; ++u)
        if (p + u < p1) {
#pragma unroll
          for (int j = 0; j < W; ++j) {
            const float a = Traits<T>::to_f32(v[u].v[j]) + cb[j];
            s0[k] += a;
            s1[k] += a * a;
          }
        }
    }
  }
  fold_to_groups<NPT>(s0, s1, CP, cpg, G, lds, partial + ((size_t)b * gridDim.x + nb) * G * 2);
}

// Fold the NB per-block partials of image b into per-group results in LDS (`res`, float2 per group), fixed order:
// 8 lanes per group (lane part sums partials part, part + 8, ...), then a 3-step butterfly inside the 8 lanes.
//   FWD : (mean, rstd)            BWD : (mean of dyhat, mean of dyhat * yhat)
// Needs G <= kThreads / 8 = 32 groups per pass (more: several passes).  Ends with a barrier.  keep != nullptr: this
// workgroup also stores the results (the forward's (mean, rstd) that the backward reads).
constexpr int kMaxStatsNB = 128;

struct NoHook {
  __device__ __forceinline__ void operator()() const {}
};
// `after_loads` runs once, when the (first pass of the) fold's own loads are out: the callers request their per-channel
// constants there — cold weights, needed only behind the fold's barrier, so the fold runs under their latency.
template <bool FWD, typename Hook = NoHook>
__device__ __forceinline__ void fold_partials(const float* __restrict__ partial, int b, int NB, int G, float inv_n,
                                              float eps, float2* res, float* keep, Hook after_loads = Hook()) {
  const float2* pp = reinterpret_cast<const float2*>(partial) + (size_t)b * NB * G;
  for (int g0 = 0; g0 < G; g0 += kThreads / 8) {
    const int g = g0 + (threadIdx.x >> 3), part = threadIdx.x & 7;
    float2 v[kMaxStatsNB / 8];
#pragma unroll
    for (int i = 0; i < kMaxStatsNB / 8; ++i) {
      // clamped, not predicated: sixteen loads back to back (under a condition each sat in its own block and the first was
      // waited for alone); what lies past the edge is dropped by the selects below
      const int nb = part + 8 * i;
      v[i] = pp[(size_t)min(nb, NB - 1) * G + min(g, G - 1)];
    }
    if (g0 == 0) after_loads();
    float sa = 0.f, sc = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxStatsNB / 8; ++i) {
      const bool in = g < G && part + 8 * i < NB;
      sa += in ? v[i].x : 0.f;
      sc += in ? v[i].y : 0.f;
    }
    sa = group_sum<8>(sa);   // (DPP adds: ga_common.h)
    sc = group_sum<8>(sc);
    if (part == 0 && g < G) {
      float o0, o1;
      if (FWD) {
        const double mean = (double)sa * inv_n;
        const double var = fmax((double)sc * inv_n - mean * mean, 0.0);
        o0 = (float)mean;
        o1 = rsqrtf((float)var + eps);
      } else {
        o0 = sa * inv_n;
        o1 = sc * inv_n;
      }
      res[g] = float2{o0, o1};
      if (keep != nullptr) {
        keep[((size_t)b * G + g) * 2] = o0;
        keep[((size_t)b * G + g) * 2 + 1] = o1;
      }
    }
  }
  __syncthreads();
}

template <typename T, bool ACT, int NPT, int W>
__global__ __launch_bounds__(kThreads) void gn_apply_kernel(const T* __restrict__ x, const T* __restrict__ cbias,
                                                            const T* __restrict__ gamma,
                                                            const T* __restrict__ beta, T* __restrict__ y,
                                                            const float* __restrict__ partial, int NB, float inv_n,
                                                            float eps, float* __restrict__ stats, int HW, int C, int G,
                                                            int PB) {
  __shared__ float2 res[64];
  const int b = blockIdx.y, CP = C / W, cpg = (C / G) / W;
  fold_partials<true>(partial, b, NB, G, inv_n, eps, res, blockIdx.x == 0 ? stats : nullptr);
  const float* mu_rs = reinterpret_cast<const float*>(res);
  float sc[NPT][W], sh[NPT][W];
#pragma unroll
  for (int k = 0; k < NPT; ++k) {
    const int cp = threadIdx.x + k * kThreads;
    if (cp < CP) {
      const int g = cp / cpg;
      const Item<T, W> ga_ = reinterpret_cast<const Item<T, W>*>(gamma)[cp];
      const Item<T, W> be = reinterpret_cast<const Item<T, W>*>(beta)[cp];
      float cb[W];
      load_chan_bias<T, W>(cbias, b, CP, cp, cb);
#pragma unroll
      for (int j = 0; j < W; ++j) {
        sc[k][j] = Traits<T>::to_f32(ga_.v[j]) * mu_rs[2 * g + 1];
        sh[k][j] = Traits<T>::to_f32(be.v[j]) - (mu_rs[2 * g] - cb[j]) * sc[k][j];  // (x + cb - mean) * scale + beta
      }
    }
  }
  const int p0 = blockIdx.x * PB, p1 = min(HW, p0 + PB);
  const Item<T, W>* xb = reinterpret_cast<const Item<T, W>*>(x) + (size_t)b * HW * CP;
  Item<T, W>* yb = reinterpret_cast<Item<T, W>*>(y) + (size_t)b * HW * CP;
#pragma unroll
  for (int k = 0; k < NPT; ++k) {
    const int cp = threadIdx.x + k * kThreads;
    if (cp >= CP) continue;
    for (int p = p0; p < p1; p += kU) {
      Item<T, W> v[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u)
        if (p + u < p1) v[u] = xb[(size_t)(p + u) * CP + cp];
#pragma unroll
      for (int u = 0; u < kU; ++u)
        if (p + u < p1) {
          Item<T, W> o;
#pragma unroll
          for (int j = 0; j < W; ++j) {
            float z = Traits<T>::to_f32(v[u].v[j]) * sc[k][j] + sh[k][j];
            if (ACT) z *= sigmoidf_(z);
            o.v[j] = Traits<T>::from_f32(z);
          }
          yb[(size_t)(p + u) * CP + cp] = o;
        }
    }
  }
}

// dyhat = dL/d(normalised value) for one element: through the affine and (optionally) SiLU
template <bool ACT>
__device__ __forceinline__ float dyhat_of(float yhat, float dy, float gam, float bet) {
  float dz = dy;
  if (ACT) {
    const float z = gam * yhat + bet;
    const float s = sigmoidf_(z);
    dz = dy * s * (1.0f + z * (1.0f - s));
  }
  return dz * gam;
}

template <typename T, bool ACT, int NPT, int W>
__global__ __launch_bounds__(kThreads) void gn_bwd_stats_kernel(const T* __restrict__ x, const T* __restrict__ cbias,
                                                                const T* __restrict__ dy,
                                                                const T* __restrict__ gamma,
                                                                const T* __restrict__ beta,
                                                                const float* __restrict__ stats,
                                                                float* __restrict__ partial, int HW, int C, int G,
                                                                int PB) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int b = blockIdx.y, nb = blockIdx.x, CP = C / W, cpg = (C / G) / W;
  const int p0 = nb * PB, p1 = min(HW, p0 + PB);
  float s0[NPT], s1[NPT], mu[NPT], rs[NPT], g0[NPT][W], b0[NPT][W], cbk[NPT][W];
#pragma unroll
  for (int k = 0; k < NPT; ++k) {
    s0[k] = s1[k] = 0.f;
    const int cp = threadIdx.x + k * kThreads;
    if (cp < CP) {
      const int g = cp / cpg;
      mu[k] = stats[((size_t)b * G + g) * 2];
      rs[k] = stats[((size_t)b * G + g) * 2 + 1];
      const Item<T, W> ga_ = reinterpret_cast<const Item<T, W>*>(gamma)[cp];
      const Item<T, W> be = reinterpret_cast<const Item<T, W>*>(beta)[cp];
      load_chan_bias<T, W>(cbias, b, CP, cp, cbk[k]);
#pragma unroll
      for (int j = 0; j < W; ++j) {
        g0[k][j] = Traits<T>::to_f32(ga_.v[j]);
        b0[k][j] = Traits<T>::to_f32(be.v[j]);
      }
    }
  }
  const Item<T, W>* xb = reinterpret_cast<const Item<T, W>*>(x) + (size_t)b * HW * CP;
  const Item<T, W>* db = reinterpret_cast<const Item<T, W>*>(dy) + (size_t)b * HW * CP;
#pragma unroll
  for (int k = 0; k < NPT; ++k) {
    const int cp = threadIdx.x + k * kThreads;
    if (cp >= CP) continue;
    for (int p = p0; p < p1; p += kU) {
      Item<T, W> v[kU], d[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u)
        if (p + u < p1) {
          v[u] = xb[(size_t)(p + u) * CP + cp];
          d[u] = db[(size_t)(p + u) * CP + cp];
        }
#pragma unroll
      for (int u = 0; u < kU; ++u)
        if (p + u < p1) {
#pragma unroll
          for (int j = 0; j < W; ++j) {
            const float yh = (Traits<T>::to_f32(v[u].v[j]) + cbk[k][j] - mu[k]) * rs[k];
            const float dh = dyhat_of<ACT>(yh, Traits<T>::to_f32(d[u].v[j]), g0[k][j], b0[k][j]);
            s0[k] += dh;
            s1[k] += dh * yh;
          }
        }
    }
  }
  fold_to_groups<NPT>(s0, s1, CP, cpg, G, lds, partial + ((size_t)b * gridDim.x + nb) * G * 2);
}

template <typename T, bool ACT, int NPT, int W>
__global__ __launch_bounds__(kThreads) void gn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ cbias,
                                                                const T* __restrict__ dy,
                                                                const T* __restrict__ gamma,
                                                                const T* __restrict__ beta,
                                                                const float* __restrict__ stats,
                                                                const float* __restrict__ partial, int NB, float inv_n,
                                                                const T* __restrict__ gres, T* __restrict__ dx, int HW,
                                                                int C, int G, int PB) {
  __shared__ float2 res[64];
  const int b = blockIdx.y, CP = C / W, cpg = (C / G) / W;
  fold_partials<false>(partial, b, NB, G, inv_n, 0.f, res, nullptr);
  float mu[NPT], rs[NPT], a1[NPT], a2[NPT], g0[NPT][W], b0[NPT][W], cbk[NPT][W];
#pragma unroll
  for (int k = 0; k < NPT; ++k) {
    const int cp = threadIdx.x + k * kThreads;
    if (cp < CP) {
      const int g = cp / cpg;
      mu[k] = stats[((size_t)b * G + g) * 2];
      rs[k] = stats[((size_t)b * G + g) * 2 + 1];
      a1[k] = res[g].x;
      a2[k] = res[g].y;
      const Item<T, W> ga_ = reinterpret_cast<const Item<T, W>*>(gamma)[cp];
      const Item<T, W> be = reinterpret_cast<const Item<T, W>*>(beta)[cp];
      load_chan_bias<T, W>(cbias, b, CP, cp, cbk[k]);
#pragma unroll
      for (int j = 0; j < W; ++j) {
        g0[k][j] = Traits<T>::to_f32(ga_.v[j]);
        b0[k][j] = Traits<T>::to_f32(be.v[j]);
      }
    }
  }
  const int p0 = blockIdx.x * PB, p1 = min(HW, p0 + PB);
  const Item<T, W>* xb = reinterpret_cast<const Item<T, W>*>(x) + (size_t)b * HW * CP;
  const Item<T, W>* db = reinterpret_cast<const Item<T, W>*>(dy) + (size_t)b * HW * CP;
  Item<T, W>* ob = reinterpret_cast<Item<T, W>*>(dx) + (size_t)b * HW * CP;
  const Item<T, W>* rb = reinterpret_cast<const Item<T, W>*>(gres) + (size_t)b * HW * CP;   // gradient of x's other consumer
#pragma unroll
  for (int k = 0; k < NPT; ++k) {
    const int cp = threadIdx.x + k * kThreads;
    if (cp >= CP) continue;
    for (int p = p0; p < p1; p += kU) {
      Item<T, W> v[kU], d[kU], gr[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u)
        if (p + u < p1) {
          v[u] = xb[(size_t)(p + u) * CP + cp];
          d[u] = db[(size_t)(p + u) * CP + cp];
          if (gres != nullptr) gr[u] = rb[(size_t)(p + u) * CP + cp];
        }
#pragma unroll
      for (int u = 0; u < kU; ++u)
        if (p + u < p1) {
          Item<T, W> o;
#pragma unroll
          for (int j = 0; j < W; ++j) {
            const float yh = (Traits<T>::to_f32(v[u].v[j]) + cbk[k][j] - mu[k]) * rs[k];
            const float dh = dyhat_of<ACT>(yh, Traits<T>::to_f32(d[u].v[j]), g0[k][j], b0[k][j]);
            const float add = gres != nullptr ? Traits<T>::to_f32(gr[u].v[j]) : 0.f;
            o.v[j] = Traits<T>::from_f32(rs[k] * (dh - a1[k] - yh * a2[k]) + add);
          }
          ob[(size_t)(p + u) * CP + cp] = o;
        }
    }
  }
}

// ---- wide path for the large levels (16-bit types, C % 8 == 0, 8 <= C/G, C <= 2048): 16-byte vectors that may
// straddle a group boundary, and ALL 256 threads busy — thread t owns vector t % (C/8) of the pixels
// p0 + t / (C/8), + RP, ... (C = 320: 40 vectors x 6 pixel rows; the 4/8-byte path above leaves 176 of 256 lanes
// idle there).  A vector's 8 channels lie in at most two groups: gA (the first `split` channels) and gA + 1.
struct WideMap {
  int VP, RP, vec, pr, gA, split;
  bool active;
  __device__ __forceinline__ WideMap(int C, int G) {
    const int cg = C / G;
    VP = C >> 3;
    RP = kThreads / VP;
    vec = threadIdx.x % VP;
    pr = threadIdx.x / VP;
    active = pr < RP;
    gA = (vec * 8) / cg;
    split = min(8, (gA + 1) * cg - vec * 8);
  }
};

template <typename T>
struct alignas(16) Vec8 {
  T v[8];
};

// per-thread (sumA0, sumA1, sumB0, sumB1) -> per-group partial sums of this pixel block, fixed order
__device__ __forceinline__ void wide_fold(const WideMap& m, const float (&c0)[8], const float (&c1)[8], int C, int G,
                                          float4* lds4, float* out) {
  float4 r = {0.f, 0.f, 0.f, 0.f};
  if (m.active) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j < m.split) {
        r.x += c0[j];
        r.y += c1[j];
      } else {
        r.z += c0[j];
        r.w += c1[j];
      }
    }
  }
  lds4[threadIdx.x] = r;
  __syncthreads();
  const int cg = C / G;
  for (int g = threadIdx.x; g < G; g += kThreads) {
    const int vlo = (g * cg) >> 3, vhi = ((g + 1) * cg - 1) >> 3;
    float a = 0.f, b = 0.f;
    for (int v = vlo; v <= vhi; ++v) {
      const bool first = (v * 8) / cg == g;  // g is this vector's gA, otherwise its gA + 1
      for (int pr = 0; pr < m.RP; ++pr) {
        const float4 e = lds4[pr * m.VP + v];
        a += first ? e.x : e.z;
        b += first ? e.y : e.w;
      }
    }
    out[2 * g] = a;
    out[2 * g + 1] = b;
  }
}

constexpr int kUS = 16;  // stats kernels: loads in flight per lane (a workgroup's whole pixel block in one round trip)

template <typename T>
__global__ __launch_bounds__(kThreads) void gn_wide_stats_kernel(const T* __restrict__ x, const T* __restrict__ cbias,
                                                                 float* __restrict__ partial, int HW, int C, int G,
                                                                 int PB) {
  __shared__ float4 lds4[kThreads];
  const WideMap m(C, G);
  const int b = blockIdx.y, nb = blockIdx.x, p0 = nb * PB, p1 = min(HW, p0 + PB);
  float c0[8], c1[8], cb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) c0[j] = c1[j] = cb[j] = 0.f;
  if (m.active) {
    const Vec8<T>* xb = reinterpret_cast<const Vec8<T>*>(x + (size_t)b * HW * C) + m.vec;
    // the first (normally the only) batch of activation loads goes out before the channel bias is touched: converted where
    // it was loaded, the bias was a round trip of its own in front of them
    Vec8<T> v[kUS];
    int p = p0 + m.pr;
#pragma unroll
    for (int u = 0; u < kUS; ++u)
      if (p + u * m.RP < p1) v[u] = xb[(size_t)(p + u * m.RP) * m.VP];
    if (cbias != nullptr) {
      const Vec8<T> bv = reinterpret_cast<const Vec8<T>*>(cbias + (size_t)b * C)[m.vec];
#pragma unroll
      for (int j = 0; j < 8; ++j) cb[j] = Traits<T>::to_f32(bv.v[j]);
    }
    for (; p < p1; p += kUS * m.RP) {
#pragma unroll
      for (int u = 0; u < kUS; ++u)
        if (p + u * m.RP < p1) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float a = Traits<T>::to_f32(v[u].v[j]) + cb[j];
            c0[j] += a;
            c1[j] += a * a;
          }
        }
      const int pn = p + kUS * m.RP;
#pragma unroll
      for (int u = 0; u < kUS; ++u)
        if (pn + u * m.RP < p1) v[u] = xb[(size_t)(pn + u * m.RP) * m.VP];
    }
  }
  wide_fold(m, c0, c1, C, G, lds4, partial + ((size_t)b * gridDim.x + nb) * G * 2);
}

// The statistics pass fused into the PRODUCER of a norm's input where that producer is a concatenation (diffusers 0.12.1
// UpBlock2D / CrossAttnUpBlock2D: torch.cat([hidden_states, res_hidden_states], dim=1) in front of resnet.norm1): the statistics
// kernel above, reading its pixels from the two sources and writing the concatenated tensor on the way — ga_cat_channels and
// gn_wide_stats_kernel as ONE launch.  Same pixel blocks, same fold order: the partial sums are bit-identical to those
// gn_wide_stats_kernel takes from the concatenated tensor.
template <typename T>
__global__ __launch_bounds__(kThreads) void gn_wide_cat_stats_kernel(const T* __restrict__ a, const T* __restrict__ b2,
                                                                     T* __restrict__ out, float* __restrict__ partial, int HW,
                                                                     int C1, int C2, int G, int PB) {
  __shared__ float4 lds4[kThreads];
  const int C = C1 + C2;
  const WideMap m(C, G);
  const int b = blockIdx.y, nb = blockIdx.x, p0 = nb * PB, p1 = min(HW, p0 + PB);
  float c0[8], c1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) c0[j] = c1[j] = 0.f;
  if (m.active) {
    const int V1 = C1 >> 3, V2 = C2 >> 3;
    const bool first = m.vec < V1;
    const Vec8<T>* src = first ? reinterpret_cast<const Vec8<T>*>(a + (size_t)b * HW * C1) + m.vec
                               : reinterpret_cast<const Vec8<T>*>(b2 + (size_t)b * HW * C2) + (m.vec - V1);
    const int sv = first ? V1 : V2;
    Vec8<T>* ob = reinterpret_cast<Vec8<T>*>(out + (size_t)b * HW * C) + m.vec;
    Vec8<T> v[kUS];
    int p = p0 + m.pr;
#pragma unroll
    for (int u = 0; u < kUS; ++u)
      if (p + u * m.RP < p1) v[u] = src[(size_t)(p + u * m.RP) * sv];
    for (; p < p1; p += kUS * m.RP) {
#pragma unroll
      for (int u = 0; u < kUS; ++u)
        if (p + u * m.RP < p1) {
          ob[(size_t)(p + u * m.RP) * m.VP] = v[u];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float x = Traits<T>::to_f32(v[u].v[j]);
            c0[j] += x;
            c1[j] += x * x;
          }
        }
      const int pn = p + kUS * m.RP;
#pragma unroll
      for (int u = 0; u < kUS; ++u)
        if (pn + u * m.RP < p1) v[u] = src[(size_t)(pn + u * m.RP) * sv];
    }
  }
  wide_fold(m, c0, c1, C, G, lds4, partial + ((size_t)b * gridDim.x + nb) * G * 2);
}

template <typename T, bool ACT>
__global__ __launch_bounds__(kThreads) void gn_wide_apply_kernel(const T* __restrict__ x, const T* __restrict__ cbias,
                                                                 const T* __restrict__ gamma,
                                                                 const T* __restrict__ beta, T* __restrict__ y,
                                                                 const float* __restrict__ partial, int NB, float inv_n,
                                                                 float eps, float* __restrict__ stats, int HW, int C,
                                                                 int G, int PB) {
  __shared__ float2 res[64];
  const WideMap m(C, G);
  const int b = blockIdx.y, p0 = blockIdx.x * PB, p1 = min(HW, p0 + PB);
  const Vec8<T>* xb = reinterpret_cast<const Vec8<T>*>(x + (size_t)b * HW * C) + m.vec;
  Vec8<T>* yb = reinterpret_cast<Vec8<T>*>(y + (size_t)b * HW * C) + m.vec;
  // the first batch of activation loads is issued BEFORE the statistics are folded: the fold is its own dependent
  // round trip to L2 (partials -> shuffle -> LDS -> barrier) and nothing in it needs x
  Vec8<T> v[kU];
  int p = p0 + m.pr;
  if (m.active) {
#pragma unroll
    for (int u = 0; u < kU; ++u)
      if (p + u * m.RP < p1) v[u] = xb[(size_t)(p + u * m.RP) * m.VP];
  }
  // gamma / beta (cold weights) and the channel bias are requested here too: behind the fold they were one more dependent
  // round trip to memory after its barrier
  // by every thread (m.vec is a valid vector index for the idle ones too): under `if (m.active)` the three values were
  // merged behind waits before the fold's own loads could go out
  Vec8<T> gm, bt, bv;
  fold_partials<true>(partial, b, NB, G, inv_n, eps, res, blockIdx.x == 0 ? stats : nullptr, [&]() {
    gm = reinterpret_cast<const Vec8<T>*>(gamma)[m.vec];
    bt = reinterpret_cast<const Vec8<T>*>(beta)[m.vec];
    bv = reinterpret_cast<const Vec8<T>*>(cbias != nullptr ? cbias + (size_t)b * C : x + (size_t)b * HW * C)[m.vec];
  });
  if (!m.active) return;
  const float* mr = reinterpret_cast<const float*>(res);
  float sc[8], sh[8];
  {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int g = j < m.split ? m.gA : m.gA + 1;
      const float cbj = cbias != nullptr ? Traits<T>::to_f32(bv.v[j]) : 0.f;
      sc[j] = Traits<T>::to_f32(gm.v[j]) * mr[2 * g + 1];
      sh[j] = Traits<T>::to_f32(bt.v[j]) - (mr[2 * g] - cbj) * sc[j];  // (x + cb - mean) * scale + beta
    }
  }
  for (; p < p1; p += kU * m.RP) {
#pragma unroll
    for (int u = 0; u < kU; ++u)
      if (p + u * m.RP < p1) {
        Vec8<T> o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float z = Traits<T>::to_f32(v[u].v[j]) * sc[j] + sh[j];
          if (ACT) z *= sigmoidf_(z);
          o.v[j] = Traits<T>::from_f32(z);
        }
        yb[(size_t)(p + u * m.RP) * m.VP] = o;
      }
    const int pn = p + kU * m.RP;   // next batch (none with the launch geometry of WideGeom: one batch per thread)
#pragma unroll
    for (int u = 0; u < kU; ++u)
      if (pn + u * m.RP < p1) v[u] = xb[(size_t)(pn + u * m.RP) * m.VP];
  }
}

// the per-channel constants the backward kernels share: `fetch` only requests them (gamma / beta are cold weights), `finish`
// converts — between the two the kernels issue their activation loads, so that everything is one round trip
template <typename T>
struct WideBwdConst {
  Vec8<T> g8, b8, c8;
  float2 st[2];
  float mu[8], rs[8], gm[8], bt[8], cb[8];
  // x_img: the image's activations — what the bias load reads when there is no channel bias (dropped by `finish`)
  __device__ __forceinline__ void fetch(const WideMap& m, const T* cbias, const T* gamma, const T* beta, const float* stats,
                                        const T* x_img, int b, int C, int G) {
    g8 = reinterpret_cast<const Vec8<T>*>(gamma)[m.vec];
    b8 = reinterpret_cast<const Vec8<T>*>(beta)[m.vec];
    c8 = reinterpret_cast<const Vec8<T>*>(cbias != nullptr ? cbias + (size_t)b * C : x_img)[m.vec];   // no branch around the load
    const float2* s2 = reinterpret_cast<const float2*>(stats) + (size_t)b * G;
    st[0] = s2[m.gA];
    st[1] = s2[min(m.gA + 1, G - 1)];
  }
  __device__ __forceinline__ void finish(const WideMap& m, bool has_cb) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float2 s = j < m.split ? st[0] : st[1];
      mu[j] = s.x;
      rs[j] = s.y;
      gm[j] = Traits<T>::to_f32(g8.v[j]);
      bt[j] = Traits<T>::to_f32(b8.v[j]);
      cb[j] = has_cb ? Traits<T>::to_f32(c8.v[j]) : 0.f;
    }
  }
};

template <typename T, bool ACT>
__global__ __launch_bounds__(kThreads) void gn_wide_bwd_stats_kernel(const T* __restrict__ x, const T* __restrict__ cbias,
                                                                     const T* __restrict__ dy,
                                                                     const T* __restrict__ gamma,
                                                                     const T* __restrict__ beta,
                                                                     const float* __restrict__ stats,
                                                                     float* __restrict__ partial, int HW, int C, int G,
                                                                     int PB) {
  __shared__ float4 lds4[kThreads];
  const WideMap m(C, G);
  const int b = blockIdx.y, nb = blockIdx.x, p0 = nb * PB, p1 = min(HW, p0 + PB);
  float c0[8], c1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) c0[j] = c1[j] = 0.f;
  if (m.active) {
    WideBwdConst<T> k;
    k.fetch(m, cbias, gamma, beta, stats, x + (size_t)b * HW * C, b, C, G);
    const Vec8<T>* xb = reinterpret_cast<const Vec8<T>*>(x + (size_t)b * HW * C) + m.vec;
    const Vec8<T>* db = reinterpret_cast<const Vec8<T>*>(dy + (size_t)b * HW * C) + m.vec;
    Vec8<T> v[kU], d[kU];
    int p = p0 + m.pr;
#pragma unroll
    for (int u = 0; u < kU; ++u)
      if (p + u * m.RP < p1) {
        v[u] = xb[(size_t)(p + u * m.RP) * m.VP];
        d[u] = db[(size_t)(p + u * m.RP) * m.VP];
      }
    k.finish(m, cbias != nullptr);
    for (; p < p1; p += kU * m.RP) {
#pragma unroll
      for (int u = 0; u < kU; ++u)
        if (p + u * m.RP < p1) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float yh = (Traits<T>::to_f32(v[u].v[j]) + k.cb[j] - k.mu[j]) * k.rs[j];
            const float dh = dyhat_of<ACT>(yh, Traits<T>::to_f32(d[u].v[j]), k.gm[j], k.bt[j]);
            c0[j] += dh;
            c1[j] += dh * yh;
          }
        }
      const int pn = p + kU * m.RP;
#pragma unroll
      for (int u = 0; u < kU; ++u)
        if (pn + u * m.RP < p1) {
          v[u] = xb[(size_t)(pn + u * m.RP) * m.VP];
          d[u] = db[(size_t)(pn + u * m.RP) * m.VP];
        }
    }
  }
  wide_fold(m, c0, c1, C, G, lds4, partial + ((size_t)b * gridDim.x + nb) * G * 2);
}

template <typename T, bool ACT>
__global__ __launch_bounds__(kThreads) void gn_wide_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ cbias,
                                                                     const T* __restrict__ dy,
                                                                     const T* __restrict__ gamma,
                                                                     const T* __restrict__ beta,
                                                                     const float* __restrict__ stats,
                                                                     const float* __restrict__ partial, int NB,
                                                                     float inv_n, const T* __restrict__ gres,
                                                                     T* __restrict__ dx, int HW, int C, int G, int PB) {
  __shared__ float2 res[64];
  const WideMap m(C, G);
  const int b = blockIdx.y, p0 = blockIdx.x * PB, p1 = min(HW, p0 + PB);
  const Vec8<T>* xb = reinterpret_cast<const Vec8<T>*>(x + (size_t)b * HW * C) + m.vec;
  const Vec8<T>* db = reinterpret_cast<const Vec8<T>*>(dy + (size_t)b * HW * C) + m.vec;
  Vec8<T>* ob = reinterpret_cast<Vec8<T>*>(dx + (size_t)b * HW * C) + m.vec;
  const Vec8<T>* rb = reinterpret_cast<const Vec8<T>*>(gres + (size_t)b * HW * C) + m.vec;   // gradient of x's other consumer
  Vec8<T> v[kU], d[kU], gr[kU];   // first batch in flight under the fold of the partial sums (as in the forward)
  int p = p0 + m.pr;
  if (m.active) {
#pragma unroll
    for (int u = 0; u < kU; ++u)
      if (p + u * m.RP < p1) {
        v[u] = xb[(size_t)(p + u * m.RP) * m.VP];
        d[u] = db[(size_t)(p + u * m.RP) * m.VP];
        if (gres != nullptr) gr[u] = rb[(size_t)(p + u * m.RP) * m.VP];
      }
  }
  WideBwdConst<T> k;   // fetched by every thread (no merge behind a wait), behind the fold's own loads
  fold_partials<false>(partial, b, NB, G, inv_n, 0.f, res, nullptr,
                       [&]() { k.fetch(m, cbias, gamma, beta, stats, x + (size_t)b * HW * C, b, C, G); });
  if (!m.active) return;
  k.finish(m, cbias != nullptr);
  float a1[8], a2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int g = j < m.split ? m.gA : m.gA + 1;
    a1[j] = res[g < G ? g : G - 1].x;
    a2[j] = res[g < G ? g : G - 1].y;
  }
  for (; p < p1; p += kU * m.RP) {
#pragma unroll
    for (int u = 0; u < kU; ++u)
      if (p + u * m.RP < p1) {
        Vec8<T> o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float yh = (Traits<T>::to_f32(v[u].v[j]) + k.cb[j] - k.mu[j]) * k.rs[j];
          const float dh = dyhat_of<ACT>(yh, Traits<T>::to_f32(d[u].v[j]), k.gm[j], k.bt[j]);
          const float add = gres != nullptr ? Traits<T>::to_f32(gr[u].v[j]) : 0.f;
          o.v[j] = Traits<T>::from_f32(k.rs[j] * (dh - a1[j] - yh * a2[j]) + add);
        }
        ob[(size_t)(p + u * m.RP) * m.VP] = o;
      }
    const int pn = p + kU * m.RP;
#pragma unroll
    for (int u = 0; u < kU; ++u)
      if (pn + u * m.RP < p1) {
        v[u] = xb[(size_t)(pn + u * m.RP) * m.VP];
        d[u] = db[(size_t)(pn + u * m.RP) * m.VP];
        if (gres != nullptr) gr[u] = rb[(size_t)(pn + u * m.RP) * m.VP];
      }
  }
}

// ---- small slabs (the 16x16 and 8x8 levels, and 32x32 at 640 channels: a group's slab of <= 20 480 elements): one launch,
// one workgroup per (image, group).
// The group's slab (HW pixels x C/G channels, <= 20 480 elements) is read once into REGISTERS while the sums are taken,
// reduced in-block, then normalised from there: a single latency chain instead of three launches.
constexpr int kSmallMaxElems = 20480;
// pixels per thread at most: ceil(HW / floor(NT / hp)) <= 2 HW hp / NT + 1, with HW hp <= 10240 at NT = 1024 and < 2048 at
// NT = 256 (small_wide): 21 and 17
constexpr int kSmallIters = 24;

template <int NT>
__device__ __forceinline__ void block_sum2(float& a, float& c, float* red) {
  constexpr int NW = NT / 64;
  a = wave_sum(a);   // DPP adds (ga_common.h): the six-step shuffle butterflies were twelve dependent LDS-crossbar round trips
  c = wave_sum(c);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    red[wave] = a;
    red[NW + wave] = c;
  }
  __syncthreads();
  float ta = 0.f, tc = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    ta += red[w];
    tc += red[NW + w];
  }
  a = ta;
  c = tc;
  __syncthreads();
}

// Thread t owns channel pair j = t % (Cg/2) of the group for pixels p0, p0 + rows, ... (rows = NT / (Cg/2)), so the
// per-channel gamma / beta / bias are loaded once and the loops carry no integer division.
// EVERYTHING a thread reads from memory — its <= kSmallIters slab pieces, the channel bias, gamma, beta — is requested in one
// batch up front and the slab stays in registers between the two passes.  The first form took the bias first (a round trip of
// its own), then the slab four pieces per round trip (`#pragma unroll 4`: five dependent round trips at 32 x 32 x 640), then
// gamma / beta behind the block reduction (cold weights: a third kind of round trip) — ~7 us for a 40 KB slab.
// IT = pieces per thread the launch is built for (the host takes the smallest of 4 / 8 / 12 / 24 that covers the shape): the
// loads are unconditional with clamped addresses — a load under `if (k < iters)` into a register array came out of the
// compiler as load, wait, copy, one at a time.
template <typename T, bool ACT, int NT, int IT, bool CAT>
// (argument order: what the first loads need — sources, sizes — inside the 16 dwords the hardware preloads into SGPRs; statistics,
// eps and the concatenation's destination, needed behind the reduction, come from the argument segment while the slab loads fly)
__global__ __launch_bounds__(NT) void gn_small_fwd_kernel(const T* __restrict__ x, const T* __restrict__ x2,
                                                          const T* __restrict__ cbias, int HW, int C, int G, int C1,
                                                          const T* __restrict__ gamma, const T* __restrict__ beta,
                                                          T* __restrict__ y, float* __restrict__ stats, float eps,
                                                          T* __restrict__ cat) {
  extern __shared__ __attribute__((aligned(16))) char smem_small[];
  float* red = reinterpret_cast<float*>(smem_small);
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (CAT) {   // the arguments past the 16 preloaded dwords, fetched here in one batch (left alone: one round trip each, where used)
    asm volatile("" ::"s"(stats), "s"(eps), "s"(cat));
  }
#endif
  const int g = blockIdx.x, b = blockIdx.y, Cg = C / G, hp = Cg >> 1;
  const int rows = NT / hp, p0 = threadIdx.x / hp, j = threadIdx.x - p0 * hp;
  const bool active = p0 < rows;
  const int ch = g * Cg + 2 * (active ? j : 0);
  const size_t base = (size_t)b * HW * C + ch;
  // x2 != nullptr: the norm's input is the CONCATENATION of x [B][HW][C1] and x2 [B][HW][C - C1] along the channels (the UpBlock's
  // torch.cat in front of resnet.norm1), never written before: this thread's channel pair lives in one of the two (C1 is even),
  // and `cat` receives the concatenated tensor on the way (the shortcut GEMM and the backward read it) — ga_cat_channels and
  // this norm as ONE launch
  // (CAT: a template parameter — the plain norm's instruction stream stays what it was)
  const bool second = CAT && ch >= C1;
  const T* src = !CAT ? x + base : (second ? x2 + (size_t)b * HW * (C - C1) + (ch - C1) : x + (size_t)b * HW * C1 + ch);
  const size_t sstride = !CAT ? (size_t)C : (size_t)(second ? C - C1 : C1);
  // no branch around any of these loads (a conditional load is waited for where it is issued): without a channel bias the
  // bias load reads a stand-in (not gamma: the compiler then re-uses gamma's register behind a wait) and a select drops it
  // the per-channel constants are requested LAST: whatever consumes one waits for everything, and everything is on its way
  Item<T, 2> xv[IT];
#pragma unroll
  for (int k = 0; k < IT; ++k) xv[k] = *reinterpret_cast<const Item<T, 2>*>(src + (size_t)min(p0 + k * rows, HW - 1) * sstride);
  // gamma / beta (cold weights: the slowest of these loads) go out last and are first needed behind the block reduction, the
  // channel bias (first pass) in front of them: the first pass's wait leaves the two outstanding and the reduction runs
  // under their latency
  Item<T, 2> cbv = xv[0];
  if constexpr (!CAT) cbv = load_pair<T>(cbias != nullptr ? cbias + (size_t)b * C + ch : src);   // CAT: no channel bias
  __builtin_amdgcn_sched_barrier(0);   // gamma / beta stay the YOUNGEST loads (the scheduler moved slab loads behind them)
  const Item<T, 2> gmv = load_pair<T>(gamma + ch), btv = load_pair<T>(beta + ch);
  const bool has_cb = !CAT && cbias != nullptr;
  const float cb0 = has_cb ? Traits<T>::to_f32(cbv.v[0]) : 0.f, cb1 = has_cb ? Traits<T>::to_f32(cbv.v[1]) : 0.f;
  float sa = 0.f, sc = 0.f;
#pragma unroll
  for (int k = 0; k < IT; ++k)
    if (active && p0 + k * rows < HW) {
      const float a0 = Traits<T>::to_f32(xv[k].v[0]) + cb0, a1 = Traits<T>::to_f32(xv[k].v[1]) + cb1;
      sa += a0 + a1;
      sc += a0 * a0 + a1 * a1;
    }
  block_sum2<NT>(sa, sc, red);
  const float inv_n = 1.0f / ((float)HW * (float)Cg);
  const double mean_d = (double)sa * inv_n;
  const float mean = (float)mean_d;
  const float rstd = rsqrtf((float)fmax((double)sc * inv_n - mean_d * mean_d, 0.0) + eps);
  if (threadIdx.x == 0) {
    stats[((size_t)b * G + g) * 2] = mean;
    stats[((size_t)b * G + g) * 2 + 1] = rstd;
  }
  if (!active) return;
  const float g0 = Traits<T>::to_f32(gmv.v[0]) * rstd, g1 = Traits<T>::to_f32(gmv.v[1]) * rstd;
  const float b0 = Traits<T>::to_f32(btv.v[0]) - mean * g0, b1 = Traits<T>::to_f32(btv.v[1]) - mean * g1;
#pragma unroll
  for (int k = 0; k < IT; ++k) {
    const int p = p0 + k * rows;
    if (p < HW) {
      float z0 = (Traits<T>::to_f32(xv[k].v[0]) + cb0) * g0 + b0, z1 = (Traits<T>::to_f32(xv[k].v[1]) + cb1) * g1 + b1;
      if (ACT) {
        z0 *= sigmoidf_(z0);
        z1 *= sigmoidf_(z1);
      }
      Item<T, 2> o;
      o.v[0] = Traits<T>::from_f32(z0);
      o.v[1] = Traits<T>::from_f32(z1);
      *reinterpret_cast<Item<T, 2>*>(y + base + (size_t)p * C) = o;
      if constexpr (CAT) *reinterpret_cast<Item<T, 2>*>(cat + base + (size_t)p * C) = xv[k];
    }
  }
}

// Backward of the same: x, dy, the skip connection's gradient, the forward's statistics and the per-channel constants are all
// requested in one batch (the first form read the statistics, then the constants, then x / dy four pixels per round trip).
// IT as in the forward; at IT = 24 (no SD shape) the skip connection's gradient is loaded where it is used (register budget).
template <typename T, bool ACT, int NT, int IT>
__global__ __launch_bounds__(NT) void gn_small_bwd_kernel(const T* __restrict__ x, const T* __restrict__ cbias,
                                                          const T* __restrict__ dy, const T* __restrict__ gamma,
                                                          const T* __restrict__ beta,
                                                          const float* __restrict__ stats,
                                                          const T* __restrict__ gres, T* __restrict__ dx, int HW, int C,
                                                          int G) {
  extern __shared__ __attribute__((aligned(16))) char smem_small[];
  float* red = reinterpret_cast<float*>(smem_small);
  float2* dh = reinterpret_cast<float2*>(red + 32);  // [HW][Cg/2] dL/dyhat
  const int g = blockIdx.x, b = blockIdx.y, Cg = C / G, hp = Cg >> 1;
  const int rows = NT / hp, p0 = threadIdx.x / hp, j = threadIdx.x - p0 * hp;
  const bool active = p0 < rows;
  const int ch = g * Cg + 2 * j;
  const size_t base = (size_t)b * HW * C + ch;
  constexpr bool kBatch = sizeof(T) == 2 || IT <= 12;       // fp32 at 24 pieces (no SD shape): loads where they are used
  constexpr bool kEarlyRes = kBatch && IT <= 12;             // register budget (128 VGPRs at 1024 threads)
  // no branch around any load, the per-channel constants last (see the forward)
  Item<T, 2> xv[kBatch ? IT : 1], dv[kBatch ? IT : 1], gr[kEarlyRes ? IT : 1];
  if constexpr (kBatch) {
#pragma unroll
    for (int k = 0; k < IT; ++k) {
      const size_t off = base + (size_t)min(p0 + k * rows, HW - 1) * C;
      xv[k] = *reinterpret_cast<const Item<T, 2>*>(x + off);
      dv[k] = *reinterpret_cast<const Item<T, 2>*>(dy + off);
    }
  }
  if constexpr (kEarlyRes) {
    const T* gsrc = gres != nullptr ? gres : dy;             // valid stand-in, dropped by the select below
#pragma unroll
    for (int k = 0; k < IT; ++k) gr[k] = *reinterpret_cast<const Item<T, 2>*>(gsrc + base + (size_t)min(p0 + k * rows, HW - 1) * C);
  }
  const float2 mr = *reinterpret_cast<const float2*>(stats + ((size_t)b * G + g) * 2);
  const Item<T, 2> gmv = *reinterpret_cast<const Item<T, 2>*>(gamma + ch), btv = *reinterpret_cast<const Item<T, 2>*>(beta + ch);
  const Item<T, 2> cbv = *reinterpret_cast<const Item<T, 2>*>(cbias != nullptr ? cbias + (size_t)b * C + ch : x + base);
  const float mean = mr.x, rstd = mr.y;
  const float cb0 = (cbias != nullptr ? Traits<T>::to_f32(cbv.v[0]) : 0.f) - mean;
  const float cb1 = (cbias != nullptr ? Traits<T>::to_f32(cbv.v[1]) : 0.f) - mean;
  const float g0 = Traits<T>::to_f32(gmv.v[0]), g1 = Traits<T>::to_f32(gmv.v[1]);
  const float b0 = Traits<T>::to_f32(btv.v[0]), b1 = Traits<T>::to_f32(btv.v[1]);
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int k = 0; k < IT; ++k) {
    const int p = p0 + k * rows;
    if (active && p < HW) {
      Item<T, 2> xk, dk;
      if constexpr (kBatch) {
        xk = xv[k];
        dk = dv[k];
      } else {
        xk = *reinterpret_cast<const Item<T, 2>*>(x + base + (size_t)p * C);
        dk = *reinterpret_cast<const Item<T, 2>*>(dy + base + (size_t)p * C);
      }
      const float y0 = (Traits<T>::to_f32(xk.v[0]) + cb0) * rstd;
      const float y1 = (Traits<T>::to_f32(xk.v[1]) + cb1) * rstd;
      const float d0 = dyhat_of<ACT>(y0, Traits<T>::to_f32(dk.v[0]), g0, b0);
      const float d1 = dyhat_of<ACT>(y1, Traits<T>::to_f32(dk.v[1]), g1, b1);
      dh[p * hp + j] = make_float2(d0, d1);   // own entries only: read back by this thread after the reduction
      s0 += d0 + d1;
      s1 += d0 * y0 + d1 * y1;
    }
  }
  block_sum2<NT>(s0, s1, red);
  if (!active) return;
  const float inv_n = 1.0f / ((float)HW * (float)Cg);
  const float m1 = s0 * inv_n, m2 = s1 * inv_n;
#pragma unroll
  for (int k = 0; k < IT; ++k) {
    const int p = p0 + k * rows;
    if (p < HW) {
      const float2 a = dh[p * hp + j];
      Item<T, 2> xk;
      if constexpr (kBatch) xk = xv[k];
      else xk = *reinterpret_cast<const Item<T, 2>*>(x + base + (size_t)p * C);
      const float y0 = (Traits<T>::to_f32(xk.v[0]) + cb0) * rstd;
      const float y1 = (Traits<T>::to_f32(xk.v[1]) + cb1) * rstd;
      float r0 = 0.f, r1 = 0.f;
      if (gres != nullptr) {
        Item<T, 2> gv;
        if constexpr (kEarlyRes) gv = gr[k];
        else gv = *reinterpret_cast<const Item<T, 2>*>(gres + base + (size_t)p * C);
        r0 = Traits<T>::to_f32(gv.v[0]);
        r1 = Traits<T>::to_f32(gv.v[1]);
      }
      Item<T, 2> o;
      o.v[0] = Traits<T>::from_f32(rstd * (a.x - m1 - y0 * m2) + r0);
      o.v[1] = Traits<T>::from_f32(rstd * (a.y - m1 - y1 * m2) + r1);
      *reinterpret_cast<Item<T, 2>*>(dx + base + (size_t)p * C) = o;
    }
  }
}

constexpr size_t kSmallHeader = 32 * sizeof(float);
inline bool small_wide(int HW, int C, int G) { return (C / G) / 2 > 256 || HW * ((C / G) / 2) >= 2048; }

struct SmallCat {        // the second source of a concatenated input and where the concatenation goes (all null / 0: plain input)
  const void* x2 = nullptr;
  int C1 = 0;
  void* cat = nullptr;
};

template <typename T, bool ACT, int NT, int IT>
int small_fwd_launch_it(const void* x, const void* cb, const void* gamma, const void* beta, void* y, float* stats, int B,
                        int HW, int C, int G, float eps, hipStream_t s, SmallCat sc) {
  const size_t lds = kSmallHeader;   // the slab stays in registers
  if (sc.x2 != nullptr) {
    auto k = gn_small_fwd_kernel<T, ACT, NT, IT, true>;
    hipLaunchKernelGGL(k, dim3(G, B), dim3(NT), lds, s, (const T*)x, (const T*)sc.x2, (const T*)nullptr, HW, C, G, sc.C1,
                       (const T*)gamma, (const T*)beta, (T*)y, stats, eps, (T*)sc.cat);
  } else {
    auto k = gn_small_fwd_kernel<T, ACT, NT, IT, false>;
    hipLaunchKernelGGL(k, dim3(G, B), dim3(NT), lds, s, (const T*)x, (const T*)nullptr, (const T*)cb, HW, C, G, 0,
                       (const T*)gamma, (const T*)beta, (T*)y, stats, eps, (T*)nullptr);
  }
  return check_launch();
}

// pieces per thread of a small-slab launch: ceil(HW / rows), rows = NT / (Cg / 2) — at most 21 (see kSmallIters)
inline int small_iters(int HW, int C, int G, int NT) {
  const int rows = NT / ((C / G) / 2);
  return (HW + rows - 1) / rows;
}

template <typename T, bool ACT, int NT>
int small_fwd_launch(const void* x, const void* cb, const void* gamma, const void* beta, void* y, float* stats, int B,
                     int HW, int C, int G, float eps, hipStream_t s, SmallCat sc) {
  const int it = small_iters(HW, C, G, NT);
  if (it <= 4) return small_fwd_launch_it<T, ACT, NT, 4>(x, cb, gamma, beta, y, stats, B, HW, C, G, eps, s, sc);
  if (it <= 8) return small_fwd_launch_it<T, ACT, NT, 8>(x, cb, gamma, beta, y, stats, B, HW, C, G, eps, s, sc);
  if (it <= 12) return small_fwd_launch_it<T, ACT, NT, 12>(x, cb, gamma, beta, y, stats, B, HW, C, G, eps, s, sc);
  if (it <= kSmallIters) return small_fwd_launch_it<T, ACT, NT, kSmallIters>(x, cb, gamma, beta, y, stats, B, HW, C, G, eps, s, sc);
  return GA_ERR_SHAPE;
}

template <typename T>
int small_fwd(const void* x, const void* cb, const void* gamma, const void* beta, void* y, float* stats, int B, int HW,
              int C, int G, float eps, int act, hipStream_t s, SmallCat sc = SmallCat{}) {
  if (small_wide(HW, C, G))
    return act ? small_fwd_launch<T, true, 1024>(x, cb, gamma, beta, y, stats, B, HW, C, G, eps, s, sc)
               : small_fwd_launch<T, false, 1024>(x, cb, gamma, beta, y, stats, B, HW, C, G, eps, s, sc);
  return act ? small_fwd_launch<T, true, 256>(x, cb, gamma, beta, y, stats, B, HW, C, G, eps, s, sc)
             : small_fwd_launch<T, false, 256>(x, cb, gamma, beta, y, stats, B, HW, C, G, eps, s, sc);
}

template <typename T, bool ACT, int NT, int IT>
int small_bwd_launch_it(const void* x, const void* cb, const void* dy, const void* gamma, const void* beta,
                        const float* stats, const void* gres, void* dx, int B, int HW, int C, int G, hipStream_t s) {
  const size_t lds = kSmallHeader + sizeof(float) * (size_t)HW * (C / G);   // dL/dyhat; x stays in registers
  auto k = gn_small_bwd_kernel<T, ACT, NT, IT>;
  const int rc = set_dyn_lds(k, lds);
  if (rc != GA_OK) return rc;
  hipLaunchKernelGGL(k, dim3(G, B), dim3(NT), lds, s, (const T*)x, (const T*)cb, (const T*)dy, (const T*)gamma,
                     (const T*)beta, stats, (const T*)gres, (T*)dx, HW, C, G);
  return check_launch();
}

template <typename T, bool ACT, int NT>
int small_bwd_launch(const void* x, const void* cb, const void* dy, const void* gamma, const void* beta,
                     const float* stats, const void* gres, void* dx, int B, int HW, int C, int G, hipStream_t s) {
  const int it = small_iters(HW, C, G, NT);
  if (it <= 4) return small_bwd_launch_it<T, ACT, NT, 4>(x, cb, dy, gamma, beta, stats, gres, dx, B, HW, C, G, s);
  if (it <= 8) return small_bwd_launch_it<T, ACT, NT, 8>(x, cb, dy, gamma, beta, stats, gres, dx, B, HW, C, G, s);
  if (it <= 12) return small_bwd_launch_it<T, ACT, NT, 12>(x, cb, dy, gamma, beta, stats, gres, dx, B, HW, C, G, s);
  if (it <= kSmallIters) return small_bwd_launch_it<T, ACT, NT, kSmallIters>(x, cb, dy, gamma, beta, stats, gres, dx, B, HW, C, G, s);
  return GA_ERR_SHAPE;
}

template <typename T>
int small_bwd(const void* x, const void* cb, const void* dy, const void* gamma, const void* beta, const float* stats,
              const void* gres, void* dx, int B, int HW, int C, int G, int act, hipStream_t s) {
  if (small_wide(HW, C, G))
    return act ? small_bwd_launch<T, true, 1024>(x, cb, dy, gamma, beta, stats, gres, dx, B, HW, C, G, s)
               : small_bwd_launch<T, false, 1024>(x, cb, dy, gamma, beta, stats, gres, dx, B, HW, C, G, s);
  return act ? small_bwd_launch<T, true, 256>(x, cb, dy, gamma, beta, stats, gres, dx, B, HW, C, G, s)
             : small_bwd_launch<T, false, 256>(x, cb, dy, gamma, beta, stats, gres, dx, B, HW, C, G, s);
}

bool small_path(int HW, int C, int G, size_t slab_bytes_per_elem) {
  const int Cg = C / G;
  return HW <= 1024 && (Cg & 1) == 0 && Cg <= 2048 && HW * Cg <= kSmallMaxElems &&
         kSmallHeader + slab_bytes_per_elem * (size_t)HW * Cg <= kMaxLdsBytes;
}

inline size_t elem_bytes(int dtype) { return dtype == GA_F32 ? 4 : 2; }

struct Geom {
  int NB, PBs, NBa, PBa, NPT, W;
  size_t lds;
};

int geometry(int B, int HW, int C, int G, Geom& g) {
  if (B < 1 || HW < 1 || C < 1 || G < 1 || G > 64 || C % G != 0) return GA_ERR_SHAPE;
  const int cg = C / G;
  g.W = (cg % 4 == 0) ? 4 : ((cg & 1) ? 1 : 2);  // 8-byte accesses when a group holds a multiple of 4 channels
  const int CP = C / g.W;
  g.NPT = (CP + kThreads - 1) / kThreads;
  if (g.NPT > kMaxNPT) return GA_ERR_SHAPE;
  g.PBs = (HW + kMaxStatsNB - 1) / kMaxStatsNB < 8 ? 8 : (HW + kMaxStatsNB - 1) / kMaxStatsNB;  // stats: <= 64 blocks
  g.NB = (HW + g.PBs - 1) / g.PBs;
  g.PBa = HW >= 4096 ? 16 : (HW >= 1024 ? 8 : 4);                            // apply: small blocks
  g.NBa = (HW + g.PBa - 1) / g.PBa;
  g.lds = sizeof(float) * 2 * CP;
  return GA_OK;
}

template <typename T, bool ACT, int NPT, int W>
int launch_fwd_t(const void* x, const void* cbias, const void* gamma, const void* beta, void* y, float* stats, float* ws,
                 int B, int HW, int C, int G, float eps, const Geom& g, hipStream_t s) {
  const float inv_n = 1.0f / ((float)HW * (float)(C / G));
  hipLaunchKernelGGL((gn_stats_kernel<T, NPT, W>), dim3(g.NB, B), dim3(kThreads), g.lds, s, (const T*)x,
                     (const T*)cbias, ws, HW, C, G, g.PBs);
  hipLaunchKernelGGL((gn_apply_kernel<T, ACT, NPT, W>), dim3(g.NBa, B), dim3(kThreads), 0, s, (const T*)x,
                     (const T*)cbias, (const T*)gamma, (const T*)beta, (T*)y, (const float*)ws, g.NB, inv_n, eps, stats,
                     HW, C, G, g.PBa);
  return check_launch();
}

template <typename T, bool ACT, int NPT, int W>
int launch_bwd_t(const void* x, const void* cbias, const void* dy, const void* gamma, const void* beta,
                 const float* stats, const void* gres, void* dx, float* ws, int B, int HW, int C, int G, const Geom& g,
                 hipStream_t s) {
  const float inv_n = 1.0f / ((float)HW * (float)(C / G));
  hipLaunchKernelGGL((gn_bwd_stats_kernel<T, ACT, NPT, W>), dim3(g.NB, B), dim3(kThreads), g.lds, s, (const T*)x,
                     (const T*)cbias, (const T*)dy, (const T*)gamma, (const T*)beta, stats, ws, HW, C, G, g.PBs);
  hipLaunchKernelGGL((gn_bwd_apply_kernel<T, ACT, NPT, W>), dim3(g.NBa, B), dim3(kThreads), 0, s, (const T*)x,
                     (const T*)cbias, (const T*)dy, (const T*)gamma, (const T*)beta, stats, (const float*)ws, g.NB, inv_n,
                     (const T*)gres, (T*)dx, HW, C, G, g.PBa);
  return check_launch();
}

#define GA_GN_NPT(FN, T, ACT, ...)                                   \
  if (g.W == 1) {                                                    \
    switch (g.NPT) {                                                 \
      case 1: return FN<T, ACT, 1, 1>(__VA_ARGS__);                  \
      case 2: return FN<T, ACT, 2, 1>(__VA_ARGS__);                  \
      default: return FN<T, ACT, 5, 1>(__VA_ARGS__);                 \
    }                                                                \
  }                                                                  \
  if (g.W == 4) {                                                    \
    switch (g.NPT) {                                                 \
      case 1: return FN<T, ACT, 1, 4>(__VA_ARGS__);                  \
      case 2: return FN<T, ACT, 2, 4>(__VA_ARGS__);                  \
      default: return FN<T, ACT, 3, 4>(__VA_ARGS__);                 \
    }                                                                \
  }                                                                  \
  switch (g.NPT) {                                                   \
    case 1: return FN<T, ACT, 1, 2>(__VA_ARGS__);                    \
    case 2: return FN<T, ACT, 2, 2>(__VA_ARGS__);                    \
    case 3: return FN<T, ACT, 3, 2>(__VA_ARGS__);                    \
    case 4: return FN<T, ACT, 4, 2>(__VA_ARGS__);                    \
    default: return FN<T, ACT, 5, 2>(__VA_ARGS__);                   \
  }

// wide path: geometry and launches
inline bool wide_ok(int C, int G, size_t elem) { return elem == 2 && C % 8 == 0 && C / G >= 8 && C <= 8 * kThreads; }

struct WideGeom {
  int PBs, NB, PBa, NBa;
  WideGeom(int HW, int C) {
    const int RP = kThreads / (C / 8);
    const int fill = (HW + kMaxStatsNB - 1) / kMaxStatsNB;     // stats: at most kMaxStatsNB partial blocks
    PBs = fill > 8 * RP ? fill : 8 * RP;             // >= 8 pixels per thread, all loaded in one batch where they fit
    NB = (HW + PBs - 1) / PBs;
    PBa = 4 * RP;                                    // apply: one 4-deep batch of loads per thread
    NBa = (HW + PBa - 1) / PBa;
  }
};

template <typename T>
int wide_fwd(const void* x, const void* cbias, const void* gamma, const void* beta, void* y, float* stats, float* ws,
             int B, int HW, int C, int G, float eps, int act, hipStream_t s) {
  const WideGeom g(HW, C);
  const float inv_n = 1.0f / ((float)HW * (float)(C / G));
  hipLaunchKernelGGL(gn_wide_stats_kernel<T>, dim3(g.NB, B), dim3(kThreads), 0, s, (const T*)x, (const T*)cbias, ws, HW,
                     C, G, g.PBs);
  if (act)
    hipLaunchKernelGGL((gn_wide_apply_kernel<T, true>), dim3(g.NBa, B), dim3(kThreads), 0, s, (const T*)x,
                       (const T*)cbias, (const T*)gamma, (const T*)beta, (T*)y, (const float*)ws, g.NB, inv_n, eps, stats,
                       HW, C, G, g.PBa);
  else
    hipLaunchKernelGGL((gn_wide_apply_kernel<T, false>), dim3(g.NBa, B), dim3(kThreads), 0, s, (const T*)x,
                       (const T*)cbias, (const T*)gamma, (const T*)beta, (T*)y, (const float*)ws, g.NB, inv_n, eps, stats,
                       HW, C, G, g.PBa);
  return check_launch();
}

template <typename T>
int wide_bwd(const void* x, const void* cbias, const void* dy, const void* gamma, const void* beta, const float* stats,
             const void* gres, void* dx, float* ws, int B, int HW, int C, int G, int act, hipStream_t s) {
  const WideGeom g(HW, C);
  const float inv_n = 1.0f / ((float)HW * (float)(C / G));
  if (act)
    hipLaunchKernelGGL((gn_wide_bwd_stats_kernel<T, true>), dim3(g.NB, B), dim3(kThreads), 0, s, (const T*)x,
                       (const T*)cbias, (const T*)dy, (const T*)gamma, (const T*)beta, stats, ws, HW, C, G, g.PBs);
  else
    hipLaunchKernelGGL((gn_wide_bwd_stats_kernel<T, false>), dim3(g.NB, B), dim3(kThreads), 0, s, (const T*)x,
                       (const T*)cbias, (const T*)dy, (const T*)gamma, (const T*)beta, stats, ws, HW, C, G, g.PBs);
  if (act)
    hipLaunchKernelGGL((gn_wide_bwd_apply_kernel<T, true>), dim3(g.NBa, B), dim3(kThreads), 0, s, (const T*)x,
                       (const T*)cbias, (const T*)dy, (const T*)gamma, (const T*)beta, stats, (const float*)ws, g.NB,
                       inv_n, (const T*)gres, (T*)dx, HW, C, G, g.PBa);
  else
    hipLaunchKernelGGL((gn_wide_bwd_apply_kernel<T, false>), dim3(g.NBa, B), dim3(kThreads), 0, s, (const T*)x,
                       (const T*)cbias, (const T*)dy, (const T*)gamma, (const T*)beta, stats, (const float*)ws, g.NB,
                       inv_n, (const T*)gres, (T*)dx, HW, C, G, g.PBa);
  return check_launch();
}

template <typename T>
int fwd_dtype(const void* x, const void* cbias, const void* gamma, const void* beta, void* y, float* stats, float* ws, int B,
              int HW, int C, int G, float eps, int act, const Geom& g, hipStream_t s) {
  if constexpr (sizeof(T) == 2) {
    if (wide_ok(C, G, sizeof(T))) return wide_fwd<T>(x, cbias, gamma, beta, y, stats, ws, B, HW, C, G, eps, act, s);
  }
  if (act) {
    GA_GN_NPT(launch_fwd_t, T, true, x, cbias, gamma, beta, y, stats, ws, B, HW, C, G, eps, g, s)
  }
  GA_GN_NPT(launch_fwd_t, T, false, x, cbias, gamma, beta, y, stats, ws, B, HW, C, G, eps, g, s)
}

template <typename T>
int bwd_dtype(const void* x, const void* cbias, const void* dy, const void* gamma, const void* beta, const float* stats,
              const void* gres, void* dx, float* ws, int B, int HW, int C, int G, int act, const Geom& g, hipStream_t s) {
  if constexpr (sizeof(T) == 2) {
    if (wide_ok(C, G, sizeof(T))) return wide_bwd<T>(x, cbias, dy, gamma, beta, stats, gres, dx, ws, B, HW, C, G, act, s);
  }
  if (act) {
    GA_GN_NPT(launch_bwd_t, T, true, x, cbias, dy, gamma, beta, stats, gres, dx, ws, B, HW, C, G, g, s)
  }
  GA_GN_NPT(launch_bwd_t, T, false, x, cbias, dy, gamma, beta, stats, gres, dx, ws, B, HW, C, G, g, s)
}

}  // namespace

extern "C" int ga_group_norm_fwd(const void* x, const void* chan_bias, const void* gamma, const void* beta, void* y,
                                 float* stats, float* workspace, int B, int HW, int C, int G, float eps, int act_silu,
                                 int dtype, ga_stream_t stream) {
  if (!x || !gamma || !beta || !y || !stats || !workspace) return GA_ERR_NULL;
  Geom g;
  int rc = geometry(B, HW, C, G, g);
  if (rc != GA_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (small_path(HW, C, G, sizeof(float))) {
    switch (dtype) {
      case GA_F16: return small_fwd<_Float16>(x, chan_bias, gamma, beta, y, stats, B, HW, C, G, eps, act_silu, s);
      case GA_BF16: return small_fwd<bf16_t>(x, chan_bias, gamma, beta, y, stats, B, HW, C, G, eps, act_silu, s);
      case GA_F32: return small_fwd<float>(x, chan_bias, gamma, beta, y, stats, B, HW, C, G, eps, act_silu, s);
      default: return GA_ERR_DTYPE;
    }
  }
  switch (dtype) {
    case GA_F16: return fwd_dtype<_Float16>(x, chan_bias, gamma, beta, y, stats, workspace, B, HW, C, G, eps, act_silu, g, s);
    case GA_BF16: return fwd_dtype<bf16_t>(x, chan_bias, gamma, beta, y, stats, workspace, B, HW, C, G, eps, act_silu, g, s);
    case GA_F32: return fwd_dtype<float>(x, chan_bias, gamma, beta, y, stats, workspace, B, HW, C, G, eps, act_silu, g, s);
    default: return GA_ERR_DTYPE;
  }
}

extern "C" int ga_group_norm_apply(const void* x, const void* chan_bias, const void* gamma, const void* beta, void* y,
                                   float* stats, const float* partials, int blocks, int B, int HW, int C, int G, float eps,
                                   int act_silu, int dtype, ga_stream_t stream) {
  // The second launch of the large-level forward alone, on partial sums somebody else took: the epilogue of the convolution
  // that PRODUCED x (ga_conv3x3_nhwc_gn).  Same kernel, same fold (fixed order) — only the statistics launch is gone.
  if (!x || !gamma || !beta || !y || !stats || !partials) return GA_ERR_NULL;
  Geom g;
  int rc = geometry(B, HW, C, G, g);
  if (rc != GA_OK) return rc;
  if (blocks < 1 || blocks > kMaxStatsNB) return GA_ERR_SHAPE;
  if (dtype == GA_F32 || !wide_ok(C, G, 2)) return GA_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const WideGeom wg(HW, C);
  const float inv_n = 1.0f / ((float)HW * (float)(C / G));
#define GA_GN_APPLY(T, ACT)                                                                                                   \
  hipLaunchKernelGGL((gn_wide_apply_kernel<T, ACT>), dim3(wg.NBa, B), dim3(kThreads), 0, s, (const T*)x, (const T*)chan_bias, \
                     (const T*)gamma, (const T*)beta, (T*)y, partials, blocks, inv_n, eps, stats, HW, C, G, wg.PBa)
  if (dtype == GA_F16) {
    if (act_silu) GA_GN_APPLY(_Float16, true);
    else GA_GN_APPLY(_Float16, false);
  } else if (dtype == GA_BF16) {
    if (act_silu) GA_GN_APPLY(bf16_t, true);
    else GA_GN_APPLY(bf16_t, false);
  } else {
    return GA_ERR_DTYPE;
  }
#undef GA_GN_APPLY
  return check_launch();
}

extern "C" int ga_group_norm_one_launch(int HW, int C, int G, int dtype) {
  /* 1 when ga_group_norm_fwd runs this shape as ONE launch (a group's slab stays in registers): the norms ga_cat_group_norm_fwd serves */
  if (HW < 1 || C < 1 || G < 1 || G > 64 || C % G != 0 || dtype < GA_F16 || dtype > GA_F32) return 0;
  return small_path(HW, C, G, sizeof(float)) ? 1 : 0;
}

extern "C" int ga_cat_group_norm_fwd(const void* a, const void* b, void* cat, const void* gamma, const void* beta, void* y,
                                     float* stats, int B, int HW, int C1, int C2, int G, float eps, int act_silu, int dtype,
                                     ga_stream_t stream) {
  /* y = [silu](group_norm(cat([a, b], channels))) AND cat itself, one launch, for the norms that are a single launch anyway (a
   * group's slab of <= 20 480 elements: the 16 x 16 and 8 x 8 levels): GA_ERR_UNSUPPORTED for the others (ga_cat_channels_gn +
   * ga_group_norm_apply serve those). */
  if (!a || !b || !cat || !gamma || !beta || !y || !stats) return GA_ERR_NULL;
  if (C1 < 2 || C2 < 2 || C1 % 2 != 0 || C2 % 2 != 0) return GA_ERR_SHAPE;
  const int C = C1 + C2;
  Geom g;
  int rc = geometry(B, HW, C, G, g);
  if (rc != GA_OK) return rc;
  if ((C / G) % 2 != 0) return GA_ERR_SHAPE;
  if (!small_path(HW, C, G, sizeof(float))) return GA_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const SmallCat sc{b, C1, cat};
  switch (dtype) {
    case GA_F16: return small_fwd<_Float16>(a, nullptr, gamma, beta, y, stats, B, HW, C, G, eps, act_silu, s, sc);
    case GA_BF16: return small_fwd<bf16_t>(a, nullptr, gamma, beta, y, stats, B, HW, C, G, eps, act_silu, s, sc);
    case GA_F32: return small_fwd<float>(a, nullptr, gamma, beta, y, stats, B, HW, C, G, eps, act_silu, s, sc);
    default: return GA_ERR_DTYPE;
  }
}

extern "C" int ga_cat_channels_gn_blocks(int HW, int C, int G, int dtype) {
  /* partial blocks per image ga_cat_channels_gn writes for a concatenated width C (0: not served — the consuming norm would be
   * one launch anyway, or is not on the wide path) */
  if (dtype == GA_F32 || HW < 1 || C < 1 || G < 1 || C % G != 0) return 0;
  if (small_path(HW, C, G, sizeof(float)) || !wide_ok(C, G, 2)) return 0;
  return WideGeom(HW, C).NB;
}

extern "C" int ga_cat_channels_gn(const void* a, const void* b, void* out, float* partials, int B, int HW, int C1, int C2, int G,
                                  int dtype, ga_stream_t stream) {
  if (!a || !b || !out || !partials) return GA_ERR_NULL;
  if (B < 1 || HW < 1 || C1 < 8 || C2 < 8 || C1 % 8 != 0 || C2 % 8 != 0 || G < 1) return GA_ERR_SHAPE;
  const int C = C1 + C2;
  if (C % G != 0 || (long long)B * HW * C >= (1LL << 31)) return GA_ERR_SHAPE;
  if (ga_cat_channels_gn_blocks(HW, C, G, dtype) == 0) return GA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(out)) & 15u) return GA_ERR_ALIGN;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const WideGeom wg(HW, C);
  if (dtype == GA_F16)
    hipLaunchKernelGGL(gn_wide_cat_stats_kernel<_Float16>, dim3(wg.NB, B), dim3(kThreads), 0, s, (const _Float16*)a,
                       (const _Float16*)b, (_Float16*)out, partials, HW, C1, C2, G, wg.PBs);
  else if (dtype == GA_BF16)
    hipLaunchKernelGGL(gn_wide_cat_stats_kernel<bf16_t>, dim3(wg.NB, B), dim3(kThreads), 0, s, (const bf16_t*)a, (const bf16_t*)b,
                       (bf16_t*)out, partials, HW, C1, C2, G, wg.PBs);
  else
    return GA_ERR_DTYPE;
  return check_launch();
}

extern "C" int ga_group_norm_two_launch(int HW, int C, int G, int dtype) {
  /* 1 when ga_group_norm_fwd takes the two-launch (statistics + apply) path for this shape in a 16-bit type — the norms for
   * which a producer's partial sums (ga_conv3x3_nhwc_gn + ga_group_norm_apply) save a launch */
  if (dtype == GA_F32 || HW < 1 || C < 1 || G < 1 || C % G != 0) return 0;
  return !small_path(HW, C, G, sizeof(float)) && wide_ok(C, G, 2) ? 1 : 0;
}

extern "C" int ga_group_norm_bwd(const void* x, const void* chan_bias, const void* dy, const void* gamma,
                                 const void* beta, const float* stats, const void* g_res, void* dx, float* workspace, int B,
                                 int HW, int C, int G, int act_silu, int dtype, ga_stream_t stream) {
  if (!x || !dy || !gamma || !beta || !stats || !dx || !workspace) return GA_ERR_NULL;
  Geom g;
  int rc = geometry(B, HW, C, G, g);
  if (rc != GA_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (small_path(HW, C, G, sizeof(float) + elem_bytes(dtype))) {
    switch (dtype) {
      case GA_F16: return small_bwd<_Float16>(x, chan_bias, dy, gamma, beta, stats, g_res, dx, B, HW, C, G, act_silu, s);
      case GA_BF16: return small_bwd<bf16_t>(x, chan_bias, dy, gamma, beta, stats, g_res, dx, B, HW, C, G, act_silu, s);
      case GA_F32: return small_bwd<float>(x, chan_bias, dy, gamma, beta, stats, g_res, dx, B, HW, C, G, act_silu, s);
      default: return GA_ERR_DTYPE;
    }
  }
  switch (dtype) {
    case GA_F16: return bwd_dtype<_Float16>(x, chan_bias, dy, gamma, beta, stats, g_res, dx, workspace, B, HW, C, G, act_silu, g, s);
    case GA_BF16: return bwd_dtype<bf16_t>(x, chan_bias, dy, gamma, beta, stats, g_res, dx, workspace, B, HW, C, G, act_silu, g, s);
    case GA_F32: return bwd_dtype<float>(x, chan_bias, dy, gamma, beta, stats, g_res, dx, workspace, B, HW, C, G, act_silu, g, s);
    default: return GA_ERR_DTYPE;
  }
}
