// K3+K4: the Gaussian-smoothed bounding-box loss over the aggregated maps A (res, res, Kt), forward
// and analytic backward.  Replaces the reference's per-pixel Python loops
// (pipeline_guided_attention.py:201-296,359-451; utils/helpers.py:164-173,215-277;
// utils/gaussian_smoothing.py:21-71).
//
// The whole problem is 16*16*77 floats (79 KB): it is launch-latency bound, not bandwidth bound, so
// it runs as ONE 256-thread workgroup that keeps every intermediate in LDS, walks the guided tokens
// sequentially (deterministic reductions, no atomics) and touches HBM once per input element.
//
//   S[p][j]  = softmax_j(100 * A[p][first + j]),  j in [0, last-first)
//   per guided token k:  M = S[:, k] -> M' = reflect-pad Gaussian smoothing -> s = sum M', Pn = M'/s
//       col = sum (j+.5) Pn, row = sum (i+.5) Pn, inside = 1 - sum_in Pn, outside = sum_out Pn
//       item = w_in*inside + 3*w_out*outside + w_c*(|col - res*cx| + 4|row - res*cy|)/(res-1)
//   loss = sum_k weight_k * item_k
// strict mode (curHyperParams["strict"], helpers.py:216-264): a per-pixel weight table W (inside: np.interp of the
// normalised distance from the box centre, outside: 1; normalised separately over the inside and the outside pixels)
// and hinge terms  inside = sum_in W * 2*max(0, 1/n_in - Pn),  outside = sum_out W * max(0, Pn).
// The reference hard-codes res = 16 ("16", "15."); res and res-1 are used here (identical at 16).
#include "aggregate.h"
#include "attn_common.h"

using namespace ga;

namespace {

constexpr int kMaxTok = 32;
constexpr int kMaxK = 7;
constexpr int kThreads = 256;

struct LossArgs {
  const float* A;
  int res, Kt, first, last, T;
  int ksize, smooth, strict;
  int stage_rows;   // pixel rows of A staged through LDS per pass of the softmax statistics (0: read from global memory)
  int use_gcol;     // the guided tokens' columns of A are kept in LDS ([T][npix]); 0 (LDS budget): re-read from global memory
  float w_in, w_out3, w_c;
  double shrink;
  ga_token_t tok[kMaxTok];
  float gw[kMaxK * kMaxK];
};

__device__ __forceinline__ int reflect_idx(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i); }

// helpers.py:164-173 inside_box at pixel centre (j+.5, i+.5); float64, same operation order as the
// reference (each product / sum rounded separately: no FMA contraction)
__device__ __forceinline__ bool inside_box(const ga_token_t& t, int res, double shrink, int i, int j) {
  const double ratio = (double)res;  // Rect.of_size: float(new_size / size), size = 1
  const double x = __dmul_rn(t.geom[0], ratio), y = __dmul_rn(t.geom[1], ratio);
  const double w = __dmul_rn(t.geom[2], ratio), h = __dmul_rn(t.geom[3], ratio);
  const double ox = __dmul_rn(shrink, w), oy = __dmul_rn(shrink, h);
  const double cx = (double)j + 0.5, cy = (double)i + 0.5;
  if (cx >= __dadd_rn(x, ox) && cx <= __dsub_rn(__dadd_rn(x, w), ox))
    if (cy >= __dadd_rn(y, oy) && cy <= __dsub_rn(__dadd_rn(y, h), oy)) return true;
  return false;
}

// block-wide sum of NV values per thread; result valid in every thread.  scratch: [4][NV] floats.
template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* scratch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = wave_reduce_sum(v[k]);
  __syncthreads();  // scratch may still be read from a previous call
  if (lane == 0)
#pragma unroll
    for (int k = 0; k < NV; ++k) scratch[wave * NV + k] = v[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = (scratch[k] + scratch[NV + k]) + (scratch[2 * NV + k] + scratch[3 * NV + k]);
}

__device__ __forceinline__ float block_max(float v, float* scratch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = wave_reduce_max(v);
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  return fmaxf(fmaxf(scratch[0], scratch[1]), fmaxf(scratch[2], scratch[3]));
}

// per-pixel softmax statistics of 100*A over the text slice: row max and sum of exponentials, each row summed in token
// order by ONE thread (the order the fixtures of the reference's fp32 softmax were matched with: a tree-ordered sum moves
// the near-one-hot rows by more than the 3e-5 bar after the x100 backward).  The rows are staged through LDS in chunks
// of `stage_rows` pixels with coalesced 16-byte loads (a chunk of A is contiguous), and a thread then walks its row in LDS
// (row stride Kt words: conflict-free for the odd Kt of the text context).  Walking the rows straight from global memory
// — 64 different cache lines per load instruction, 150 dependent-latency loads per thread — was most of the 35 us the
// single-workgroup kernel took in round 2; `stage_rows` = 0 (maps too large for the LDS budget) keeps that form.
__device__ __forceinline__ float* align16(float* p) {
  return reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(p) + 15) & ~(uintptr_t)15);
}

__device__ __forceinline__ void row_stats(const LossArgs& a, const float* row, float& m_out, float& s_out) {
  // eight reads in flight per trip; the maximum and the sum still run in token order (one thread, one row).  The plain
  // `for c: m = max(m, row[c] * 100)` loop paid a full LDS round trip per element: 15 us of the launch for 256 rows.
  float m = -INFINITY;
  for (int c0 = a.first; c0 < a.last; c0 += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = row[min(c0 + u, a.last - 1)] * 100.0f;
#pragma unroll
    for (int u = 0; u < 8; ++u) m = fmaxf(m, v[u]);          // the clamped repeats of the last element change nothing
  }
  float s = 0.f;
  for (int c0 = a.first; c0 < a.last; c0 += 8) {
    float e[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) e[u] = expf(row[min(c0 + u, a.last - 1)] * 100.0f - m);
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (c0 + u < a.last) s += e[u];
  }
  m_out = m;
  s_out = s;
}

// gcol[t][p] = A[p][column of guided token t]: taken while the row is at hand, so that the token loops never go back to
// global memory (one dependent ~1-2 us load per token and phase otherwise: the launch is one workgroup, nothing hides it)
__device__ __forceinline__ void gather_guided(const LossArgs& a, const float* row, int p, int npix, float* gcol) {
  if (!a.use_gcol) return;
  for (int t = 0; t < a.T; ++t) gcol[t * npix + p] = row[a.first + a.tok[t].token - 1];
}
// A[p][column of guided token t]
__device__ __forceinline__ float guided_value(const LossArgs& a, const float* gcol, int t, int p, int npix) {
  return a.use_gcol ? gcol[(size_t)t * npix + p] : a.A[(size_t)p * a.Kt + a.first + a.tok[t].token - 1];
}

__device__ __forceinline__ void pixel_softmax_stats(const LossArgs& a, float* mx, float* sm, float* stage, float* gcol) {
  const int npix = a.res * a.res;
  if (a.stage_rows == 0) {
    for (int p = threadIdx.x; p < npix; p += kThreads) {
      row_stats(a, a.A + (size_t)p * a.Kt, mx[p], sm[p]);
      gather_guided(a, a.A + (size_t)p * a.Kt, p, npix, gcol);
    }
    return;
  }
  for (int p0 = 0; p0 < npix; p0 += a.stage_rows) {
    const int rows = min(a.stage_rows, npix - p0), n = rows * a.Kt;
    const float* src = a.A + (size_t)p0 * a.Kt;      // 16-byte aligned: stage_rows is a multiple of 4, A is
    // eight 16-byte loads per thread in flight before the first LDS store (a load -> store loop pays one memory round
    // trip per iteration: 19 of them for the 16 x 16 x 77 map, most of what this launch took)
    for (int e0 = 0; e0 + 3 < n; e0 += 4 * kThreads * 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + 4 * (threadIdx.x + u * kThreads);
        v[u] = *reinterpret_cast<const f32x4*>(src + min(e, (n & ~3) - 4));
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + 4 * (threadIdx.x + u * kThreads);
        if (e + 3 < n) *reinterpret_cast<f32x4*>(stage + e) = v[u];
      }
    }
    for (int e = (n & ~3) + threadIdx.x; e < n; e += kThreads) stage[e] = src[e];
    __syncthreads();
    for (int r = threadIdx.x; r < rows; r += kThreads) {
      row_stats(a, stage + r * a.Kt, mx[p0 + r], sm[p0 + r]);
      gather_guided(a, stage + r * a.Kt, p0 + r, npix, gcol);
    }
    __syncthreads();
  }
}

struct TokenStats {
  float s, mxv, col, row, in, out, at_most;
};

// helpers.py:159-162 get_corresponding_weight = np.interp(x, [0, .333, .666, 1], [3, 2.5, 1, .2]) in float64
__device__ __forceinline__ double interp_weight(double x) {
  const double xp[4] = {0.0, .333, .666, 1.0}, fp[4] = {3.0, 2.5, 1.0, .2};
  if (x <= xp[0]) return fp[0];
  if (x >= xp[3]) return fp[3];
  int j = 0;
  while (j < 2 && x >= xp[j + 1]) ++j;
  const double slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]);
  return slope * (x - xp[j]) + fp[j];
}

// helpers.py:216-246: the strict-mode weight table of one BOX token into W[npix] (LDS), normalised separately over
// the inside and the outside pixels; the two sums run in pixel order in fp32 like the reference's (one thread: 256 to
// 4096 adds, strict mode is off by default).  Returns 1/n_inside rounded to fp32 (`at_most`, helpers.py:249).
__device__ __forceinline__ float strict_weights(const LossArgs& a, const ga_token_t& tk, float* W, float* scratch) {
  const int res = a.res, npix = res * res;
  const double ratio = (double)res;
  const double x = __dmul_rn(tk.geom[0], ratio), y = __dmul_rn(tk.geom[1], ratio);
  const double w = __dmul_rn(tk.geom[2], ratio), h = __dmul_rn(tk.geom[3], ratio);
  const double ccx = __dadd_rn(x, w / 2.0), ccy = __dadd_rn(y, h / 2.0);  // Rect.center() of the scaled rect
  for (int p = threadIdx.x; p < npix; p += kThreads) {
    const int i = p / res, j = p - i * res;
    float wv = 1.0f;  // outside: get_corresponding_weight_distance_from == 1
    if (inside_box(tk, res, a.shrink, i, j)) {
      const double dx = __ddiv_rn(__dmul_rn(2.0, __dsub_rn(ccx, (double)j + 0.5)), w);
      const double dy = __ddiv_rn(__dmul_rn(2.0, __dsub_rn(ccy, (double)i + 0.5)), h);
      const double d = __ddiv_rn(__dsqrt_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy))), __dsqrt_rn(2.0));
      wv = (float)interp_weight(d);
    }
    W[p] = wv;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float s_in = 0.f, s_out = 0.f;
    int n_in = 0;
    for (int p = 0; p < npix; ++p) {
      const int i = p / res, j = p - i * res;
      if (inside_box(tk, res, a.shrink, i, j)) {
        s_in += W[p];
        ++n_in;
      } else {
        s_out += W[p];
      }
    }
    scratch[0] = s_in;
    scratch[1] = s_out;
    scratch[2] = (float)(1.0 / (double)n_in);
  }
  __syncthreads();
  const float s_in = scratch[0], s_out = scratch[1], at_most = scratch[2];
  for (int p = threadIdx.x; p < npix; p += kThreads) {
    const int i = p / res, j = p - i * res;
    W[p] = W[p] / (inside_box(tk, res, a.shrink, i, j) ? s_in : s_out);
  }
  __syncthreads();
  return at_most;
}

// Forward of one token into LDS: M (raw map), Pn (smoothed, normalised).  Returns the reductions.
__device__ __forceinline__ TokenStats token_forward(const LossArgs& a, const ga_token_t& tk, const float* mx,
                                                    const float* sm, const float* gcol, int t_idx, float* M, float* Pn,
                                                    float* W, float* scratch) {
  const int res = a.res, npix = res * res;
  const bool strict = a.strict && tk.kind == GA_TOK_BOX;
  float at_most = 0.f;
  if (strict) at_most = strict_weights(a, tk, W, scratch);
  // A[p][first + token - 1]  (pipeline:228 "index - 1" into the [first:last) slice)
  for (int p = threadIdx.x; p < npix; p += kThreads) M[p] = expf(guided_value(a, gcol, t_idx, p, npix) * 100.0f - mx[p]) / sm[p];
  __syncthreads();
  const int pad = a.ksize >> 1;
  float v2[2] = {0.f, 0.f};
  float vmax = -INFINITY;
  for (int p = threadIdx.x; p < npix; p += kThreads) {
    float acc;
    if (a.smooth) {
      const int i = p / res, j = p - i * res;
      acc = 0.f;
      for (int u = 0; u < a.ksize; ++u) {
        const int ii = reflect_idx(i + u - pad, res);
        for (int v = 0; v < a.ksize; ++v) acc += a.gw[u * a.ksize + v] * M[ii * res + reflect_idx(j + v - pad, res)];
      }
    } else {
      acc = M[p];
    }
    Pn[p] = acc;
    v2[0] += acc;
    vmax = fmaxf(vmax, acc);
  }
  block_sum<2>(v2, scratch);
  TokenStats st;
  st.s = v2[0];
  st.mxv = block_max(vmax, scratch);
  float v4[4] = {0.f, 0.f, 0.f, 0.f};
  for (int p = threadIdx.x; p < npix; p += kThreads) {
    const int i = p / res, j = p - i * res;
    const float pn = Pn[p] / st.s;
    Pn[p] = pn;
    v4[0] += ((float)j + 0.5f) * pn;
    v4[1] += ((float)i + 0.5f) * pn;
    if (tk.kind == GA_TOK_BOX) {
      const bool in = inside_box(tk, res, a.shrink, i, j);
      if (strict) {  // helpers.py:250-264
        if (in)
          v4[2] += W[p] * (2.0f * fmaxf(0.f, at_most - pn));
        else
          v4[3] += W[p] * fmaxf(0.f, pn);
      } else if (in) {
        v4[2] += pn;
      } else {
        v4[3] += pn;
      }
    }
  }
  block_sum<4>(v4, scratch);
  st.col = v4[0];
  st.row = v4[1];
  st.in = v4[2];
  st.out = v4[3];
  st.at_most = at_most;
  return st;
}

struct TokenLoss {
  float inside, outside, item, unscaled, dc, dr, w_in, w_out3, w_c;
};

__device__ __forceinline__ TokenLoss token_loss(const LossArgs& a, const ga_token_t& tk, const TokenStats& st) {
  TokenLoss r;
  const float res = (float)a.res;
  float cx, cy;
  if (tk.kind == GA_TOK_BOX) {  // helpers.py:26-27 Rect.center in float64, then used against fp32 tensors
    cx = (float)(tk.geom[0] + tk.geom[2] / 2.0);
    cy = (float)(tk.geom[1] + tk.geom[3] / 2.0);
    r.inside = a.strict ? st.in : 1.0f - st.in;  // helpers.py:261 (strict) / :275
    r.outside = st.out;                          // helpers.py:263 (strict) / :276
    r.w_in = a.w_in;
    r.w_out3 = a.w_out3;
    r.w_c = a.w_c > 0.f ? a.w_c : 0.f;
  } else {
    cx = (float)tk.geom[0];
    cy = (float)tk.geom[1];
    r.inside = r.outside = 0.f;
    r.w_in = r.w_out3 = 0.f;
    r.w_c = 1.0f;
  }
  r.dc = st.col - cx * res;
  r.dr = st.row - cy * res;
  const float centering = fabsf(r.dc) / (res - 1.0f) + 4.0f * fabsf(r.dr) / (res - 1.0f);  // pipeline:391-395
  if (tk.kind == GA_TOK_BOX) {
    r.item = r.w_in * r.inside + r.w_out3 * r.outside + r.w_c * centering;  // pipeline:423-434
    r.unscaled = r.inside + r.outside;
  } else {
    r.item = r.unscaled = centering;  // pipeline:409-414
  }
  return r;
}

__device__ __forceinline__ void loss_forward(const LossArgs& a, float* lds, float* __restrict__ terms,
                                             float* __restrict__ loss) {
  const int npix = a.res * a.res;
  float* mx = lds;
  float* sm = mx + npix;
  float* M = sm + npix;
  float* Pn = M + npix;
  float* scratch = Pn + npix;  // 16 floats
  float* W = scratch + 16;     // [npix], strict mode only
  float* gcol = W + (a.strict ? npix : 0);             // [T][npix]
  float* stage = align16(gcol + (a.use_gcol ? (size_t)a.T * npix : 0));   // [stage_rows][Kt]
  pixel_softmax_stats(a, mx, sm, stage, gcol);
  __syncthreads();
  float total = 0.f;
  for (int t = 0; t < a.T; ++t) {
    const ga_token_t& tk = a.tok[t];
    const TokenStats st = token_forward(a, tk, mx, sm, gcol, t, M, Pn, W, scratch);
    const TokenLoss tl = token_loss(a, tk, st);
    total += tk.weight * tl.item;
    if (threadIdx.x == 0) {
      float* o = terms + t * GA_TERMS;
      o[0] = st.mxv;
      o[1] = st.col;
      o[2] = st.row;
      o[3] = tl.inside;
      o[4] = tl.outside;
      o[5] = tl.item;
      o[6] = tl.unscaled;
      o[7] = st.s;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = total;
}

__global__ __launch_bounds__(kThreads) void smooth_loss_fwd_kernel(LossArgs a, float* __restrict__ terms,
                                                                   float* __restrict__ loss) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  loss_forward(a, lds, terms, loss);
}

// K2 + K3 + K4 in ONE launch (utils/ptp_utils.py:279-289 -> pipeline_guided_attention.py:217-219): every workgroup
// averages 256 (pixel, token) elements over the head-maps (aggregate.h: list order, no atomics) and stores them to A;
// the workgroup whose ticket comes last then evaluates the loss on the complete A.  Hand-off: stores drained by every
// wave, workgroup barrier, one agent-scope release + one relaxed ticket add per workgroup; the last arriver makes one
// agent-scope acquire before its plain loads of A and returns the ticket word to zero for the next launch.
template <typename T>
__global__ __launch_bounds__(kThreads) void aggregate_loss_fwd_kernel(AggArgs g, LossArgs a, int n_elem, float* __restrict__ A,
                                                                      float* __restrict__ terms, float* __restrict__ loss,
                                                                      unsigned* __restrict__ ticket) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int e = blockIdx.x * kThreads + threadIdx.x;
  if (e < n_elem) A[e] = aggregate_element<T>(g, e, n_elem);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int* flag = reinterpret_cast<int*>(lds);
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the fence's own wait can be dropped by the compiler: keep this one
    const unsigned old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = old == gridDim.x - 1;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    flag[0] = last;
  }
  __syncthreads();
  const int last = flag[0];
  __syncthreads();   // the flag word is part of the loss's LDS image
  if (!last) return;
  loss_forward(a, lds, terms, loss);
}

template <typename T>
__global__ __launch_bounds__(kThreads) void smooth_loss_bwd_kernel(LossArgs a, const float* __restrict__ dloss,
                                                                   float* __restrict__ dA, T* __restrict__ dPb,
                                                                   float bcast_scale) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int res = a.res, npix = res * res;
  float* mx = lds;
  float* sm = mx + npix;
  float* M = sm + npix;
  float* Pn = M + npix;
  float* G = Pn + npix;         // dItem/dM' per pixel
  float* dot = G + npix;        // sum_k dS[p][k] S[p][k]
  float* scratch = dot + npix;  // 16 floats
  int* colmap = reinterpret_cast<int*>(scratch + 16);  // [Kt]: guided-token slot of column c, or -1
  float* dS = reinterpret_cast<float*>(colmap + ((a.Kt + 3) & ~3));  // [T][npix]
  float* W = dS + (size_t)a.T * npix;                                // [npix], strict mode only
  float* gcol = W + (a.strict ? npix : 0);                           // [T][npix]
  float* stage = align16(gcol + (a.use_gcol ? (size_t)a.T * npix : 0));   // [stage_rows][Kt]

  pixel_softmax_stats(a, mx, sm, stage, gcol);
  for (int c = threadIdx.x; c < a.Kt; c += kThreads) colmap[c] = -1;
  __syncthreads();
  const int pad = a.ksize >> 1;
  const float rm1 = (float)res - 1.0f;
  for (int t = 0; t < a.T; ++t) {
    const ga_token_t& tk = a.tok[t];
    const TokenStats st = token_forward(a, tk, mx, sm, gcol, t, M, Pn, W, scratch);
    const TokenLoss tl = token_loss(a, tk, st);
    if (threadIdx.x == 0) colmap[a.first + tk.token - 1] = t;
    const float sgc = tl.dc > 0.f ? 1.f : (tl.dc < 0.f ? -1.f : 0.f);
    const float sgr = tl.dr > 0.f ? 1.f : (tl.dr < 0.f ? -1.f : 0.f);
    float gd[1] = {0.f};
    for (int p = threadIdx.x; p < npix; p += kThreads) {
      const int i = p / res, j = p - i * res;
      float g = tl.w_c * (sgc / rm1 * ((float)j + 0.5f) + 4.0f * sgr / rm1 * ((float)i + 0.5f));
      if (tk.kind == GA_TOK_BOX) {
        const bool in = inside_box(tk, res, a.shrink, i, j);
        if (a.strict)  // hinge terms: the gradient passes only where the hinge is open (Python max(min_loss, v): v > 0)
          g += in ? (st.at_most - Pn[p] > 0.f ? -2.0f * tl.w_in * W[p] : 0.f) : (Pn[p] > 0.f ? tl.w_out3 * W[p] : 0.f);
        else
          g += in ? -tl.w_in : tl.w_out3;
      }
      G[p] = g;
      gd[0] += g * Pn[p];
    }
    block_sum<1>(gd, scratch);
    for (int p = threadIdx.x; p < npix; p += kThreads) G[p] = (G[p] - gd[0]) / st.s;  // through Pn = M'/sum(M')
    __syncthreads();
    // adjoint of reflect-pad + correlation, as a gather (deterministic): dM[a][b] = sum over (i,u),(j,v)
    // with reflect(i+u-pad) == a and reflect(j+v-pad) == b of gw[u][v] * G[i][j]
    float* dSt = dS + (size_t)t * npix;
    for (int p = threadIdx.x; p < npix; p += kThreads) {
      float acc;
      if (a.smooth) {
        const int ai = p / res, bj = p - ai * res;
        acc = 0.f;
        for (int i = max(ai - pad, 0); i <= min(ai + pad, res - 1); ++i)
          for (int u = 0; u < a.ksize; ++u) {
            if (reflect_idx(i + u - pad, res) != ai) continue;
            for (int j = max(bj - pad, 0); j <= min(bj + pad, res - 1); ++j)
              for (int v = 0; v < a.ksize; ++v)
                if (reflect_idx(j + v - pad, res) == bj) acc += a.gw[u * a.ksize + v] * G[i * res + j];
          }
      } else {
        acc = G[p];
      }
      dSt[p] = tk.weight * acc;
    }
    __syncthreads();
  }
  // softmax backward: dA[p][c] = 100 * S[p][c] * (dS[p][c] - sum_k dS[p][k] S[p][k]) on the text slice
  for (int p = threadIdx.x; p < npix; p += kThreads) {
    float d = 0.f;
    for (int t = 0; t < a.T; ++t) d += dS[(size_t)t * npix + p] * (expf(guided_value(a, gcol, t, p, npix) * 100.0f - mx[p]) / sm[p]);
    dot[p] = d;
  }
  __syncthreads();
  // every workgroup has derived the same per-pixel tables (the token work above is 3 x 256 pixels: repeating it costs
  // less than any hand-off); the element-wise tail over the (pixel, token) grid is split between them
  const float dl = dloss ? dloss[0] : 1.0f;
  const int total = npix * a.Kt;
  for (int e0 = blockIdx.x * kThreads + threadIdx.x; e0 < total; e0 += 4 * gridDim.x * kThreads) {
    float av[4];   // four loads in flight per thread, then the math
#pragma unroll
    for (int u = 0; u < 4; ++u) av[u] = a.A[min(e0 + u * (int)(gridDim.x * kThreads), total - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + u * (int)(gridDim.x * kThreads);
      if (e >= total) break;
      const int p = e / a.Kt, c = e - p * a.Kt;
      float g = 0.f;
      if (c >= a.first && c < a.last) {
        const float S = expf(av[u] * 100.0f - mx[p]) / sm[p];
        const int t = colmap[c];
        g = dl * 100.0f * S * ((t >= 0 ? dS[(size_t)t * npix + p] : 0.f) - dot[p]);
      }
      dA[e] = g;
      if (dPb) dPb[e] = Traits<T>::from_f32(g * bcast_scale);
    }
  }
}

size_t fwd_lds(int npix, int strict) { return sizeof(float) * ((4 + (strict ? 1 : 0)) * (size_t)npix + 16); }
size_t bwd_lds(int npix, int Kt, int T, int strict) {
  return sizeof(float) * ((6 + (strict ? 1 : 0)) * (size_t)npix + 16 + ((Kt + 3) & ~3) + (size_t)T * npix);
}
constexpr size_t kLdsBudget = 150 * 1024;
// rows of A staged per pass: as many as fit beside the kernel's tables, at most one per thread, a multiple of 4
int choose_stage_rows(size_t base_lds, int npix, int Kt, const float* A) {
  if ((reinterpret_cast<uintptr_t>(A) & 15) != 0) return 0;
  base_lds += 16;   // the staging area starts on a 16-byte boundary
  for (int rows = kThreads; rows >= 32; rows >>= 1) {
    const int r = rows < npix ? rows : ((npix + 3) & ~3);
    if (base_lds + sizeof(float) * (size_t)r * Kt <= kLdsBudget) return r;
  }
  return 0;
}

int fill_args(LossArgs& a, const float* A, int res, int Kt, int first, int last, const ga_token_t* tokens, int T,
              const ga_loss_params_t* hp) {
  if (!A || !tokens || !hp) return GA_ERR_NULL;
  if (res < 2 || res > 64 || Kt < 2 || T < 1 || T > kMaxTok) return GA_ERR_SHAPE;
  if (first < 0 || last > Kt || last - first < 1) return GA_ERR_SHAPE;
  if (hp->smooth && (hp->ksize < 1 || hp->ksize > kMaxK || (hp->ksize & 1) == 0 || hp->ksize / 2 >= res))
    return GA_ERR_SHAPE;
  for (int t = 0; t < T; ++t) {
    const int col = first + tokens[t].token - 1;
    if (col < first || col >= last) return GA_ERR_SHAPE;
    if (tokens[t].kind != GA_TOK_BOX && tokens[t].kind != GA_TOK_COOR) return GA_ERR_UNSUPPORTED;
    a.tok[t] = tokens[t];
  }
  a.A = A;
  a.res = res;
  a.Kt = Kt;
  a.first = first;
  a.last = last;
  a.T = T;
  a.ksize = hp->smooth ? hp->ksize : 1;
  a.smooth = hp->smooth ? 1 : 0;
  a.strict = hp->strict ? 1 : 0;
  a.w_in = hp->inside_scale;
  a.w_out3 = hp->outside_scale * 3.0f;
  a.w_c = hp->center_weight;
  a.shrink = hp->shrink;
  if (a.smooth) {
    int rc = ga_gaussian_weights(hp->ksize, hp->sigma, a.gw);
    if (rc != GA_OK) return rc;
  }
  return GA_OK;
}

}  // namespace

extern "C" int ga_gaussian_weights(int ksize, float sigma, float* w) {
  if (!w) return GA_ERR_NULL;
  if (ksize < 1 || ksize > kMaxK || !(sigma > 0.f)) return GA_ERR_SHAPE;
  // utils/gaussian_smoothing.py:30-43 in fp32: per dimension 1/(std*sqrt(2*pi)) * exp(-((x-mean)/(2*std))^2),
  // product over the two dimensions, then divided by the sum
  float g1[kMaxK];
  const float mean = (float)((ksize - 1) / 2.0);
  const float norm = (float)(1.0 / ((double)sigma * 2.5066282746310002));
  for (int i = 0; i < ksize; ++i) {
    const float z = ((float)i - mean) / (2.0f * sigma);
    g1[i] = norm * expf(-(z * z));
  }
  float sum = 0.f;
  for (int i = 0; i < ksize; ++i)
    for (int j = 0; j < ksize; ++j) {
      w[i * ksize + j] = g1[i] * g1[j];
      sum += w[i * ksize + j];
    }
  for (int i = 0; i < ksize * ksize; ++i) w[i] /= sum;
  return GA_OK;
}

extern "C" int ga_smooth_loss_fwd(const float* A, int res, int Kt, int first, int last, const ga_token_t* tokens, int T,
                                  const ga_loss_params_t* hp, float* terms, float* loss, ga_stream_t stream) {
  if (!terms || !loss) return GA_ERR_NULL;
  LossArgs a;
  int rc = fill_args(a, A, res, Kt, first, last, tokens, T, hp);
  if (rc != GA_OK) return rc;
  size_t lds = fwd_lds(res * res, a.strict);
  if (lds > kLdsBudget) return GA_ERR_SHAPE;
  a.use_gcol = lds + sizeof(float) * (size_t)T * res * res <= kLdsBudget / 2 ? 1 : 0;   // the columns first, the staging area with what is left
  if (a.use_gcol) lds += sizeof(float) * (size_t)T * res * res;
  a.stage_rows = choose_stage_rows(lds, res * res, Kt, A);
  lds += sizeof(float) * (size_t)a.stage_rows * Kt + 16;
  if (set_dyn_lds(smooth_loss_fwd_kernel, lds) != GA_OK) return GA_ERR_LAUNCH;
  hipLaunchKernelGGL(smooth_loss_fwd_kernel, dim3(1), dim3(kThreads), lds, static_cast<hipStream_t>(stream), a, terms,
                     loss);
  return check_launch();
}

template <typename T>
static int launch_loss_bwd(const LossArgs& a, const float* dloss, float* dA, void* dPb, float bs, size_t lds,
                           hipStream_t s) {
  auto k = smooth_loss_bwd_kernel<T>;
  if (set_dyn_lds(k, lds) != GA_OK) return GA_ERR_LAUNCH;
  // up to 16 workgroups, at least 4 elements of the tail per thread
  const int total = a.res * a.res * a.Kt;
  const int wgs = max(1, min(16, total / (4 * kThreads)));
  hipLaunchKernelGGL(k, dim3(wgs), dim3(kThreads), lds, s, a, dloss, dA, (T*)dPb, bs);
  return check_launch();
}

extern "C" int ga_smooth_loss_bwd(const float* A, int res, int Kt, int first, int last, const ga_token_t* tokens, int T,
                                  const ga_loss_params_t* hp, const float* dloss, float* dA, void* dP_bcast,
                                  float bcast_scale, int dtype, ga_stream_t stream) {
  if (!dA) return GA_ERR_NULL;
  LossArgs a;
  int rc = fill_args(a, A, res, Kt, first, last, tokens, T, hp);
  if (rc != GA_OK) return rc;
  size_t lds = bwd_lds(res * res, Kt, T, a.strict);
  if (lds > kLdsBudget) return GA_ERR_SHAPE;
  a.use_gcol = lds + sizeof(float) * (size_t)T * res * res <= kLdsBudget / 2 ? 1 : 0;   // the columns first, the staging area with what is left
  if (a.use_gcol) lds += sizeof(float) * (size_t)T * res * res;
  a.stage_rows = choose_stage_rows(lds, res * res, Kt, A);
  lds += sizeof(float) * (size_t)a.stage_rows * Kt + 16;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case GA_F16:
      return launch_loss_bwd<_Float16>(a, dloss, dA, dP_bcast, bcast_scale, lds, s);
    case GA_BF16:
      return launch_loss_bwd<bf16_t>(a, dloss, dA, dP_bcast, bcast_scale, lds, s);
    case GA_F32:
      return launch_loss_bwd<float>(a, dloss, dA, dP_bcast, bcast_scale, lds, s);
    default:
      return GA_ERR_DTYPE;
  }
}

template <typename T>
static int launch_aggregate_loss(const AggArgs& g, const LossArgs& a, int n_elem, float* A, float* terms, float* loss,
                                 unsigned* ticket, size_t lds, hipStream_t s) {
  auto k = aggregate_loss_fwd_kernel<T>;
  if (set_dyn_lds(k, lds) != GA_OK) return GA_ERR_LAUNCH;
  hipLaunchKernelGGL(k, dim3((n_elem + kThreads - 1) / kThreads), dim3(kThreads), lds, s, g, a, n_elem, A, terms, loss,
                     ticket);
  return check_launch();
}

extern "C" int ga_aggregate_loss_fwd(const void* const* maps, const int* heads, int n_maps, int res, int Kt, int first,
                                     int last, const ga_token_t* tokens, int T, const ga_loss_params_t* hp, float* A,
                                     float* terms, float* loss, unsigned* ticket, int dtype, ga_stream_t stream) {
  if (!terms || !loss || !ticket) return GA_ERR_NULL;
  AggArgs g;
  int rc = fill_agg_args(g, maps, heads, n_maps);
  if (rc != GA_OK) return rc;
  LossArgs a;
  rc = fill_args(a, A, res, Kt, first, last, tokens, T, hp);
  if (rc != GA_OK) return rc;
  size_t lds = fwd_lds(res * res, a.strict);
  if (lds > kLdsBudget) return GA_ERR_SHAPE;
  a.use_gcol = lds + sizeof(float) * (size_t)T * res * res <= kLdsBudget / 2 ? 1 : 0;   // the columns first, the staging area with what is left
  if (a.use_gcol) lds += sizeof(float) * (size_t)T * res * res;
  a.stage_rows = choose_stage_rows(lds, res * res, Kt, A);
  lds += sizeof(float) * (size_t)a.stage_rows * Kt + 16;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int n_elem = res * res * Kt;
  switch (dtype) {
    case GA_F16:
      return launch_aggregate_loss<_Float16>(g, a, n_elem, A, terms, loss, ticket, lds, s);
    case GA_BF16:
      return launch_aggregate_loss<bf16_t>(g, a, n_elem, A, terms, loss, ticket, lds, s);
    case GA_F32:
      return launch_aggregate_loss<float>(g, a, n_elem, A, terms, loss, ticket, lds, s);
    default:
      return GA_ERR_DTYPE;
  }
}
