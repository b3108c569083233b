// K1: attention-store capture for short key sequences (the 77-token text context).
//
//   forward : P = softmax_k(scale * Q K^T),  O = P V,  P optionally written once in the
//             reference's (B*H, N, Kt) layout (utils/ptp_utils.py:77-86,97-146)
//   backward: dQ = scale * [P o (dP - rowsum(dP o P))] K   with dP = dO V^T (+ direct dP)
//
// gfx950 mapping.  One wave owns 16 query rows of one (batch, head); a workgroup of WAVES waves
// shares the head's K and V, staged once in LDS (77 x D fits for every layer: <= 27 KB each).
// The score tile is computed TRANSPOSED, S^T = K Q^T, with v_mfma_f32_16x16x16 (f16/bf16) or the
// exact-f32 v_mfma_f32_16x16x4: keys land on the accumulator rows, so
//   * a lane holds NT*4 keys of ONE query: the row softmax is in-lane plus two cross-lane steps;
//   * the accumulator layout of P^T (and dS^T) IS the B-operand layout of the next MFMA
//     (O^T = V^T P^T, dQ^T = K^T dS^T): no LDS round trip between the two contractions.
// Q, dO are read straight from the projection layout [B][N][H][D] (no head transpose copies) and
// O, dQ are written back in it; P leaves through an LDS staging tile as 16-byte coalesced stores.
// The kernels are HBM/L2-bound (about 26 FLOP/B), so MFMA shape efficiency is not the concern.
#include "attn_common.h"

using namespace ga;

namespace {

constexpr int kNT = 8;  // key tiles of 16 -> Kt <= 128

template <typename T, int NT>
using Lds = TileLds<T, NT * 16>;

// Stage one (batch, head) slice [key][D] of a [B][Kt][H][D] projection into LDS as a row-major image and/or a
// transposed image; padding keys / columns are zero-filled.  All global loads of a pass are issued before the
// first LDS store (one L2 round trip per pass instead of one per vector); up to two sources share a pass.
// The "column image" (tr*) is the one whose ROWS are the contraction index of the second product (V in P.V, K in
// dS.K): a transposed [d][key] image for f32, a row-major image with row stride trs<NK>() read through
// ds_read_b64_tr_b16 for the 16-bit types (attn_common.h) — no 2-byte scatter stores.
template <typename T, int NT, int NK>
__host__ __device__ constexpr int col_image_elems() {
  return kTrRead<T> ? NT * 16 * trs<NK>() : Lds<T, NT>::tr_size(NK * 16);
}

template <typename T, int NT, int NK, int NTHREADS>
__device__ __forceinline__ void stage_kv2(const T* __restrict__ srcA, T* rowA, T* trA, const T* __restrict__ srcB,
                                          T* rowB, T* trB, int H, int Kt, int D) {
  constexpr int VEC = Lds<T, NT>::VEC;
  constexpr int KP = Lds<T, NT>::KP;
  constexpr int UNROLL = 4;
  constexpr int DP = NK * 16;
  constexpr int KS = DP + VEC;
  constexpr int vpr = DP / VEC;  // vectors per row
  constexpr int total = KP * vpr;
  for (int base = 0; base < total; base += NTHREADS * UNROLL) {
    uint4 a[UNROLL], b[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int idx = base + u * NTHREADS + threadIdx.x;
      const int key = idx / vpr;
      const int d = (idx - key * vpr) * VEC;
      const bool live = idx < total && key < Kt && d < D;
      a[u] = uint4{0, 0, 0, 0};
      b[u] = uint4{0, 0, 0, 0};
      if (live) {
        const size_t off = (size_t)key * H * D + d;
        a[u] = *reinterpret_cast<const uint4*>(srcA + off);
        if (srcB) b[u] = *reinterpret_cast<const uint4*>(srcB + off);
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int idx = base + u * NTHREADS + threadIdx.x;
      if (idx >= total) continue;
      const int key = idx / vpr;
      const int d = (idx - key * vpr) * VEC;
      if (rowA) *reinterpret_cast<uint4*>(rowA + key * KS + d) = a[u];
      if (rowB) *reinterpret_cast<uint4*>(rowB + key * KS + d) = b[u];
      if constexpr (kTrRead<T>) {
        if (trA) *reinterpret_cast<uint4*>(trA + key * trs<NK>() + d) = a[u];
        if (trB) *reinterpret_cast<uint4*>(trB + key * trs<NK>() + d) = b[u];
      } else {
        const T* ea = reinterpret_cast<const T*>(&a[u]);
        const T* eb = reinterpret_cast<const T*>(&b[u]);
        if (trA) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) trA[Lds<T, NT>::tr(d + i) + key] = ea[i];
        }
        if (trB) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) trB[Lds<T, NT>::tr(d + i) + key] = eb[i];
        }
      }
    }
  }
}

// S^T tiles: acc[t][r] = sum_d Kimg[16t + 4g + r][d] * X[q][d]   (X = Q or dO row of this lane's query)
// k-chunks go through the MFMA in pairs (one 16x16x32 does the work of two 16x16x16 in the same cycles on gfx950);
// an odd last chunk of a 16-bit type is the same instruction with the column operand's upper half zero and the
// chunk-0 fragment as a finite filler on the row side (never a legacy 16x16x16 chained onto a 16x16x32: self_attn.hip).
template <typename T, int NT, int NK>
__device__ __forceinline__ void qk_tiles(const typename Traits<T>::frag (&x)[NK], const T* kimg, int KS, int c, int g,
                                         f32x4 (&acc)[NT]) {
  using Tr = Traits<T>;
  const typename Tr::frag z = zero_frag<T>();
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kc = 0; kc < NK; kc += 2) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const T* row = kimg + (t * 16 + c) * KS + 4 * g;
      if (kc + 1 < NK) acc[t] = Tr::mma16x2(lds_frag<T>(row + kc * 16), lds_frag<T>(row + kc * 16 + 16), x[kc], x[kc + 1], acc[t]);
      else if constexpr (sizeof(T) == 2) acc[t] = Tr::mma16x2(lds_frag<T>(row + kc * 16), lds_frag<T>(row), x[kc], z, acc[t]);
      else acc[t] = Tr::mma16(load_frag<T>(row + kc * 16), x[kc], acc[t]);
    }
  }
}

// out^T[d block dt][this lane's query] = sum over the NT key blocks of ColImg^T . F   (O^T = V^T P^T, dQ^T = K^T dS^T)
template <typename T, int NT, int NK>
__device__ __forceinline__ f32x4 keys_times_frags(const T* colimg, const typename Traits<T>::frag (&f)[NT], int dt,
                                                  int lane) {
  using Tr = Traits<T>;
  f32x4 o = {0.f, 0.f, 0.f, 0.f};
  if constexpr (kTrRead<T>) {
    constexpr int S = trs<NK>();
    const int i = lane & 15;
    const T* base = colimg + (4 * (lane >> 4) + (i >> 2)) * S + 4 * (i & 3) + dt * 16;
    const typename Tr::frag z = zero_frag<T>();
#pragma unroll
    for (int t = 0; t < NT; t += 2) {
      const typename Tr::frag a0 = tr_read<T>(base + t * 16 * S);
      if (t + 1 < NT) o = Tr::mma16x2(a0, tr_read<T>(base + (t + 1) * 16 * S), f[t], f[t + 1], o);
      else o = Tr::mma16x2(a0, a0, f[t], z, o);  // odd block count: finite filler x zero
    }
  } else {
    const int c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int t = 0; t < NT; ++t) o = Tr::mma16(load_frag<T>(colimg + Lds<T, NT>::tr(dt * 16 + c) + t * 16 + 4 * g), f[t], o);
  }
  return o;
}

// in place: acc (raw q.k) -> normalised probabilities (f32); keys >= Kt get 0
template <int NT>
__device__ __forceinline__ void softmax_keys(f32x4 (&acc)[NT], int Kt, int g, float scale) {
  const float c1 = scale * 1.4426950408889634f;
  float m = -INFINITY;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (t * 16 + 4 * g + r < Kt) m = fmaxf(m, acc[t][r]);
  m = fmaxf(m, __shfl_xor(m, 16, 64));
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float p = (t * 16 + 4 * g + r < Kt) ? __builtin_amdgcn_exp2f((acc[t][r] - m) * c1) : 0.f;  // arg <= 0
      acc[t][r] = p;
      sum += p;
    }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[t][r] *= inv;
}

// paint-with-words (utils/ptp_utils.py:113-138): scores' = scale * q.k + bias[n][key] * coef, then the same softmax.
// in place: acc (raw q.k) -> normalised probabilities of the biased scores
template <typename T, int NT>
__device__ __forceinline__ void softmax_keys_biased(f32x4 (&acc)[NT], const T* __restrict__ bias_row, float coef, int Kt,
                                                    int g, float scale, f32x4 (&bias_out)[NT]) {
  float m = -INFINITY;
  T braw[NT][4];   // all of this lane's bias entries requested together, from keys clamped into the row (see load_row_frags)
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) braw[t][r] = bias_row[min(t * 16 + 4 * g + r, Kt - 1)];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = t * 16 + 4 * g + r;
      const float bv = key < Kt ? Traits<T>::to_f32(braw[t][r]) : 0.f;
      bias_out[t][r] = bv;
      acc[t][r] = acc[t][r] * scale + bv * coef;
      if (key < Kt) m = fmaxf(m, acc[t][r]);
    }
  m = fmaxf(m, __shfl_xor(m, 16, 64));
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float p = (t * 16 + 4 * g + r < Kt) ? __builtin_amdgcn_exp2f((acc[t][r] - m) * 1.4426950408889634f) : 0.f;
      acc[t][r] = p;
      sum += p;
    }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[t][r] *= inv;
}

// monotonic map f32 -> u32 (so that an unsigned atomic max orders floats)
__device__ __forceinline__ unsigned ordered_bits(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// max over EVERY scaled score of the layer call (all batches, heads, queries, keys) and where it sits — the
// `attention_scores.max()` of utils/ptp_utils.py:134, whose value scales the paint-with-words bias and through
// which the reference's autograd also sends a gradient.  packed[0] = (ordered bits << 32) | flat index into
// [B*H][N][Kt]; the caller zeroes it first.
template <typename T, int NT, int WAVES, int NK>
__global__ __launch_bounds__(WAVES * 64) void attn_scores_max_kernel(const T* __restrict__ Q, const T* __restrict__ K,
                                                                     unsigned long long* __restrict__ packed, int H,
                                                                     int N, int Kt, int D, int DP, float scale) {
  using Tr = Traits<T>;
  constexpr int ROWS = WAVES * 16;
  const int KS = Lds<T, NT>::ks(DP);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* Ks = reinterpret_cast<T*>(smem);
  const int b = blockIdx.z, head = blockIdx.y, q_wg = blockIdx.x * ROWS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const size_t kv_off = ((size_t)b * Kt * H + head) * D;
  const int q = q_wg + wave * 16 + c;
  const bool ok = q < N;
  const size_t row_off = (((size_t)b * N + (ok ? q : 0)) * H + head) * D;
  typename Tr::frag xq[NK];
  load_row_frags<T, NK>(Q + row_off, ok, D, g, xq);
  stage_kv2<T, NT, NK, WAVES * 64>(K + kv_off, Ks, nullptr, nullptr, nullptr, nullptr, H, Kt, D);
  __syncthreads();
  f32x4 acc[NT];
  qk_tiles<T, NT, NK>(xq, Ks, KS, c, g, acc);
  unsigned long long best = 0;
  if (ok) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = t * 16 + 4 * g + r;
        if (key < Kt) {
          // the reference's scores are rounded to the activation type by baddbmm: so is the value compared here
          const float sv = Tr::to_f32(Tr::from_f32(acc[t][r] * scale));
          const unsigned long long idx = ((unsigned long long)(b * H + head) * N + q) * Kt + key;
          const unsigned long long cand = ((unsigned long long)ordered_bits(sv) << 32) | (idx & 0xffffffffull);
          best = cand > best ? cand : best;
        }
      }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long other = __shfl_xor(best, o, 64);
    best = other > best ? other : best;
  }
  if (lane == 0 && best != 0) atomicMax(packed, best);
}

template <typename T, int NT, int WAVES, int NK, bool BIAS = false>
__global__ __launch_bounds__(WAVES * 64) void attn_capture_fwd_kernel(const T* __restrict__ Q, const T* __restrict__ K,
                                                                      const T* __restrict__ V, T* __restrict__ O,
                                                                      T* __restrict__ P, int H, int N, int Kt, int D,
                                                                      int DP, float scale,
                                                                      const T* __restrict__ bias = nullptr,
                                                                      const float* __restrict__ coef = nullptr) {
  using Tr = Traits<T>;
  constexpr int KP = Lds<T, NT>::KP;
  constexpr int ROWS = WAVES * 16;
  const int KS = Lds<T, NT>::ks(DP);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* Ks = reinterpret_cast<T*>(smem);  // [KP][KS]
  T* Vt = Ks + KP * KS;                // V column image
  T* Pst = Vt + col_image_elems<T, NT, NK>();  // [ROWS][Kt] (only when P != nullptr)

  const int b = blockIdx.z, head = blockIdx.y, q_wg = blockIdx.x * ROWS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const size_t kv_off = ((size_t)b * Kt * H + head) * D;
  const int q = q_wg + wave * 16 + c;
  const bool ok = q < N;
  const size_t row_off = (((size_t)b * N + (ok ? q : 0)) * H + head) * D;
  typename Tr::frag xq[NK];
  load_row_frags<T, NK>(Q + row_off, ok, D, g, xq);  // in flight while K/V are staged
  stage_kv2<T, NT, NK, WAVES * 64>(K + kv_off, Ks, nullptr, V + kv_off, nullptr, Vt, H, Kt, D);
  __syncthreads();

  f32x4 acc[NT];
  qk_tiles<T, NT, NK>(xq, Ks, KS, c, g, acc);
  if constexpr (BIAS) {
    f32x4 unused[NT];
    softmax_keys_biased<T, NT>(acc, bias + (size_t)(ok ? q : 0) * Kt, coef[0], Kt, g, scale, unused);
  } else {
    softmax_keys<NT>(acc, Kt, g, scale);
  }

  typename Tr::frag pf[NT];  // P^T as the B operand of O^T = V^T P^T (and the value stored in P)
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) pf[t][r] = Tr::from_f32(acc[t][r]);

  if (P != nullptr) {
    T* prow = Pst + (wave * 16 + c) * Kt;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = t * 16 + 4 * g + r;
        if (key < Kt) prow[key] = pf[t][r];
      }
  }

  T* orow = O + row_off;
#pragma unroll
  for (int dt = 0; dt < NK; ++dt) {
    const f32x4 o = keys_times_frags<T, NT, NK>(Vt, pf, dt, lane);
    const int d = (dt << 4) + (g << 2);  // o[r] = O[q][d + r]
    if (ok && d < D) {
      typename Tr::frag of;
#pragma unroll
      for (int r = 0; r < 4; ++r) of[r] = Tr::from_f32(o[r]);
      store_frag<T>(orow + d, of);
    }
  }

  if (P != nullptr) {  // uniform branch: P is a kernel argument
    __syncthreads();
    const int rows = min(ROWS, N - q_wg);
    const int n = rows * Kt;
    T* dst = P + ((size_t)(b * H + head) * N + q_wg) * Kt;
    constexpr int VEC = 16 / sizeof(T);
    int done = 0;
    if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
      const int nv = n / VEC;
      const uint4* s4 = reinterpret_cast<const uint4*>(Pst);
      uint4* d4 = reinterpret_cast<uint4*>(dst);
      for (int i = threadIdx.x; i < nv; i += WAVES * 64) d4[i] = s4[i];
      done = nv * VEC;
    }
    for (int i = done + threadIdx.x; i < n; i += WAVES * 64) dst[i] = Pst[i];
  }
}

template <typename T, int NT, int WAVES, int NK, bool BIAS = false>
__global__ __launch_bounds__(WAVES * 64) void attn_capture_bwd_kernel(
    const T* __restrict__ Q, const T* __restrict__ K, const T* __restrict__ V, const T* __restrict__ dO,
    const T* __restrict__ dP, long long dP_sb, long long dP_sn, T* __restrict__ dQ, int H, int N, int Kt, int D, int DP,
    float scale, const T* __restrict__ bias = nullptr, const float* __restrict__ coef = nullptr,
    float* __restrict__ bias_grad = nullptr) {
  using Tr = Traits<T>;
  constexpr int KP = Lds<T, NT>::KP;
  constexpr int ROWS = WAVES * 16;
  const int KS = Lds<T, NT>::ks(DP);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* Ks = reinterpret_cast<T*>(smem);  // [KP][KS]   K row-major  (scores)
  T* Vs = Ks + KP * KS;                // [KP][KS]   V row-major  (dP = dO V^T)
  T* Ktr = Vs + KP * KS;               // K column image (dQ = dS K)

  const int b = blockIdx.z, head = blockIdx.y, q_wg = blockIdx.x * ROWS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const size_t kv_off = ((size_t)b * Kt * H + head) * D;
  const int q = q_wg + wave * 16 + c;
  const bool ok = q < N;
  const size_t row_off = (((size_t)b * N + (ok ? q : 0)) * H + head) * D;
  typename Tr::frag xq[NK], xdo[NK];
  load_row_frags<T, NK>(Q + row_off, ok, D, g, xq);
  load_row_frags<T, NK>(dO + row_off, ok, D, g, xdo);
  // the gradient that arrives through the stored map (the loss's dA), this lane's 4 NT probabilities: requested here, with
  // the other operands, from clamped addresses (without one: from K, and dropped below).  Loaded where it is added — one
  // 2-byte load per key under `if (key < Kt)`, each waited for on its own — it was 4 NT dependent round trips in the
  // middle of the kernel.
  T up[NT][4];
  {
    const T* src = dP != nullptr ? dP + (long long)(b * H + head) * dP_sb + (long long)(ok ? q : 0) * dP_sn : K + kv_off;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) up[t][r] = src[min(t * 16 + 4 * g + r, Kt - 1)];
  }
  stage_kv2<T, NT, NK, WAVES * 64>(K + kv_off, Ks, Ktr, V + kv_off, Vs, nullptr, H, Kt, D);
  __syncthreads();

  f32x4 p[NT], dp[NT], bv[NT];
  qk_tiles<T, NT, NK>(xq, Ks, KS, c, g, p);
  if constexpr (BIAS) softmax_keys_biased<T, NT>(p, bias + (size_t)(ok ? q : 0) * Kt, coef[0], Kt, g, scale, bv);
  else softmax_keys<NT>(p, Kt, g, scale);      // identical instruction sequence to the forward
  qk_tiles<T, NT, NK>(xdo, Vs, KS, c, g, dp);  // dP^T[key][q] = sum_d V[key][d] dO[q][d]

  {
    const bool use = dP != nullptr && ok;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) dp[t][r] += (use && t * 16 + 4 * g + r < Kt) ? Tr::to_f32(up[t][r]) : 0.f;
  }
  float dot = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) dot += dp[t][r] * p[t][r];
  dot += __shfl_xor(dot, 16, 64);
  dot += __shfl_xor(dot, 32, 64);
  float amax = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float ds = p[t][r] * (dp[t][r] - dot);
      p[t][r] = ds;
      amax = fmaxf(amax, fabsf(ds));
    }
  if constexpr (BIAS) {
    // d loss / d coef = sum over every score of dS * bias: what the reference's autograd sends on through
    // `attention_scores.max()` (utils/ptp_utils.py:134).  One float atomic per wave.
    if (bias_grad != nullptr) {
      float gsum = 0.f;
      if (ok) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) gsum += p[t][r] * bv[t][r];
      }
      gsum = wave_reduce_sum(gsum);
      if (lane == 0 && gsum != 0.f) atomicAdd(bias_grad, gsum);
    }
  }
  // dS feeds the MFMA in T: rescale each query row by a power of two so its largest |dS| sits in
  // [0.5, 1) (fp16 would otherwise put the ~1e-5 loss gradients in the subnormal range); exact undo below
  float undo = scale;
  if (sizeof(T) == 2) {
    amax = fmaxf(amax, __shfl_xor(amax, 16, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
    if (amax > 0.f && amax < INFINITY) {
      int e;
      (void)frexpf(amax, &e);
      const float sc = ldexpf(1.0f, -e);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) p[t][r] *= sc;
      undo = scale * ldexpf(1.0f, e);
    }
  }
  typename Tr::frag dsf[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) dsf[t][r] = Tr::from_f32(p[t][r]);

  T* orow = dQ + row_off;
#pragma unroll
  for (int dt = 0; dt < NK; ++dt) {
    const f32x4 o = keys_times_frags<T, NT, NK>(Ktr, dsf, dt, lane);
    const int d = (dt << 4) + (g << 2);
    if (ok && d < D) {
      typename Tr::frag of;
#pragma unroll
      for (int r = 0; r < 4; ++r) of[r] = Tr::from_f32(o[r] * undo);
      store_frag<T>(orow + d, of);
    }
  }
}

template <typename T, int NT, int NK>
size_t fwd_lds_bytes(int Kt, int waves, bool withP) {
  return sizeof(T) * ((size_t)Lds<T, NT>::KP * Lds<T, NT>::ks(NK * 16) + (size_t)col_image_elems<T, NT, NK>() +
                      (withP ? (size_t)waves * 16 * Kt : 0));
}
template <typename T, int NT, int NK>
size_t bwd_lds_bytes() {
  return sizeof(T) * (2 * (size_t)Lds<T, NT>::KP * Lds<T, NT>::ks(NK * 16) + (size_t)col_image_elems<T, NT, NK>());
}

constexpr size_t kLdsLimit = 160 * 1024;

// Dispatch on NK = ceil(D / 16): exact instantiations for the head sizes of SD-1.x (40, 80, 160), SD-2.x / SDXL
// (64) and small test sizes; anything else runs the next larger instantiation with zero-padded chunks.
template <typename T, int NT, int NK>
int launch_fwd_nk(const void* Q, const void* K, const void* V, void* O, void* P, int B, int H, int N, int Kt, int D,
                  float scale, hipStream_t s) {
  const int DP = NK * 16;
  const size_t lds = fwd_lds_bytes<T, NT, NK>(Kt, 4, P != nullptr);
  if (lds > kLdsLimit) return GA_ERR_SHAPE;
  dim3 grid((N + 63) / 64, H, B);
  auto k = attn_capture_fwd_kernel<T, NT, 4, NK>;
  int rc = set_dyn_lds(k, lds);
  if (rc != GA_OK) return rc;
  hipLaunchKernelGGL(k, grid, dim3(256), lds, s, (const T*)Q, (const T*)K, (const T*)V, (T*)O, (T*)P, H, N, Kt, D, DP,
                     scale, (const T*)nullptr, (const float*)nullptr);
  return check_launch();
}

template <typename T, int NT, int NK>
int launch_bwd_nk(const void* Q, const void* K, const void* V, const void* dO, const void* dP, int64_t sb, int64_t sn,
                  void* dQ, int B, int H, int N, int Kt, int D, float scale, hipStream_t s) {
  const int DP = NK * 16;
  const size_t lds = bwd_lds_bytes<T, NT, NK>();
  if (lds > kLdsLimit) return GA_ERR_SHAPE;
  dim3 grid((N + 63) / 64, H, B);
  auto k = attn_capture_bwd_kernel<T, NT, 4, NK>;
  int rc = set_dyn_lds(k, lds);
  if (rc != GA_OK) return rc;
  hipLaunchKernelGGL(k, grid, dim3(256), lds, s, (const T*)Q, (const T*)K, (const T*)V, (const T*)dO, (const T*)dP,
                     (long long)sb, (long long)sn, (T*)dQ, H, N, Kt, D, DP, scale, (const T*)nullptr,
                     (const float*)nullptr, (float*)nullptr);
  return check_launch();
}

template <typename T, int NK>
int launch_max_nk(const void* Q, const void* K, unsigned long long* packed, int B, int H, int N, int Kt, int D, float scale,
                  hipStream_t s) {
  constexpr int NT = 5;
  const size_t lds = sizeof(T) * (size_t)Lds<T, NT>::KP * Lds<T, NT>::ks(NK * 16);
  if (lds > kLdsLimit) return GA_ERR_SHAPE;
  auto k = attn_scores_max_kernel<T, NT, 4, NK>;
  int rc = set_dyn_lds(k, lds);
  if (rc != GA_OK) return rc;
  hipLaunchKernelGGL(k, dim3((N + 63) / 64, H, B), dim3(256), lds, s, (const T*)Q, (const T*)K, packed, H, N, Kt, D,
                     NK * 16, scale);
  return check_launch();
}

template <typename T, int NK>
int launch_fwd_biased_nk(const void* Q, const void* K, const void* V, void* O, void* P, const void* bias, const float* coef,
                         int B, int H, int N, int Kt, int D, float scale, hipStream_t s) {
  constexpr int NT = 5;
  const size_t lds = fwd_lds_bytes<T, NT, NK>(Kt, 4, P != nullptr);
  if (lds > kLdsLimit) return GA_ERR_SHAPE;
  auto k = attn_capture_fwd_kernel<T, NT, 4, NK, true>;
  int rc = set_dyn_lds(k, lds);
  if (rc != GA_OK) return rc;
  hipLaunchKernelGGL(k, dim3((N + 63) / 64, H, B), dim3(256), lds, s, (const T*)Q, (const T*)K, (const T*)V, (T*)O, (T*)P,
                     H, N, Kt, D, NK * 16, scale, (const T*)bias, coef);
  return check_launch();
}

template <typename T, int NK>
int launch_bwd_biased_nk(const void* Q, const void* K, const void* V, const void* dO, const void* dP, int64_t sb,
                         int64_t sn, void* dQ, const void* bias, const float* coef, float* bias_grad, int B, int H, int N,
                         int Kt, int D, float scale, hipStream_t s) {
  constexpr int NT = 5;
  const size_t lds = bwd_lds_bytes<T, NT, NK>();
  if (lds > kLdsLimit) return GA_ERR_SHAPE;
  auto k = attn_capture_bwd_kernel<T, NT, 4, NK, true>;
  int rc = set_dyn_lds(k, lds);
  if (rc != GA_OK) return rc;
  hipLaunchKernelGGL(k, dim3((N + 63) / 64, H, B), dim3(256), lds, s, (const T*)Q, (const T*)K, (const T*)V,
                     (const T*)dO, (const T*)dP, (long long)sb, (long long)sn, (T*)dQ, H, N, Kt, D, NK * 16, scale,
                     (const T*)bias, coef, bias_grad);
  return check_launch();
}

#define GA_NK_DISPATCH_COARSE(CALL)          \
  do {                                       \
    const int nk = (D + 15) / 16;            \
    if (nk <= 2) return CALL(2);             \
    if (nk <= 4) return CALL(4);             \
    if (nk <= 8) return CALL(8);             \
    return CALL(16);                         \
  } while (0)

#define GA_NK_DISPATCH(CALL)                 \
  do {                                       \
    const int nk = (D + 15) / 16;            \
    if (nk <= 1) return CALL(1);             \
    if (nk == 2) return CALL(2);             \
    if (nk == 3) return CALL(3);             \
    if (nk == 4) return CALL(4);             \
    if (nk == 5) return CALL(5);             \
    if (nk <= 8) return CALL(8);             \
    if (nk <= 10) return CALL(10);           \
    return CALL(16);                         \
  } while (0)

template <typename T, int NT>
int launch_fwd(const void* Q, const void* K, const void* V, void* O, void* P, int B, int H, int N, int Kt, int D,
               float scale, hipStream_t s) {
#define GA_CALL(NKV) launch_fwd_nk<T, NT, NKV>(Q, K, V, O, P, B, H, N, Kt, D, scale, s)
  if (NT != 5) GA_NK_DISPATCH_COARSE(GA_CALL);  // 81..128 keys: fewer instantiations
  GA_NK_DISPATCH(GA_CALL);
#undef GA_CALL
}

template <typename T, int NT>
int launch_bwd(const void* Q, const void* K, const void* V, const void* dO, const void* dP, int64_t sb, int64_t sn,
               void* dQ, int B, int H, int N, int Kt, int D, float scale, hipStream_t s) {
#define GA_CALL(NKV) launch_bwd_nk<T, NT, NKV>(Q, K, V, dO, dP, sb, sn, dQ, B, H, N, Kt, D, scale, s)
  if (NT != 5) GA_NK_DISPATCH_COARSE(GA_CALL);
  GA_NK_DISPATCH(GA_CALL);
#undef GA_CALL
}

int check_common(int B, int H, int N, int Kt, int D) {
  if (B <= 0 || H <= 0 || N <= 0 || Kt <= 0 || D <= 0) return GA_ERR_SHAPE;
  if (Kt > kNT * 16 || D > 256 || B > 65535 || H > 65535) return GA_ERR_SHAPE;
  if (D % 8 != 0) return GA_ERR_ALIGN;
  return GA_OK;
}
bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int ga_attn_capture_fwd(const void* Q, const void* K, const void* V, void* O, void* P, int B, int H, int N,
                                   int Kt, int D, float scale, int dtype, ga_stream_t stream) {
  if (!Q || !K || !V || !O) return GA_ERR_NULL;
  int rc = check_common(B, H, N, Kt, D);
  if (rc != GA_OK) return rc;
  if (!aligned16(Q) || !aligned16(K) || !aligned16(V) || !aligned16(O)) return GA_ERR_ALIGN;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool small = Kt <= 80;  // the 77-token case: 5 key tiles instead of 8
  switch (dtype) {
    case GA_F16:
      return small ? launch_fwd<_Float16, 5>(Q, K, V, O, P, B, H, N, Kt, D, scale, s)
                   : launch_fwd<_Float16, kNT>(Q, K, V, O, P, B, H, N, Kt, D, scale, s);
    case GA_BF16:
      return small ? launch_fwd<bf16_t, 5>(Q, K, V, O, P, B, H, N, Kt, D, scale, s)
                   : launch_fwd<bf16_t, kNT>(Q, K, V, O, P, B, H, N, Kt, D, scale, s);
    case GA_F32:
      return small ? launch_fwd<float, 5>(Q, K, V, O, P, B, H, N, Kt, D, scale, s)
                   : launch_fwd<float, kNT>(Q, K, V, O, P, B, H, N, Kt, D, scale, s);
    default:
      return GA_ERR_DTYPE;
  }
}

extern "C" int ga_attn_capture_bwd(const void* Q, const void* K, const void* V, const void* dO, const void* dP,
                                   int64_t dP_stride_bh, int64_t dP_stride_n, void* dQ, void* dK, void* dV, int B,
                                   int H, int N, int Kt, int D, float scale, int dtype, ga_stream_t stream) {
  if (!Q || !K || !V || !dO || !dQ) return GA_ERR_NULL;
  if (dK != nullptr || dV != nullptr) return GA_ERR_UNSUPPORTED;
  int rc = check_common(B, H, N, Kt, D);
  if (rc != GA_OK) return rc;
  if (!aligned16(Q) || !aligned16(K) || !aligned16(V) || !aligned16(dO) || !aligned16(dQ)) return GA_ERR_ALIGN;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool small = Kt <= 80;
  switch (dtype) {
    case GA_F16:
      return small ? launch_bwd<_Float16, 5>(Q, K, V, dO, dP, dP_stride_bh, dP_stride_n, dQ, B, H, N, Kt, D, scale, s)
                   : launch_bwd<_Float16, kNT>(Q, K, V, dO, dP, dP_stride_bh, dP_stride_n, dQ, B, H, N, Kt, D, scale, s);
    case GA_BF16:
      return small ? launch_bwd<bf16_t, 5>(Q, K, V, dO, dP, dP_stride_bh, dP_stride_n, dQ, B, H, N, Kt, D, scale, s)
                   : launch_bwd<bf16_t, kNT>(Q, K, V, dO, dP, dP_stride_bh, dP_stride_n, dQ, B, H, N, Kt, D, scale, s);
    case GA_F32:
      return small ? launch_bwd<float, 5>(Q, K, V, dO, dP, dP_stride_bh, dP_stride_n, dQ, B, H, N, Kt, D, scale, s)
                   : launch_bwd<float, kNT>(Q, K, V, dO, dP, dP_stride_bh, dP_stride_n, dQ, B, H, N, Kt, D, scale, s);
    default:
      return GA_ERR_DTYPE;
  }
}

// ---- paint-with-words entry points (utils/ptp_utils.py:113-138; off by default in the reference) ---------------------
namespace {
int check_biased(int B, int H, int N, int Kt, int D) {
  int rc = check_common(B, H, N, Kt, D);
  if (rc != GA_OK) return rc;
  if (Kt > 80) return GA_ERR_UNSUPPORTED;  // the reference applies the mask to the 77-token text context only
  if ((unsigned long long)B * H * N * Kt > 0xffffffffull) return GA_ERR_SHAPE;
  return GA_OK;
}
}  // namespace

extern "C" int ga_attn_scores_max(const void* Q, const void* K, int B, int H, int N, int Kt, int D, float scale, int dtype,
                                  unsigned long long* packed, ga_stream_t stream) {
  if (!Q || !K || !packed) return GA_ERR_NULL;
  int rc = check_biased(B, H, N, Kt, D);
  if (rc != GA_OK) return rc;
  if (!aligned16(Q) || !aligned16(K)) return GA_ERR_ALIGN;
  hipStream_t s = static_cast<hipStream_t>(stream);
#define GA_CALL(NKV) launch_max_nk<T_, NKV>(Q, K, packed, B, H, N, Kt, D, scale, s)
  switch (dtype) {
    case GA_F16: { using T_ = _Float16; GA_NK_DISPATCH_COARSE(GA_CALL); }
    case GA_BF16: { using T_ = bf16_t; GA_NK_DISPATCH_COARSE(GA_CALL); }
    case GA_F32: { using T_ = float; GA_NK_DISPATCH_COARSE(GA_CALL); }
    default: return GA_ERR_DTYPE;
  }
#undef GA_CALL
}

extern "C" int ga_attn_capture_fwd_biased(const void* Q, const void* K, const void* V, void* O, void* P, const void* bias,
                                          const float* coef, int B, int H, int N, int Kt, int D, float scale, int dtype,
                                          ga_stream_t stream) {
  if (!Q || !K || !V || !O || !bias || !coef) return GA_ERR_NULL;
  int rc = check_biased(B, H, N, Kt, D);
  if (rc != GA_OK) return rc;
  if (!aligned16(Q) || !aligned16(K) || !aligned16(V) || !aligned16(O)) return GA_ERR_ALIGN;
  hipStream_t s = static_cast<hipStream_t>(stream);
#define GA_CALL(NKV) launch_fwd_biased_nk<T_, NKV>(Q, K, V, O, P, bias, coef, B, H, N, Kt, D, scale, s)
  switch (dtype) {
    case GA_F16: { using T_ = _Float16; GA_NK_DISPATCH_COARSE(GA_CALL); }
    case GA_BF16: { using T_ = bf16_t; GA_NK_DISPATCH_COARSE(GA_CALL); }
    case GA_F32: { using T_ = float; GA_NK_DISPATCH_COARSE(GA_CALL); }
    default: return GA_ERR_DTYPE;
  }
#undef GA_CALL
}

extern "C" int ga_attn_capture_bwd_biased(const void* Q, const void* K, const void* V, const void* dO, const void* dP,
                                          int64_t dP_stride_bh, int64_t dP_stride_n, void* dQ, const void* bias,
                                          const float* coef, float* bias_grad, int B, int H, int N, int Kt, int D,
                                          float scale, int dtype, ga_stream_t stream) {
  if (!Q || !K || !V || !dO || !dQ || !bias || !coef) return GA_ERR_NULL;
  int rc = check_biased(B, H, N, Kt, D);
  if (rc != GA_OK) return rc;
  if (!aligned16(Q) || !aligned16(K) || !aligned16(V) || !aligned16(dO) || !aligned16(dQ)) return GA_ERR_ALIGN;
  hipStream_t s = static_cast<hipStream_t>(stream);
#define GA_CALL(NKV) \
  launch_bwd_biased_nk<T_, NKV>(Q, K, V, dO, dP, dP_stride_bh, dP_stride_n, dQ, bias, coef, bias_grad, B, H, N, Kt, D, scale, s)
  switch (dtype) {
    case GA_F16: { using T_ = _Float16; GA_NK_DISPATCH_COARSE(GA_CALL); }
    case GA_BF16: { using T_ = bf16_t; GA_NK_DISPATCH_COARSE(GA_CALL); }
    case GA_F32: { using T_ = float; GA_NK_DISPATCH_COARSE(GA_CALL); }
    default: return GA_ERR_DTYPE;
  }
#undef GA_CALL
}
