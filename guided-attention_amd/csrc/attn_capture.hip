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
#include "ga_common.h"

using namespace ga;

namespace {

constexpr int kNT = 8;  // key tiles of 16 -> Kt <= 128

__host__ __device__ constexpr int round16(int x) { return (x + 15) & ~15; }

template <typename T, int NT>
struct Lds {
  static constexpr int KP = NT * 16;  // padded key count
  static constexpr int VS = KP + 4;   // row stride of a transposed [D][key] image
  __host__ __device__ static int ks(int DP) { return DP + 4; }  // row stride of a [key][D] image
};

// Stage rows [key][D] of one (batch, head) from the [B][Kt][H][D] projection into LDS, as a row-major
// image (stride DP+4) and/or a transposed image (stride KP+4); padding keys / columns are zero-filled.
template <typename T, int NT, int NTHREADS>
__device__ __forceinline__ void stage_kv(const T* __restrict__ src, int H, int Kt, int D, int DP, T* rowmaj,
                                         T* transposed) {
  using Tr = Traits<T>;
  constexpr int KP = Lds<T, NT>::KP;
  constexpr int VS = Lds<T, NT>::VS;
  const int KS = Lds<T, NT>::ks(DP);
  const int chunks = DP >> 2;
  for (int idx = threadIdx.x; idx < KP * chunks; idx += NTHREADS) {
    const int key = idx / chunks;
    const int d = (idx - key * chunks) << 2;
    typename Tr::frag f = zero_frag<T>();
    if (key < Kt && d < D) f = load_frag<T>(src + (size_t)key * H * D + d);
    if (rowmaj) store_frag<T>(rowmaj + key * KS + d, f);
    if (transposed) {
#pragma unroll
      for (int i = 0; i < 4; ++i) transposed[(d + i) * VS + key] = f[i];
    }
  }
}

// S^T tiles: acc[t][r] = sum_d Kimg[16t + 4g + r][d] * X[q][d]   (X = Q or dO row of this lane's query)
template <typename T, int NT>
__device__ __forceinline__ void qk_tiles(const T* __restrict__ xrow, bool ok, const T* kimg, int KS, int D, int DP,
                                         int c, int g, f32x4 (&acc)[NT]) {
  using Tr = Traits<T>;
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int kc = 0; kc < (DP >> 4); ++kc) {
    const int d = (kc << 4) + (g << 2);
    typename Tr::frag b = zero_frag<T>();
    if (ok && d < D) b = load_frag<T>(xrow + d);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = Tr::mma16(load_frag<T>(kimg + (t * 16 + c) * KS + d), b, acc[t]);
  }
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
      const float p = (t * 16 + 4 * g + r < Kt) ? exp2f((acc[t][r] - m) * c1) : 0.f;
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

template <typename T, int NT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void attn_capture_fwd_kernel(const T* __restrict__ Q, const T* __restrict__ K,
                                                                      const T* __restrict__ V, T* __restrict__ O,
                                                                      T* __restrict__ P, int H, int N, int Kt, int D,
                                                                      int DP, float scale) {
  using Tr = Traits<T>;
  constexpr int KP = Lds<T, NT>::KP;
  constexpr int VS = Lds<T, NT>::VS;
  constexpr int ROWS = WAVES * 16;
  const int KS = Lds<T, NT>::ks(DP);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* Ks = reinterpret_cast<T*>(smem);  // [KP][KS]
  T* Vt = Ks + KP * KS;                // [DP][VS]
  T* Pst = Vt + DP * VS;               // [ROWS][Kt] (only when P != nullptr)

  const int b = blockIdx.z, head = blockIdx.y, q_wg = blockIdx.x * ROWS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const size_t kv_off = ((size_t)b * Kt * H + head) * D;
  stage_kv<T, NT, WAVES * 64>(K + kv_off, H, Kt, D, DP, Ks, nullptr);
  stage_kv<T, NT, WAVES * 64>(V + kv_off, H, Kt, D, DP, nullptr, Vt);
  __syncthreads();

  const int q = q_wg + wave * 16 + c;
  const bool ok = q < N;
  const size_t row_off = (((size_t)b * N + (ok ? q : 0)) * H + head) * D;
  f32x4 acc[NT];
  qk_tiles<T, NT>(Q + row_off, ok, Ks, KS, D, DP, c, g, acc);
  softmax_keys<NT>(acc, Kt, g, scale);

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
  for (int dt = 0; dt < (DP >> 4); ++dt) {
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; ++t) o = Tr::mma16(load_frag<T>(Vt + (dt * 16 + c) * VS + t * 16 + 4 * g), pf[t], o);
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

template <typename T, int NT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void attn_capture_bwd_kernel(
    const T* __restrict__ Q, const T* __restrict__ K, const T* __restrict__ V, const T* __restrict__ dO,
    const T* __restrict__ dP, long long dP_sb, long long dP_sn, T* __restrict__ dQ, int H, int N, int Kt, int D, int DP,
    float scale) {
  using Tr = Traits<T>;
  constexpr int KP = Lds<T, NT>::KP;
  constexpr int VS = Lds<T, NT>::VS;
  constexpr int ROWS = WAVES * 16;
  const int KS = Lds<T, NT>::ks(DP);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* Ks = reinterpret_cast<T*>(smem);  // [KP][KS]   K row-major  (scores)
  T* Vs = Ks + KP * KS;                // [KP][KS]   V row-major  (dP = dO V^T)
  T* Ktr = Vs + KP * KS;               // [DP][VS]   K transposed (dQ = dS K)

  const int b = blockIdx.z, head = blockIdx.y, q_wg = blockIdx.x * ROWS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const size_t kv_off = ((size_t)b * Kt * H + head) * D;
  stage_kv<T, NT, WAVES * 64>(K + kv_off, H, Kt, D, DP, Ks, Ktr);
  stage_kv<T, NT, WAVES * 64>(V + kv_off, H, Kt, D, DP, Vs, nullptr);
  __syncthreads();

  const int q = q_wg + wave * 16 + c;
  const bool ok = q < N;
  const size_t row_off = (((size_t)b * N + (ok ? q : 0)) * H + head) * D;
  f32x4 p[NT], dp[NT];
  qk_tiles<T, NT>(Q + row_off, ok, Ks, KS, D, DP, c, g, p);
  softmax_keys<NT>(p, Kt, g, scale);                          // identical instruction sequence to the forward
  qk_tiles<T, NT>(dO + row_off, ok, Vs, KS, D, DP, c, g, dp);  // dP^T[key][q] = sum_d V[key][d] dO[q][d]

  if (dP != nullptr && ok) {
    const T* src = dP + (long long)(b * H + head) * dP_sb + (long long)q * dP_sn;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = t * 16 + 4 * g + r;
        if (key < Kt) dp[t][r] += Tr::to_f32(src[key]);
      }
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
  for (int dt = 0; dt < (DP >> 4); ++dt) {
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; ++t) o = Tr::mma16(load_frag<T>(Ktr + (dt * 16 + c) * VS + t * 16 + 4 * g), dsf[t], o);
    const int d = (dt << 4) + (g << 2);
    if (ok && d < D) {
      typename Tr::frag of;
#pragma unroll
      for (int r = 0; r < 4; ++r) of[r] = Tr::from_f32(o[r] * undo);
      store_frag<T>(orow + d, of);
    }
  }
}

template <typename T, int NT>
size_t fwd_lds_bytes(int DP, int Kt, int waves, bool withP) {
  return sizeof(T) * ((size_t)Lds<T, NT>::KP * Lds<T, NT>::ks(DP) + (size_t)DP * Lds<T, NT>::VS +
                      (withP ? (size_t)waves * 16 * Kt : 0));
}
template <typename T, int NT>
size_t bwd_lds_bytes(int DP) {
  return sizeof(T) * (2 * (size_t)Lds<T, NT>::KP * Lds<T, NT>::ks(DP) + (size_t)DP * Lds<T, NT>::VS);
}

constexpr size_t kLdsLimit = 160 * 1024;

template <typename KernelT>
int set_dyn_lds(KernelT kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return GA_OK;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  return e == hipSuccess ? GA_OK : GA_ERR_LAUNCH;
}

int pick_waves(int B, int H, int N) {
  // a launch wants >= 256 workgroups (one per CU); 4-wave groups amortise the K/V staging
  const long long wg4 = (long long)((N + 63) / 64) * H * B;
  return wg4 >= 256 ? 4 : 1;
}

template <typename T, int NT>
int launch_fwd(const void* Q, const void* K, const void* V, void* O, void* P, int B, int H, int N, int Kt, int D,
               float scale, hipStream_t s) {
  const int DP = round16(D);
  const int waves = pick_waves(B, H, N);
  const size_t lds = fwd_lds_bytes<T, NT>(DP, Kt, waves, P != nullptr);
  if (lds > kLdsLimit) return GA_ERR_SHAPE;
  dim3 grid((N + waves * 16 - 1) / (waves * 16), H, B);
  int rc;
  if (waves == 4) {
    auto k = attn_capture_fwd_kernel<T, NT, 4>;
    if ((rc = set_dyn_lds(k, lds)) != GA_OK) return rc;
    hipLaunchKernelGGL(k, grid, dim3(256), lds, s, (const T*)Q, (const T*)K, (const T*)V, (T*)O, (T*)P, H, N, Kt, D, DP,
                       scale);
  } else {
    auto k = attn_capture_fwd_kernel<T, NT, 1>;
    if ((rc = set_dyn_lds(k, lds)) != GA_OK) return rc;
    hipLaunchKernelGGL(k, grid, dim3(64), lds, s, (const T*)Q, (const T*)K, (const T*)V, (T*)O, (T*)P, H, N, Kt, D, DP,
                       scale);
  }
  return check_launch();
}

template <typename T, int NT>
int launch_bwd(const void* Q, const void* K, const void* V, const void* dO, const void* dP, int64_t sb, int64_t sn,
               void* dQ, int B, int H, int N, int Kt, int D, float scale, hipStream_t s) {
  const int DP = round16(D);
  const int waves = pick_waves(B, H, N);
  const size_t lds = bwd_lds_bytes<T, NT>(DP);
  if (lds > kLdsLimit) return GA_ERR_SHAPE;
  dim3 grid((N + waves * 16 - 1) / (waves * 16), H, B);
  int rc;
  if (waves == 4) {
    auto k = attn_capture_bwd_kernel<T, NT, 4>;
    if ((rc = set_dyn_lds(k, lds)) != GA_OK) return rc;
    hipLaunchKernelGGL(k, grid, dim3(256), lds, s, (const T*)Q, (const T*)K, (const T*)V, (const T*)dO, (const T*)dP,
                       (long long)sb, (long long)sn, (T*)dQ, H, N, Kt, D, DP, scale);
  } else {
    auto k = attn_capture_bwd_kernel<T, NT, 1>;
    if ((rc = set_dyn_lds(k, lds)) != GA_OK) return rc;
    hipLaunchKernelGGL(k, grid, dim3(64), lds, s, (const T*)Q, (const T*)K, (const T*)V, (const T*)dO, (const T*)dP,
                       (long long)sb, (long long)sn, (T*)dQ, H, N, Kt, D, DP, scale);
  }
  return check_launch();
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
