// Shared device helpers for the gfx950 kernels of libga_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ga_hip.h"

namespace ga {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct bf16_t {
  uint16_t bits;
};

__device__ __forceinline__ float bf16_to_f32(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
  __bf16 h = (__bf16)f;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN
  return __builtin_bit_cast(uint16_t, h);
}

// Per-dtype traits.  `frag` = 4 consecutive k-elements of an MFMA operand held by one lane.
template <typename T>
struct Traits;

template <>
struct Traits<_Float16> {
  using elem = _Float16;
  using frag = f16x4;
  static constexpr int kDtype = GA_F16;
  __device__ static __forceinline__ float to_f32(elem x) { return (float)x; }
  __device__ static __forceinline__ elem from_f32(float x) { return (elem)x; }
  __device__ static __forceinline__ elem zero() { return (elem)0.0f; }
  // acc += A(16 x 16k) * B(16k x 16); lane (c = lane&15, g = lane>>4) supplies k = 4g..4g+3 of both
  __device__ static __forceinline__ f32x4 mma16(frag a, frag b, f32x4 acc) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, acc, 0, 0, 0);
  }
  // two k-chunks in one v_mfma_f32_16x16x32_f16 (same cycles as ONE 16x16x16 on gfx950, measured): lane g's
  // eight k-values are chunk 0's 4g..4g+3 followed by chunk 1's 4g..4g+3 — any k order works as long as A and B agree
  __device__ static __forceinline__ f32x4 mma16x2(frag a0, frag a1, frag b0, frag b1, f32x4 acc) {
    const f16x8 a = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
    const f16x8 b = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
  }
};

template <>
struct Traits<bf16_t> {
  using elem = bf16_t;
  // Four bf16 bit patterns in ONE vector register pair.  (A struct holding `bf16_t v[4]` is an aggregate the compiler
  // has to scalarise: in the self-attention kernels with two query blocks per wave the 8-fragment probability array
  // stayed a 64-byte private-segment array, i.e. scratch memory traffic in the inner loop.)  Elements are reached
  // through a proxy, so `f[i] = from_f32(x)` / `to_f32(f[i])` read as for the other types.
  struct frag {
    s16x4 v;
    struct ref {
      s16x4& v;
      int i;
      __device__ __forceinline__ operator bf16_t() const { return bf16_t{(uint16_t)v[i]}; }
      __device__ __forceinline__ ref& operator=(bf16_t x) {
        v[i] = (short)x.bits;
        return *this;
      }
    };
    __device__ __forceinline__ ref operator[](int i) { return ref{v, i}; }
    __device__ __forceinline__ bf16_t operator[](int i) const { return bf16_t{(uint16_t)v[i]}; }
  };
  static constexpr int kDtype = GA_BF16;
  __device__ static __forceinline__ float to_f32(elem x) { return bf16_to_f32(x.bits); }
  __device__ static __forceinline__ elem from_f32(float x) { return elem{f32_to_bf16(x)}; }
  __device__ static __forceinline__ elem zero() { return elem{0}; }
  __device__ static __forceinline__ f32x4 mma16(frag a, frag b, f32x4 acc) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.v, b.v, acc, 0, 0, 0);
  }
  __device__ static __forceinline__ f32x4 mma16x2(frag a0, frag a1, frag b0, frag b1, f32x4 acc) {
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 a = __builtin_shufflevector(a0.v, a1.v, 0, 1, 2, 3, 4, 5, 6, 7);
    const s16x8 b = __builtin_shufflevector(b0.v, b1.v, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
  }
};

template <>
struct Traits<float> {
  using elem = float;
  using frag = f32x4;
  static constexpr int kDtype = GA_F32;
  __device__ static __forceinline__ float to_f32(elem x) { return x; }
  __device__ static __forceinline__ elem from_f32(float x) { return x; }
  __device__ static __forceinline__ elem zero() { return 0.0f; }
  // exact-f32 MFMA (v_mfma_f32_16x16x4_f32): step j contracts k = {4g + j}, g = 0..3
  __device__ static __forceinline__ f32x4 mma16(frag a, frag b, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc, 0, 0, 0);
    return acc;
  }
  __device__ static __forceinline__ f32x4 mma16x2(frag a0, frag a1, frag b0, frag b1, f32x4 acc) {
    return mma16(a1, b1, mma16(a0, b0, acc));
  }
};

template <typename T>
__device__ __forceinline__ typename Traits<T>::frag zero_frag() {
  typename Traits<T>::frag f;
#pragma unroll
  for (int i = 0; i < 4; ++i) f[i] = Traits<T>::zero();
  return f;
}

// load 4 consecutive elements (8 B for 16-bit types, 16 B for f32); p must be 8/16-byte aligned
template <typename T>
__device__ __forceinline__ typename Traits<T>::frag load_frag(const T* p) {
  return *reinterpret_cast<const typename Traits<T>::frag*>(p);
}
template <typename T>
__device__ __forceinline__ void store_frag(T* p, typename Traits<T>::frag f) {
  *reinterpret_cast<typename Traits<T>::frag*>(p) = f;
}

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Cross-lane sums on the VALU (DPP), not through the LDS crossbar.  `__shfl_xor(x, o, 64)` compiles to ds_bpermute_b32 + a wait:
// the three-step row sum behind every output vector of linear_kernel's epilogue was 24 dependent LDS round trips per thread (128 x
// 64 tile with row partial sums: 1.8 us of a 6.4 us workgroup, tools/micro/lin_stamps.py); the same adds as DPP operands cost a
// few cycles.  group_sum<W>: every lane gets the sum over its aligned group of W adjacent lanes (W = 2 ... 16, inside a row of
// 16); wave_sum: the sum over the 64 lanes in every lane (rows combined by row_bcast, read back from lane 63).
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_src(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROW_MASK, 0xf, false));
}
template <int W>
__device__ __forceinline__ float group_sum(float x) {
  static_assert(W == 1 || W == 2 || W == 4 || W == 8 || W == 16, "aligned groups inside a DPP row");
  if constexpr (W >= 2) x += dpp_src<0xB1>(x);    // quad_perm [1, 0, 3, 2]
  if constexpr (W >= 4) x += dpp_src<0x4E>(x);    // quad_perm [2, 3, 0, 1]
  if constexpr (W >= 8) x += dpp_src<0x141>(x);   // row_half_mirror: lane i <- lane 7 - i of its half row (the other quad's sum)
  if constexpr (W >= 16) x += dpp_src<0x140>(x);  // row_mirror: lane i <- lane 15 - i (the other half row's sum)
  return x;
}
__device__ __forceinline__ float wave_sum(float x) {
  x = group_sum<16>(x);
  x += dpp_src<0x142, 0xa>(x);                     // row_bcast:15 into rows 1 and 3
  x += dpp_src<0x143, 0xc>(x);                     // row_bcast:31 into rows 2 and 3: row 3 holds the total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}

inline int check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? GA_OK : GA_ERR_LAUNCH;
}

// x / d for a divisor known to the host: one 32-bit multiply-high, m = floor(2^32 / d) + 1, exact while x d < 2^32.  The host
// checks that against the largest dividend (make_fastdiv -> ok = false otherwise, and launch_tile then keeps the launch off
// the kernels that rely on it); d = 1 is m = 0 and returns x.
struct FastDiv {
  unsigned d, m;
};
inline FastDiv make_fastdiv(long long d, unsigned long long max_x, bool& ok) {
  FastDiv f{(unsigned)(d > 0 ? d : 1), 0u};
  if (d > 1) {
    if (max_x * (unsigned long long)d < (1ull << 32)) f.m = (unsigned)((1ull << 32) / (unsigned long long)d) + 1u;
    else ok = false;
  }
  return f;
}
__device__ __forceinline__ int fdiv(int x, FastDiv f) {   // x >= 0; the host vouched for the divisor (launch_tile: fast_ok)
  return f.m != 0u ? (int)__umulhi((unsigned)x, f.m) : x;
}
__device__ __forceinline__ int sdiv(int x, FastDiv f) {   // the same for the kernels that also take the shapes it cannot serve
  return f.m != 0u ? (int)__umulhi((unsigned)x, f.m) : (int)((unsigned)x / f.d);
}

// Holds `v` in its registers up to this point of the instruction stream.  Used after the sc1 (write-through) buffer stores
// of the split-K slabs: the compiler re-used a stored accumulator register for the NEXT store's address in the instruction
// right behind `buffer_store_dwordx4 ... sc1` (it knows no hazard there when the store has an SGPR offset), and on the MI355X
// lanes 12-15 of each 16-lane row then stored the new value now and then (about one tile in a few hundred launches:
// linear_kernel<128, 64, 3>, round 3).  With the accumulators live until the `s_waitcnt vmcnt(0)` behind the stores nothing
// can overwrite them early; tools/store_hazard_scan.py looks for the instruction pair in the built library.
template <typename V>
__device__ __forceinline__ void keep_live(const V& v) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" ::"v"(v));
#endif
}

}  // namespace ga
