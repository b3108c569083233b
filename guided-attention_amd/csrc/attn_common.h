// Building blocks shared by the attention kernels (attn_capture.hip, self_attn.hip): LDS tile layouts, operand
// fragment loads, dynamic-LDS bookkeeping.
#pragma once
#include <mutex>
#include <unordered_map>

#include "ga_common.h"

namespace ga {

__host__ __device__ constexpr int round16(int x) { return (x + 15) & ~15; }

// LDS images of a [KP keys][DP] tile.
template <typename T, int KP_>
struct TileLds {
  static constexpr int VEC = 16 / sizeof(T);  // elements per 16-byte vector
  static constexpr int KP = KP_;              // (padded) keys per tile
  // Row strides are padded by one 16-byte vector: every row stays 16-byte aligned for the staging stores,
  // and the MFMA fragment reads (8 B per lane, 16 rows x 4 k-groups) fall on 64 distinct banks.
  static constexpr int VS = KP + VEC;                             // transposed [D][key] image
  __host__ __device__ static int ks(int DP) { return DP + VEC; }  // row-major [key][D] image
  // The transposed image is written with 2-byte stores by lanes that hold consecutive 16-byte pieces of ONE key
  // row, i.e. image rows VEC apart: with a plain stride those all fall on two banks (20-way conflict, ~5 us per
  // workgroup at D = 160).  Row r is therefore shifted by 4 elements per VEC rows, which walks the banks.
  static constexpr int ROT = 4;
  __host__ __device__ static int tr(int r) { return r * VS + ROT * (r / VEC); }
  __host__ __device__ static int tr_size(int DP) { return DP * VS + ROT * (DP / VEC); }
};

// NK = number of 16-wide chunks of the (padded) head dimension, a template parameter so that the per-lane
// fragment arrays are exactly sized and statically indexed (NK = 3 / 5 / 10 for SD-1.x, 4 for SD-2.x / SDXL).

// This lane's operand fragments of one activation row X[q][:] (Q or dO), all loads issued together so that
// they overlap the K/V staging instead of forming one global round trip per 16-wide chunk.
template <typename T, int NK>
__device__ __forceinline__ void load_row_frags(const T* __restrict__ xrow, bool ok, int D, int g,
                                               typename Traits<T>::frag (&x)[NK]) {
  // `xrow` is a valid row also when !ok (the callers clamp the row index).  Every chunk is loaded, from a column clamped
  // into the row, and dropped by a select where it lies past the head size or the row does not exist: under
  // `if (ok && d < D)` each load sat in a block of its own and was waited for there — NK dependent round trips
  // (ten at head size 160, twice that in the backward) in front of the K/V staging they were meant to overlap.
  typename Traits<T>::frag v[NK];
#pragma unroll
  for (int kc = 0; kc < NK; ++kc) v[kc] = load_frag<T>(xrow + min((kc << 4) + (g << 2), D - 4));
#pragma unroll
  for (int kc = 0; kc < NK; ++kc) {
    const int d = (kc << 4) + (g << 2);
    x[kc] = (ok && d < D) ? v[kc] : zero_frag<T>();
  }
}

// 16-bit tiles whose rows are the contraction index of a product (V in P.V, K in dS.K, Q / dO in the dK / dV sums)
// stay ROW-major in LDS and are read with gfx950's transposing ds_read_b64_tr_b16: no second, transposed image, no
// 2-byte scatter stores.  Row stride TRS = DP rounded up to an odd multiple of 16 elements (8 dwords): the 8 rows x
// 8 dwords a 32-lane half touches per read then tile the 64 banks exactly.
template <int NK>
__host__ __device__ constexpr int trs() { return NK * 16 + ((NK & 1) ? 0 : 16); }
template <typename T>
constexpr bool kTrRead = sizeof(T) == 2;

template <typename T>
__device__ __forceinline__ typename Traits<T>::frag tr_read(const T* p) {
  typedef __attribute__((address_space(3))) s16x4* lds_ptr;
  return __builtin_bit_cast(typename Traits<T>::frag, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)p));
}

// 8-byte LDS fragment read the compiler may not fuse with its neighbours.  Left alone, hipcc pairs adjacent 8-byte
// reads into ds_read2_b64 across UNRELATED fragments: half the LDS rate (128 B/clk, 32-bank rule -> the padded rows
// conflict 2-way), an address add per row block (8-bit offsets) and two v_mov per MFMA to reassemble the operands —
// a quarter of the forward loop's VALU instructions.  A volatile access is never merged; ds_read_b64 runs at
// 256 B/clk, conflict-free on these images, with a 16-bit immediate offset, straight into the operand registers.
template <typename T>
__device__ __forceinline__ typename Traits<T>::frag lds_frag(const T* p) {
  if constexpr (sizeof(T) == 2) {
    typedef const volatile __attribute__((address_space(3))) unsigned long long* lds_ptr;  // keep it a ds_ access
    const unsigned long long v = *(lds_ptr)p;
    return __builtin_bit_cast(typename Traits<T>::frag, v);
  } else {
    return load_frag<T>(p);
  }
}

// Raise a kernel's dynamic-LDS limit above the 64 KB default; remembered per kernel so that the runtime call
// happens on the first (eager, warm-up) launch only and never inside a stream capture.
constexpr size_t kMaxLdsBytes = 160 * 1024;

template <typename KernelT>
inline int set_dyn_lds(KernelT kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return GA_OK;
  static std::mutex mu;
  static std::unordered_map<const void*, size_t> granted;
  const void* fn = reinterpret_cast<const void*>(kernel);
  std::lock_guard<std::mutex> lock(mu);
  auto it = granted.find(fn);
  if (it != granted.end() && it->second >= bytes) return GA_OK;
  if (bytes > kMaxLdsBytes) return GA_ERR_SHAPE;
  // ask for the whole 160 KB once, so that a later, larger shape never needs a second runtime call mid-capture
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes);
  if (e != hipSuccess) return GA_ERR_LAUNCH;
  granted[fn] = kMaxLdsBytes;
  return GA_OK;
}

}  // namespace ga
