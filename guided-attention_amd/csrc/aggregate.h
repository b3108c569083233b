// K2 (utils/ptp_utils.py:273-289, select = 0): one (pixel, token) element of the head-map mean — shared by
// ga_aggregate_maps (aggregate.hip) and the fused aggregate + loss launch (smooth_loss.hip).
#pragma once
#include "ga_common.h"

namespace ga {

constexpr int kMaxMaps = 128;  // SDXL at 1024^2: 60 stored 32x32 cross maps (AggArgs stays under the 4 KB kernarg limit)

struct AggArgs {
  const void* maps[kMaxMaps];
  int heads[kMaxMaps];
  int n_maps;
  int total_heads;
};

inline int fill_agg_args(AggArgs& a, const void* const* maps, const int* heads, int n_maps) {
  if (!maps || !heads) return GA_ERR_NULL;
  if (n_maps < 1 || n_maps > kMaxMaps) return GA_ERR_SHAPE;
  a.n_maps = n_maps;
  a.total_heads = 0;
  for (int i = 0; i < n_maps; ++i) {
    if (!maps[i]) return GA_ERR_NULL;
    if (heads[i] < 1) return GA_ERR_SHAPE;
    a.maps[i] = maps[i];
    a.heads[i] = heads[i];
    a.total_heads += heads[i];
  }
  return GA_OK;
}

// A[e] = (1/M) * sum over every head-map m of every listed tensor of map_m[e], walked in list order (fixed order:
// bitwise reproducible, no atomics); reads are coalesced across the lanes (token fastest).
template <typename T>
__device__ __forceinline__ float aggregate_element(const AggArgs& a, int e, int n_elem) {
  float acc = 0.f;
  // Four tensors x eight head-maps = 32 loads in flight per lane (the 5 x 8 maps of the SD-1.x case take two round
  // trips instead of five); the adds then run in list order within the batch: tensor-major, head-minor.  Tensors whose
  // head count is not a multiple of 8 finish in the scalar tail below, still in order.
  for (int m0 = 0; m0 < a.n_maps; m0 += 4) {
    int hmax = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (m0 + i < a.n_maps) hmax = max(hmax, a.heads[m0 + i] & ~7);
    for (int h0 = 0; h0 < hmax; h0 += 8) {
      T v[4][8];
      bool lv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool live = m0 + i < a.n_maps && h0 + 8 <= (a.heads[m0 + i < a.n_maps ? m0 + i : 0] & ~7);
        // every load is issued, from an address that exists (head 0 of the batch's first tensor when the slot is not
        // live), and dropped by the select in the sum: `live ? load : 0` compiled to one load and one wait at a time —
        // 32 dependent round trips where 32 loads in flight were meant
        const T* src = static_cast<const T*>(a.maps[live ? m0 + i : m0]) + e;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[i][j] = src[(size_t)(live ? h0 + j : 0) * n_elem];
        lv[i] = live;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += lv[i] ? Traits<T>::to_f32(v[i][j]) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (m0 + i >= a.n_maps) continue;
      const T* src = static_cast<const T*>(a.maps[m0 + i]) + e;
      for (int h = a.heads[m0 + i] & ~7; h < a.heads[m0 + i]; ++h) acc += Traits<T>::to_f32(src[(size_t)h * n_elem]);
    }
  }
  return acc / (float)a.total_heads;
}

}  // namespace ga
