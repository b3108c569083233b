// The UNet's two edge convolutions: conv_in (4 latent channels -> 320) and conv_out (320 -> 4), 3x3, padding 1, stride 1
// (diffusers 0.12.1 UNet2DConditionModel.conv_in / conv_out, called from pipeline_guided_attention.py:647-738 through the UNet
// forward), and — each being the other's adjoint — their backward-to-input passes (conv_in's carries the guidance gradient to
// the latents, pipeline_guided_attention.py:_update_latent).
//
// Neither is matrix-core work: 36 products per output on the thin-input side, four outputs per pixel on the thin-output side —
// one pass over the 320-channel tensor (2.6 MB per image at 64 x 64) bounds both.  The library ran them as per-sample im2col +
// GEMM with layout copies either side (round 4 trace: 250 us of a 5.6 ms batch-3 pass).  Here:
//   thin_in   reads the NCHW latents (a 3 x 18 x 4 patch per 16-pixel row segment through LDS), four output channels per
//             thread with their 36 x 4 weights in registers (v_dot2c_f32_f16 / _bf16: f32 accumulation), writes NHWC.
//   thin_out  stages the 3 x 18 x C patch in LDS, a thread = (pixel, 1/32 of the channels) with that slice's 9 x 4 weight
//             vectors in registers, the 32 partial sums folded on the VALU (DPP), writes NCHW.
// On the largest maps a workgroup takes a few consecutive row segments with the same weights.
// Weights are packed once per weight (ga_conv3x3_thin_pack), transposed + flipped for the adjoint use.
#include "ga_common.h"

namespace {
using namespace ga;

constexpr int kSeg = 16;           // pixels of one image row per workgroup
constexpr int kPatchW = kSeg + 2;  // with the halo
constexpr int kTaps = 9;

template <typename T>
struct Dot2;
template <>
struct Dot2<_Float16> {
  typedef _Float16 v2 __attribute__((ext_vector_type(2)));
  __device__ static __forceinline__ float acc(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(v2, a), __builtin_bit_cast(v2, b), c, false);
  }
};
template <>
struct Dot2<bf16_t> {
  typedef __bf16 v2 __attribute__((ext_vector_type(2)));
  __device__ static __forceinline__ float acc(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2, a), __builtin_bit_cast(v2, b), c, false);
  }
};

template <typename T>
__device__ __forceinline__ uint16_t bits_of(T x) {
  return __builtin_bit_cast(uint16_t, x);
}

// ---- pack: logical V[n][c][ty][tx] = flip ? W[c][n][2 - ty][2 - tx] : W[n][c][ty][tx]   (n outputs, c inputs of the operation)
//   thin input  (C == 4): out[t][p][n]   = (V[n][2p][t], V[n][2p + 1][t])          one 32-bit pair per (tap, input pair, output)
//   thin output (N == 4): out[t][q][n]   = (V[n][2q][t], V[n][2q + 1][t])          q = input-channel pair
__global__ void thin_pack_kernel(const uint16_t* __restrict__ w, uint32_t* __restrict__ out, int N, int C, long long so,
                                 long long si, long long sy, long long sx, int flip) {
  const int total = kTaps * (C / 2) * N;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int n, q, t;
  if (C == 4) {   // [t][p][n]
    n = i % N;
    q = (i / N) % 2;
    t = i / (2 * N);
  } else {        // [t][q][n], N == 4
    n = i % 4;
    q = (i / 4) % (C / 2);
    t = i / (4 * (C / 2));
  }
  const int ty = t / 3, tx = t % 3;
  uint32_t v[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int c = 2 * q + e;
    const long long off = flip ? c * so + n * si + (2 - ty) * sy + (2 - tx) * sx : n * so + c * si + ty * sy + tx * sx;
    v[e] = w[off];
  }
  out[i] = v[0] | (v[1] << 16);
}

// ---- thin input: x [B][4][H][W] -> y [B][H][W][N]
// blockDim = G * PP >= 216 (one patch element per thread), G = N / 4 channel groups, PP pixels at a time; a workgroup takes
// segs_per_wg consecutive 16-pixel row segments with the same weights, the next segment's patch element in flight under the
// current one's arithmetic
constexpr int kInPatch = 3 * kPatchW * 4;

template <typename T>
__global__ void __launch_bounds__(1024)
thin_in_kernel(const T* __restrict__ x, const uint4* __restrict__ wp, const T* __restrict__ bias, T* __restrict__ y, int H, int W,
               int N, int G, int PP, int total_segs, int segs_per_wg) {
  __shared__ uint2 patch[3 * kPatchW];   // [row][col] -> the four input channels of that position
  const int tid = threadIdx.x;
  const int segs = W / kSeg;
  const int g = tid % G, slot = tid / G;

  // the thread's weights: 18 (tap, input pair) x 4 outputs, requested before anything waits
  uint4 wv[2 * kTaps];
#pragma unroll
  for (int i = 0; i < 2 * kTaps; ++i) wv[i] = wp[(size_t)i * G + g];
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (bias) {
    const typename Traits<T>::frag bf = load_frag<T>(bias + 4 * g);
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[j] = Traits<T>::to_f32(bf[j]);
  }

  const int pc = tid % kPatchW, pr = (tid / kPatchW) % 3, pci = tid / (3 * kPatchW);   // this thread's patch element
  auto fetch = [&](int sg) -> uint16_t {
    const int seg = sg % segs, row = (sg / segs) % H, b = sg / (segs * H);
    const int yy = row + pr - 1, xx = seg * kSeg + pc - 1;
    if (tid < kInPatch && yy >= 0 && yy < H && xx >= 0 && xx < W) return bits_of(x[(((size_t)b * 4 + pci) * H + yy) * W + xx]);
    return 0;
  };
  uint16_t* pl = reinterpret_cast<uint16_t*>(patch);
  const int first = blockIdx.x * segs_per_wg;
  const int last = min(first + segs_per_wg, total_segs);
  uint16_t nxt = fetch(first);
  for (int sg = first; sg < last; ++sg) {
    const int seg = sg % segs, row = (sg / segs) % H, b = sg / (segs * H);
    if (sg != first) __syncthreads();
    if (tid < kInPatch) pl[(pr * kPatchW + pc) * 4 + pci] = nxt;
    __syncthreads();
    if (sg + 1 < last) nxt = fetch(sg + 1);
    for (int s = slot; s < kSeg; s += PP) {
      float acc[4] = {bv[0], bv[1], bv[2], bv[3]};
#pragma unroll
      for (int t = 0; t < kTaps; ++t) {
        const uint2 xv = patch[(t / 3) * kPatchW + s + (t % 3)];
        const uint4 w0 = wv[2 * t], w1 = wv[2 * t + 1];
        acc[0] = Dot2<T>::acc(xv.x, w0.x, acc[0]);
        acc[1] = Dot2<T>::acc(xv.x, w0.y, acc[1]);
        acc[2] = Dot2<T>::acc(xv.x, w0.z, acc[2]);
        acc[3] = Dot2<T>::acc(xv.x, w0.w, acc[3]);
        acc[0] = Dot2<T>::acc(xv.y, w1.x, acc[0]);
        acc[1] = Dot2<T>::acc(xv.y, w1.y, acc[1]);
        acc[2] = Dot2<T>::acc(xv.y, w1.z, acc[2]);
        acc[3] = Dot2<T>::acc(xv.y, w1.w, acc[3]);
      }
      typename Traits<T>::frag o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = Traits<T>::from_f32(acc[j]);
      store_frag<T>(y + (((size_t)b * H + row) * W + seg * kSeg + s) * N + 4 * g, o);
    }
  }
}

// ---- thin output: x [B][H][W][C] -> y [B][4][H][W]
// 256 threads: lane & 31 = channel slice (C / 32 channels = PAIRS pairs, PAIRS = 1 ... 5), tid >> 5 = pixel slot (8 at a time).
// The slice's weights for all nine taps and four outputs stay in registers (36 x PAIRS words) for every segment the
// workgroup takes; per (pixel, tap) a thread reads PAIRS words of the patch and issues 4 x PAIRS dot products.
constexpr int kOutThreads = 256;
constexpr int kOutSlices = 32;

template <typename T, int PAIRS>
__global__ void __launch_bounds__(kOutThreads)
thin_out_kernel(const T* __restrict__ x, const uint32_t* __restrict__ wp, const T* __restrict__ bias, T* __restrict__ y, int H, int W,
                int total_segs, int segs_per_wg) {
  constexpr int C = PAIRS * 64;
  constexpr int kVecPx = C / 8;                          // 16-byte vectors per pixel
  constexpr int kPatchVecs = 3 * kPatchW * kVecPx;
  constexpr int kLoads = (kPatchVecs + kOutThreads - 1) / kOutThreads;
  __shared__ uint4 patch[kPatchVecs];
  const int tid = threadIdx.x;
  const int sl = tid & (kOutSlices - 1), slot = tid >> 5;
  const int segs = W / kSeg;

  // wr[t][q][n]: packed [t][C / 2][4] words, this slice's pairs q = sl * PAIRS ... + PAIRS - 1
  uint4 wr[kTaps][PAIRS];
#pragma unroll
  for (int t = 0; t < kTaps; ++t)
#pragma unroll
    for (int q = 0; q < PAIRS; ++q) wr[t][q] = reinterpret_cast<const uint4*>(wp)[t * (C / 2) + sl * PAIRS + q];
  float bv = 0.f;
  if (bias && sl < 4) bv = Traits<T>::to_f32(bias[sl]);

  // the patch of a segment: every load requested in one batch; the NEXT segment's batch goes out before this segment's arithmetic
  // (a workgroup holds 270 registers — one per CU — so nothing else covers that round trip)
  auto fetch = [&](int sg, uint4 (&val)[kLoads]) {
    const int seg = sg % segs, row = (sg / segs) % H, b = sg / (segs * H);
#pragma unroll
    for (int j = 0; j < kLoads; ++j) {
      const int i = tid + j * kOutThreads;
      const int v = i % kVecPx, c = (i / kVecPx) % kPatchW, r = i / (kVecPx * kPatchW);
      const int yy = row + r - 1, xx = seg * kSeg + c - 1;
      // no branch around the load (a load under `if` into a register array is waited for where it is issued): a halo
      // position outside the image reads pixel 0 of the image and a select drops it
      const bool in = i < kPatchVecs && yy >= 0 && yy < H && xx >= 0 && xx < W;
      const size_t pix = in ? ((size_t)b * H + yy) * W + xx : (size_t)b * H * W;
      const uint4 t = reinterpret_cast<const uint4*>(x + pix * C)[in ? v : 0];
      val[j] = in ? t : uint4{0u, 0u, 0u, 0u};
    }
  };
  const int first = blockIdx.x * segs_per_wg;
  const int last = min(first + segs_per_wg, total_segs);
  uint4 val[kLoads];
  fetch(first, val);
  for (int sg = first; sg < last; ++sg) {
    const int seg = sg % segs, row = (sg / segs) % H, b = sg / (segs * H);
    if (sg != first) __syncthreads();   // the previous segment's readers are done with the patch
#pragma unroll
    for (int j = 0; j < kLoads; ++j) {
      const int i = tid + j * kOutThreads;
      if (i < kPatchVecs) patch[i] = val[j];
    }
    __syncthreads();
    if (sg + 1 < last) fetch(sg + 1, val);

    const uint32_t* pw = reinterpret_cast<const uint32_t*>(patch);
#pragma unroll
    for (int h = 0; h < kSeg / 8; ++h) {
      const int p = slot + 8 * h;
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < kTaps; ++t) {
        const uint32_t* xp = pw + ((t / 3) * kPatchW + p + (t % 3)) * (C / 2) + sl * PAIRS;
#pragma unroll
        for (int q = 0; q < PAIRS; ++q) {
          const uint32_t xv = xp[q];
          acc[0] = Dot2<T>::acc(xv, wr[t][q].x, acc[0]);
          acc[1] = Dot2<T>::acc(xv, wr[t][q].y, acc[1]);
          acc[2] = Dot2<T>::acc(xv, wr[t][q].z, acc[2]);
          acc[3] = Dot2<T>::acc(xv, wr[t][q].w, acc[3]);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[j] = group_sum<16>(acc[j]);
        acc[j] += __shfl_xor(acc[j], 16, 64);   // the other half of the 32 slices
      }
      if (sl < 4) {
        const float v = (sl == 0 ? acc[0] : sl == 1 ? acc[1] : sl == 2 ? acc[2] : acc[3]) + bv;
        y[(((size_t)b * 4 + sl) * H + row) * W + seg * kSeg + p] = Traits<T>::from_f32(v);
      }
    }
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// segments per workgroup: at most four workgroups per CU's worth of them (measured: fewer, longer workgroups were slower at batch 3
// on the 64 x 64 map — 5-wave workgroups fill four SIMDs unevenly); the weights are fetched once per workgroup
inline void seg_grid(long long total, int& per, unsigned& grid) {
  per = (int)((total + 1023) / 1024);
  grid = (unsigned)((total + per - 1) / per);
}

template <typename T>
int launch_in(const void* x, const void* wp, const void* bias, void* y, int B, int H, int W, int N, hipStream_t s) {
  const int G = N / 4;
  int PP = (256 + G - 1) / G;
  PP = PP < 1 ? 1 : PP > kSeg ? kSeg : PP;
  if (G * PP < kInPatch || G * PP > 1024) return GA_ERR_SHAPE;
  const long long total = (long long)B * H * (W / kSeg);
  int per;
  unsigned grid;
  seg_grid(total, per, grid);
  hipLaunchKernelGGL(thin_in_kernel<T>, dim3(grid), dim3((unsigned)(G * PP)), 0, s, (const T*)x, (const uint4*)wp, (const T*)bias,
                     (T*)y, H, W, N, G, PP, (int)total, per);
  return check_launch();
}

template <typename T>
int launch_out(const void* x, const void* wp, const void* bias, void* y, int B, int H, int W, int C, hipStream_t s) {
  const long long total = (long long)B * H * (W / kSeg);
  // One workgroup per CU is resident (register budget).  Up to 1024 segments (the 64 x 64 maps): 256 workgroups, each streaming
  // its share with the next patch in flight (batch 3: 17.4 -> 15.1 us, batch 2: 12.0 -> 11.0); the larger maps measured faster
  // as up to 1024 shorter workgroups (128 x 128 at batch 3: 42 against 50 us) — gpurun_out probes 9 and 18.
  const int per = total <= 1024 ? (int)((total + 255) / 256) : (int)((total + 1023) / 1024);
  const unsigned grid = (unsigned)((total + per - 1) / per);
#define GA_THIN_OUT(P)                                                                                                   \
  case P:                                                                                                                \
    hipLaunchKernelGGL((thin_out_kernel<T, P>), dim3(grid), dim3(kOutThreads), 0, s, (const T*)x, (const uint32_t*)wp,    \
                       (const T*)bias, (T*)y, H, W, (int)total, per);                                                    \
    break;
  switch (C / 64) {
    GA_THIN_OUT(1)
    GA_THIN_OUT(2)
    GA_THIN_OUT(3)
    GA_THIN_OUT(4)
    GA_THIN_OUT(5)
    default: return GA_ERR_SHAPE;
  }
#undef GA_THIN_OUT
  return check_launch();
}

}  // namespace

extern "C" long long ga_conv3x3_thin_packed_elems(int Cout, int Cin) {
  if (Cin == 4 && Cout >= 4) return 36LL * Cout;
  if (Cout == 4 && Cin >= 4) return 36LL * Cin;
  return 0;
}

extern "C" int ga_conv3x3_thin_supported(int H, int W, int Cin, int Cout) {
  if (H < 1 || W < kSeg || W % kSeg != 0) return 0;
  if (Cin == 4) return Cout % 4 == 0 && Cout >= 64 && Cout <= 4096 ? 1 : 0;
  if (Cout == 4) return Cin % 64 == 0 && Cin >= 64 && Cin <= 320 ? 1 : 0;
  return 0;
}

extern "C" int ga_conv3x3_thin_pack(const void* weight, void* packed, int Cout, int Cin, long long so, long long si, long long sy,
                                    long long sx, int transpose_flip, int dtype, ga_stream_t stream) {
  if (!weight || !packed) return GA_ERR_NULL;
  if (dtype != GA_F16 && dtype != GA_BF16) return GA_ERR_DTYPE;
  const int N = transpose_flip ? Cin : Cout, C = transpose_flip ? Cout : Cin;   // outputs / inputs of the operation
  if (!((C == 4 && N >= 4 && N % 4 == 0) || (N == 4 && C % 2 == 0 && C >= 4))) return GA_ERR_SHAPE;
  if (!aligned16(packed)) return GA_ERR_ALIGN;
  const int total = kTaps * (C / 2) * N;
  hipLaunchKernelGGL(thin_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     (const uint16_t*)weight, (uint32_t*)packed, N, C, so, si, sy, sx, transpose_flip ? 1 : 0);
  return check_launch();
}

extern "C" int ga_conv3x3_thin_in(const void* x, const void* packed, const void* bias, void* y, int B, int H, int W, int Cout,
                                  int dtype, ga_stream_t stream) {
  if (!x || !packed || !y) return GA_ERR_NULL;
  if (B < 1 || !ga_conv3x3_thin_supported(H, W, 4, Cout) || Cout / 4 > 1024) return GA_ERR_SHAPE;
  if ((long long)B * H * (W / kSeg) >= (1LL << 31)) return GA_ERR_SHAPE;
  if (!aligned16(packed) || (reinterpret_cast<uintptr_t>(y) & 7u) || (bias && (reinterpret_cast<uintptr_t>(bias) & 7u)))
    return GA_ERR_ALIGN;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case GA_F16: return launch_in<_Float16>(x, packed, bias, y, B, H, W, Cout, s);
    case GA_BF16: return launch_in<bf16_t>(x, packed, bias, y, B, H, W, Cout, s);
    default: return GA_ERR_DTYPE;
  }
}

extern "C" int ga_conv3x3_thin_out(const void* x, const void* packed, const void* bias, void* y, int B, int H, int W, int Cin,
                                   int dtype, ga_stream_t stream) {
  if (!x || !packed || !y) return GA_ERR_NULL;
  if (B < 1 || !ga_conv3x3_thin_supported(H, W, Cin, 4)) return GA_ERR_SHAPE;
  if ((long long)B * H * (W / kSeg) >= (1LL << 31)) return GA_ERR_SHAPE;
  if (!aligned16(x) || !aligned16(packed)) return GA_ERR_ALIGN;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case GA_F16: return launch_out<_Float16>(x, packed, bias, y, B, H, W, Cin, s);
    case GA_BF16: return launch_out<bf16_t>(x, packed, bias, y, B, H, W, Cin, s);
    default: return GA_ERR_DTYPE;
  }
}
