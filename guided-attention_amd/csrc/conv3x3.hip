// 3x3 convolution (pad 1, stride 1 or 2) on channels-last activations as an implicit GEMM on gfx950 MFMA — the
// ResnetBlock / up- / down-sampling convolutions of the UNet the guidance pass runs forward and backward
// (diffusers 0.12.1 blocks called from pipeline_guided_attention.py:583-743; the reference leaves them to cuDNN).
//
//   Y[m][n] = sum over tap t = (ky, kx), channel c of  X[pixel(m) + tap offset][c] * Wp[t][n][c]      (+ bias[n] + R[m][n])
//   m = (batch, oy, ox) output pixel, n = output channel, GEMM depth K = 9 * Cin
//
// Why not the library kernel: at guidance batch 1 the UNet's convolutions are small-M GEMMs (M = 4096 / 1024 / 256 / 64
// pixels against N = 320 ... 1280 channels); a 128 x 128 macro-tile leaves 20 ... 96 workgroups for 256 CUs.  This
// kernel takes the tile from {128x128, 128x64, 64x64} and splits the GEMM depth across workgroups (split-K, f32 partial
// slabs summed by a small epilogue kernel) so that every shape fills the chip.
//
// Mapping: 256 threads = 4 waves in a 2 x 2 grid over the macro-tile, each wave (BM/2) x (BN/2) as 32 x 32 MFMA blocks
// (v_mfma_f32_32x32x16: 16 accumulator registers per block).  The weight fragment is the MFMA's A operand and the
// pixel fragment its B operand, so a lane ends up with 4 consecutive output channels of ONE pixel per register quad: the
// tile goes through LDS once more and leaves as whole 16-byte row pieces with bias and residual added on the way.
// LDS rows hold 64 channels (one k-step) at a 144-byte stride: the 16-byte fragment reads are conflict-free.
// Two kernels share the tile mapping and the epilogue:
//   conv3x3_kernel        per k-step (64 channels of one tap) an A tile [BM pixels][64] gathered from the shifted input
//                         pixels and a B tile [BN][64] of the pre-packed weights Wp[tap][n][c], double-buffered, two
//                         steps of loads in flight in two register sets, one barrier per step.  Any shape: stride 2,
//                         8x8 maps, maps whose rows do not tile into whole 128- / 64-pixel tiles, ga_gemm_nt (one tap).
//   conv3x3_patch_kernel  stride 1, tiles of whole image rows: the (rows + 2) x (W + 2) input patch of the tile is staged
//                         ONCE per 64-channel chunk and all nine taps read it at shifted addresses; weights per
//                         (chunk, tap) with 3 - 6 steps in flight.  See the comment above it.
// Staging loads are raw buffer loads with 32-bit offsets; whatever lies outside the image or the tile carries an offset
// past the descriptor's size and comes back as zeros (no branch, no zero-fill pass).  Workgroups are mapped XCD-aware
// (tile_of_workgroup): the tiles that share a weight slice (or a pixel slab) run on one XCD's L2.
// The backward-to-input of a stride-1 convolution is the same kernel on the upstream gradient with the weights
// flipped and transposed (pre-packed once per weight version by the host).
#include <cstdlib>

#include "attn_common.h"

using namespace ga;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

#ifndef GA_CONV_PRIO
#define GA_CONV_PRIO 0
#endif
#ifndef GA_CONV_KC
#define GA_CONV_KC 64
#endif
// Ablation switches for tools/conv_tune.py variants (micro-benchmark builds only; results are wrong by construction,
// only the time is read): bit 0 no global loads inside the loop, bit 1 no LDS stores inside the loop, bit 2 no MFMAs,
// bit 3 no barriers inside the loop, bit 4 (patch variant) no patch reload per chunk; bit 5 (patch variant) no fragment
// reads (one set of fragments is read once).
#ifndef GA_CONV_ABL
#define GA_CONV_ABL 0
#endif
constexpr int kKC = GA_CONV_KC;    // depth of one k-step (channels of one tap)
constexpr int kLD = kKC + 8;       // LDS row stride in elements (80 bytes): conflict-free 16-byte fragment reads
constexpr int kThreads = 256;

template <typename T>
struct Mma32;
template <>
struct Mma32<_Float16> {
  __device__ static __forceinline__ f32x16 run(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};
template <>
struct Mma32<bf16_t> {
  __device__ static __forceinline__ f32x16 run(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};

struct ConvArgs {
  int B, H, W, Cin, Ho, Wo, Cout, stride;
  int M;            // B * Ho * Wo
  int steps;        // k-steps in all: 9 * Cin / 32
  int steps_per;    // k-steps per split
  int tm, tn;       // m tiles, n tiles of the launch (grid = tm * tn * splits workgroups, one dimension)
  int splits;       // k slices per tile (split-K)
  int n_fastest;    // workgroup order inside an XCD's run: n tiles fastest (1) or m tiles fastest (0)
  int pad;          // 1: 3x3 taps around the pixel; 0: a single tap (plain GEMM  Y[m][n] = sum_c X[m][c] W[n][c])
  int lane_rot;     // patch kernel on 16-wide maps: lanes 16..31 of a 32-pixel block take their row's pixels rotated by 2
                    // (see lane_pixel); 0 everywhere else
  unsigned x_bytes, w_bytes;   // sizes of X and Wp (buffer descriptors: loads beyond them return zeros); < 2 GiB
  // Weight addressing, in elements: (tap t, row n, channel chunk c) starts at t * w_tap + (n / 64) * w_blk + (n % 64) * w_row
  // + c * w_chunk.  The 3x3 pack is BLOCKED — [tap][n / 64][c][n % 64][64 channels]: the 64 rows x 128 bytes a workgroup
  // fetches per k-step are ONE contiguous 8 KB run (w_row = 64, w_chunk = 4096).  In the row-major [N][K] layout the plain
  // GEMM entry point reads (the framework's own Linear weights; w_row = K, w_chunk = 64) the same fetch is 64 pieces of 128
  // bytes 2 K bytes apart, and the cold weight streams of the 1280-channel levels ran at 1.3 - 1.8 TB/s on it.
  int w_tap, w_blk, w_row, w_chunk;
  // Patch kernels: the tile geometry (patch_geometry) and the divisors of the per-thread index arithmetic, prepared by the
  // host.  The prologue of conv3x3_patch_dma_kernel computed all of it per workgroup with runtime integer divisions — 12
  // reciprocals and 962 VALU instructions, 190 of them quarter-rate 32-bit multiplies, before the first load was issued
  // (~2.6 us on the critical path of a 17 - 30 us launch).
  int pg_nseg, pg_srows;
  int pg_dpr, pg_dpc;   // a thread's next patch piece is 32 patch pixels on: 32 / (W + 2) rows and 32 % (W + 2) columns
  FastDiv d_tm, d_tn, d_w, d_hw, d_pw, d_segpw, d_segpx, d_wo, d_howo;
  int up;           // 1: X is [B][H/2][W/2][Cin] and the convolution runs on its nearest-neighbour 2x up-sampling (H, W are the
                    // up-sampled sizes): only the patch gather of conv3x3_patch_dma_kernel changes — pixel (iy, ix) of the
                    // patch is read from (iy >> 1, ix >> 1) — and the [B][H][W][Cin] intermediate is never written
  // Optional GroupNorm statistics of the STORED result for the norm layer that consumes it (ga_conv3x3_nhwc_gn): per (image,
  // m tile, group) partial (sum, sum of squares) in the layout ga_group_norm_apply folds — the consumer's statistics launch
  // (a full read of the tensor by its own kernel) disappears.  gn_cbias: the per-(image, channel) term that norm adds to its
  // input (the ResnetBlock's time embedding): the sums are those of result + term, the stored result is unchanged.
  float* gn_partials;      // [B][2 * gn_tpi][gn_G][2] or null
  const void* gn_cbias;    // [B][Cout] T or null
  int gn_cg, gn_G, gn_tpi; // channels per group, groups, m tiles per image (tiles never straddle images: Ho * Wo % BM == 0)
};

// All of the argument block in SGPRs at this point: an empty asm that "reads" every field makes the compiler fetch them in
// one batch of scalar loads behind ONE wait.  Left to itself it loads a field where it is first used — the patch-DMA kernel
// began with four dependent load / wait stages over the block's four cache lines (two of the stages missed the scalar cache)
// in front of its first weight DMA.
__device__ __forceinline__ void conv_args_resident(const ConvArgs& a) {
#if defined(__HIP_DEVICE_COMPILE__)
#define GA_TOUCH(x) asm volatile("" ::"s"(x))
  GA_TOUCH(a.B); GA_TOUCH(a.H); GA_TOUCH(a.W); GA_TOUCH(a.Cin); GA_TOUCH(a.Ho); GA_TOUCH(a.Wo); GA_TOUCH(a.Cout); GA_TOUCH(a.stride);
  GA_TOUCH(a.M); GA_TOUCH(a.steps); GA_TOUCH(a.steps_per); GA_TOUCH(a.tm); GA_TOUCH(a.tn); GA_TOUCH(a.splits); GA_TOUCH(a.n_fastest);
  GA_TOUCH(a.pad); GA_TOUCH(a.lane_rot); GA_TOUCH(a.x_bytes); GA_TOUCH(a.w_bytes);
  GA_TOUCH(a.w_tap); GA_TOUCH(a.w_blk); GA_TOUCH(a.w_row); GA_TOUCH(a.w_chunk);
  GA_TOUCH(a.pg_nseg); GA_TOUCH(a.pg_srows); GA_TOUCH(a.pg_dpr); GA_TOUCH(a.pg_dpc);
  GA_TOUCH(a.d_tm.d); GA_TOUCH(a.d_tm.m); GA_TOUCH(a.d_tn.d); GA_TOUCH(a.d_tn.m); GA_TOUCH(a.d_w.d); GA_TOUCH(a.d_w.m);
  GA_TOUCH(a.d_hw.d); GA_TOUCH(a.d_hw.m); GA_TOUCH(a.d_pw.d); GA_TOUCH(a.d_pw.m); GA_TOUCH(a.d_segpw.d); GA_TOUCH(a.d_segpw.m);
  GA_TOUCH(a.d_segpx.d); GA_TOUCH(a.d_segpx.m); GA_TOUCH(a.d_wo.d); GA_TOUCH(a.d_wo.m); GA_TOUCH(a.d_howo.d); GA_TOUCH(a.d_howo.m);
  GA_TOUCH(a.up);
#undef GA_TOUCH
#endif
}

__device__ __forceinline__ unsigned w_row_bytes(const ConvArgs& a, int n, int piece) {   // (row n, 16-byte piece) of chunk 0, tap 0
  return (unsigned)(((n >> 6) * a.w_blk + (n & 63) * a.w_row + 8 * piece) * 2);
}
__device__ __forceinline__ unsigned w_step_bytes(const ConvArgs& a, int tap, int chunk) {
  return (unsigned)((tap * a.w_tap + chunk * a.w_chunk) * 2);
}

// Which pixel of its 32-pixel block an MFMA lane (0..31) owns.  Identity, except in the patch kernel on 16-wide maps: a
// block is two image rows there, lanes 16..31 read the patch 18 (not 16) rows further, and the bank of a 144-byte row
// repeats every 16 rows — within the hardware's 16-lane read groups two lanes then meet on one bank (30 % of the LDS
// cycles were conflict cycles on the 16x16 level).  Letting lane 16 + j own pixel (j - 2) mod 16 of the second row puts
// every lane of a group on its own bank again; the epilogue follows the same map.
__device__ __forceinline__ int lane_pixel(int fr, int rot) { return fr < 16 || rot == 0 ? fr : 16 + ((fr - 16 - rot) & 15); }

// Workgroup -> (m tile, n tile, k split), XCD-aware.  The hardware deals consecutive workgroup ids round-robin over the
// 8 XCDs (ids b and b + 8 share an L2).  The workgroups that read the same weight slice Wp[taps of split][n tile] are
// its tm m-tiles: give each XCD one contiguous run of the m-fastest order, so a weight slice is fetched into ONE L2
// instead of up to 8 (the 1280-channel layers otherwise re-read their 29.5 MB of weights once per m tile: 7x the
// algorithmic bytes at the fabric counters).  Bijective for any grid size; placement is a speed matter only.
template <int BM, int BN>
__device__ __forceinline__ void tile_of_workgroup(const ConvArgs& a, int& m0, int& n0, int& split) {
  const int total = a.tm * a.tn * a.splits, q = total >> 3, r = total & 7;   // == gridDim.x, without the hidden-argument load
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  // n_fastest: activations outweigh the weights (64x64 level): an XCD keeps an m tile's pixels and sweeps the n tiles.
  // Written with selects, not two branches assigning m0 / n0 in turn (the compiler merged those into a two-element stack array)
  const bool nf = a.n_fastest != 0;
  const FastDiv d1 = nf ? a.d_tn : a.d_tm, d2 = nf ? a.d_tm : a.d_tn;
  const int t1 = nf ? a.tn : a.tm, t2 = nf ? a.tm : a.tn;
  const int rest = fdiv(logical, d1), i1 = logical - rest * t1;
  split = fdiv(rest, d2);
  const int i2 = rest - split * t2;
  m0 = (nf ? i2 : i1) * BM;
  n0 = (nf ? i1 : i2) * BN;
}

// Epilogue of both kernels.  acc[j][i][r]: output channel n = n0 + wn*WN + j*32 + (r & 3) + 8 * (r >> 2) + 4 * fh,
//                                          pixel          m = m0 + wm*WM + i*32 + fr
// SPLITK: the depth is divided over `splits` workgroups per tile and the reduction happens INSIDE the launch (round 2
// ran a second kernel over f32 slabs: 42 k launches and 4 % of the kernel time per bench run).  Every slice stores its f32
// accumulators write-through (sc1) in its own thread order — 16 bytes per lane, perfectly coalesced, and the reducer's
// thread t reads exactly what thread t of the other slices wrote — drains, and one lane takes a ticket; the slice whose
// ticket is last re-reads all of them with sc1 loads IN SLICE ORDER (bitwise reproducible whoever arrives last), goes on
// to the common epilogue below and returns the ticket word to zero for the next launch.
// Then through LDS (every wave has passed the loop's last barrier: the staging buffers are free) and out as whole
// 16-byte row pieces with bias and residual added.
template <typename T, int BM, int BN, bool SPLITK>
__device__ __forceinline__ void conv_epilogue(f32x16 (&acc)[BN / 64][BM / 64], T* lds, T* __restrict__ Y,
                                              float* __restrict__ part, unsigned* __restrict__ tickets,
                                              const T* __restrict__ bias, const T* __restrict__ residual,
                                              const ConvArgs& a, int m0, int n0, int split) {
  constexpr int WM = BM / 2, WN = BN / 2, IM = WM / 32, JN = WN / 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, fr = lane_pixel(lane & 31, a.lane_rot), fh = lane >> 5;
  // Bias and residual of this thread's output vectors, requested in ONE batch with clamped addresses — at entry, or, in a
  // split-K launch, by the slice that goes on to reduce (in front of its slab reads).  The first form loaded them inside the
  // output loop, each behind its own wait: two dependent round trips per output vector, 8 (128 x 64 tile) to 16 (128 x 128) in
  // a row at the end of every launch.
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  constexpr int VPR = BN / 8;                      // 16-byte vectors per tile row
  constexpr int NV = BM * VPR / kThreads;          // output vectors per thread; kThreads % VPR == 0: one column run per thread
  static_assert(BM * VPR % kThreads == 0 && kThreads % VPR == 0, "output vector map");
  const int cv = (tid % VPR) * 8, r0 = tid / VPR;
  u32x4 bvec = {0u, 0u, 0u, 0u}, rvec[NV], gvec = {0u, 0u, 0u, 0u};
  const bool gn = a.gn_partials != nullptr;
  const int gn_b = gn ? sdiv(m0, a.d_howo) : 0;      // the tile's image (gn: whole tiles of one image)
  auto prefetch = [&]() {
    const int nc = min(n0 + cv, a.Cout - 8);
    if (bias != nullptr) bvec = *reinterpret_cast<const u32x4*>(bias + nc);
    if (gn && a.gn_cbias != nullptr) gvec = *reinterpret_cast<const u32x4*>(static_cast<const T*>(a.gn_cbias) + (size_t)gn_b * a.Cout + nc);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      rvec[k] = u32x4{0u, 0u, 0u, 0u};
      if (residual != nullptr) {
        const int m = min(m0 + r0 + k * (kThreads / VPR), a.M - 1);
        rvec[k] = *reinterpret_cast<const u32x4*>(residual + (size_t)m * a.Cout + nc);
      }
    }
  };
  if constexpr (!SPLITK) prefetch();
  if constexpr (SPLITK) {
    constexpr int QUADS = JN * IM * 4;
    const int tile = (n0 / BN) * a.tm + m0 / BM;
    const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(part, 0, 0x7ffffff0, 0x00020000);
    const unsigned per_slice = (unsigned)(a.tm * a.tn) * (BM * BN * 4u);
    const unsigned tbase = (unsigned)tile * (BM * BN * 4u) + (unsigned)tid * 16u;
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
      for (int i = 0; i < IM; ++i)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const u32x4 v = {__float_as_uint(acc[j][i][4 * qd]), __float_as_uint(acc[j][i][4 * qd + 1]),
                           __float_as_uint(acc[j][i][4 * qd + 2]), __float_as_uint(acc[j][i][4 * qd + 3])};
          __builtin_amdgcn_raw_buffer_store_b128(v, srsrc, tbase + ((j * IM + i) * 4 + qd) * (kThreads * 16u),
                                                 split * per_slice, 16);   // aux 16 = sc1: write-through
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
      for (int i = 0; i < IM; ++i) keep_live(acc[j][i]);   // the stored registers stay untouched until the stores are done
    __syncthreads();
    int* flag = reinterpret_cast<int*>(lds);
    if (tid == 0) {
      const unsigned old = __hip_atomic_fetch_add(tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = old == (unsigned)a.splits - 1;
      if (last) __hip_atomic_store(tickets + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      flag[0] = last;
    }
    __syncthreads();
    const int last = flag[0];
    __syncthreads();
    if (!last) return;
    prefetch();
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
      for (int i = 0; i < IM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
    // SB slices' slabs in flight at a time (one slice per round trip was `splits` dependent round trips); slots past the last
    // slice re-read it and add zero.  Still summed in slice order.
    constexpr int SB = QUADS <= 4 ? 4 : (QUADS <= 8 ? 2 : 1);
    for (int s0 = 0; s0 < a.splits; s0 += SB) {
      u32x4 v[SB][QUADS];
#pragma unroll
      for (int sb = 0; sb < SB; ++sb) {
        const unsigned soff = (unsigned)min(s0 + sb, a.splits - 1) * per_slice;
#pragma unroll
        for (int q = 0; q < QUADS; ++q)
          v[sb][q] = __builtin_amdgcn_raw_buffer_load_b128(srsrc, tbase + q * (kThreads * 16u), soff, 16);   // sc1
      }
#pragma unroll
      for (int sb = 0; sb < SB; ++sb) {
        const bool in = s0 + sb < a.splits;
#pragma unroll
        for (int j = 0; j < JN; ++j)
#pragma unroll
          for (int i = 0; i < IM; ++i)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                acc[j][i][4 * qd + r] += in ? __uint_as_float(v[sb][(j * IM + i) * 4 + qd][r]) : 0.f;
      }
    }
  }
  {
    constexpr int LDC = BN + 8;
    T* Cs = lds;
    float gsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, gsq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < IM; ++i)
#pragma unroll
      for (int j = 0; j < JN; ++j)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          typename Traits<T>::frag f;
#pragma unroll
          for (int r = 0; r < 4; ++r) f[r] = Traits<T>::from_f32(acc[j][i][4 * qd + r]);
          store_frag<T>(Cs + (wm * WM + i * 32 + fr) * LDC + wn * WN + j * 32 + 8 * qd + 4 * fh, f);
        }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int r = r0 + k * (kThreads / VPR);
      const int m = m0 + r, n = n0 + cv;
      uint4 val = *reinterpret_cast<const uint4*>(Cs + r * LDC + cv);
      if (bias != nullptr || residual != nullptr) {
        T* e = reinterpret_cast<T*>(&val);
        const T* be = reinterpret_cast<const T*>(&bvec);
        const T* re = reinterpret_cast<const T*>(&rvec[k]);
#pragma unroll
        for (int q = 0; q < 8; ++q)
          e[q] = Traits<T>::from_f32(Traits<T>::to_f32(e[q]) + Traits<T>::to_f32(be[q]) + Traits<T>::to_f32(re[q]));
      }
      if (m < a.M && n < a.Cout) *reinterpret_cast<uint4*>(Y + (size_t)m * a.Cout + n) = val;
      if (gn && m < a.M && n < a.Cout) {   // sums of the values AS STORED (+ the consumer's per-channel term)
        const T* e = reinterpret_cast<const T*>(&val);
        const T* ge = reinterpret_cast<const T*>(&gvec);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float x = Traits<T>::to_f32(e[q]) + Traits<T>::to_f32(ge[q]);
          gsum[q] += x;
          gsq[q] += x * x;
        }
      }
    }
    if (gn) {
      // this thread's 8 channels lie in at most two groups (8 <= channels per group): gA takes the first `split` of them
      const int c_abs = n0 + cv, gA = c_abs / a.gn_cg, split = min(8, (gA + 1) * a.gn_cg - c_abs);
      float4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (q < split) {
          r.x += gsum[q];
          r.y += gsq[q];
        } else {
          r.z += gsum[q];
          r.w += gsq[q];
        }
      }
      __syncthreads();                                   // every thread has read its rows of the staging tile
      float4* lds4 = reinterpret_cast<float4*>(lds);
      lds4[tid] = r;
      __syncthreads();
      const int n_end = min(n0 + BN, a.Cout);
      const int g_first = n0 / a.gn_cg, g_last = (n_end - 1) / a.gn_cg;
      const int g = g_first + tid;
      if (g <= g_last) {
        const int c_lo = max(g * a.gn_cg, n0), c_hi = min((g + 1) * a.gn_cg, n_end);      // this tile's channels of group g
        float sa = 0.f, sq = 0.f;
        for (int v = (c_lo - n0) >> 3; v <= (c_hi - 1 - n0) >> 3; ++v) {
          const bool first = (n0 + 8 * v) / a.gn_cg == g;   // g is this vector's gA, otherwise its gA + 1
          for (int pr = 0; pr < kThreads / VPR; ++pr) {   // fixed order: the same bits whoever runs first
            const float4 e = lds4[pr * VPR + v];
            sa += first ? e.x : e.z;
            sq += first ? e.y : e.w;
          }
        }
        // slot 2 t of m tile t: the n tile a group STARTS in; slot 2 t + 1: the n tile it continues into (zero when it does not)
        const int t = (m0 - gn_b * (a.Ho * a.Wo)) / BM;
        float2* out = reinterpret_cast<float2*>(a.gn_partials) + ((size_t)gn_b * 2 * a.gn_tpi + 2 * t) * a.gn_G + g;
        const bool starts = g * a.gn_cg >= n0, ends = (g + 1) * a.gn_cg <= n_end;
        if (starts) {
          out[0] = float2{sa, sq};
          if (ends) out[a.gn_G] = float2{0.f, 0.f};
        } else {
          out[a.gn_G] = float2{sa, sq};
        }
      }
    }
  }
}

// OUT_F32 = true: split-K launch (the reduction happens in conv_epilogue)
template <typename T, int BM, int BN, bool OUT_F32>
__global__ __launch_bounds__(kThreads, 2) void conv3x3_kernel(const T* __restrict__ X, const T* __restrict__ Wp,
                                                           T* __restrict__ Y, float* __restrict__ part,
                                                           unsigned* __restrict__ tickets,
                                                           const T* __restrict__ bias, const T* __restrict__ residual,
                                                           ConvArgs a) {
  constexpr int WM = BM / 2, WN = BN / 2;     // wave tile
  constexpr int IM = WM / 32, JN = WN / 32;   // 32x32 blocks per wave
  constexpr int QP = kKC / 8;                 // 16-byte pieces per row of a k-step
  constexpr int RPP = kThreads / QP;          // rows staged per pass
  constexpr int PA = BM / RPP, PB = BN / RPP; // staging passes
  constexpr int kTile = (BM + BN) * kLD;      // elements per LDS buffer
  constexpr int kCtile = BM * (BN + 8);       // output staging tile (T), row stride BN + 8
  constexpr int kLds = 2 * kTile > kCtile ? 2 * kTile : kCtile;
  __shared__ __attribute__((aligned(16))) T lds[kLds];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int m0, n0, split;
  conv_args_resident(a);
  tile_of_workgroup<BM, BN>(a, m0, n0, split);
  const int it0 = split * a.steps_per, it1 = min(a.steps, it0 + a.steps_per);
  const int cchunks = a.Cin / kKC;

  // ---- staging assignment: thread -> (row = tid / QP (+ RPP per pass), 16-byte piece sq of the row's 128-byte slice).
  // Loads are raw buffer loads with a 32-bit byte offset per lane: a lane whose pixel falls outside the image (or whose
  // row is beyond M / Cout) carries an offset of 2 GiB, past the descriptor's size, and the hardware returns zeros —
  // no branch around a load, no zero-initialised destination (either one makes the compiler drain every load in
  // flight before the next group is issued, which would undo the look-ahead below).
  constexpr unsigned kOob = 0x80000000u;
  const int srow = tid / QP, sq = tid % QP;
  int a_iy[PA], a_ix[PA];        // top-left input pixel of the 3x3 window (may be -1); iy = -4 marks a row beyond M
  int a_off[PA];                 // byte offset of (pixel (iy, ix), channel 8 * sq) — may be negative, only used in bounds
#pragma unroll
  for (int p = 0; p < PA; ++p) {
    const int m = m0 + srow + RPP * p;
    const bool ok = m < a.M;
    const int mm = ok ? m : 0;
    const int b = sdiv(mm, a.d_howo), rem = mm - b * (a.Ho * a.Wo);
    const int oy = sdiv(rem, a.d_wo), ox = rem - oy * a.Wo;
    a_iy[p] = ok ? oy * a.stride - a.pad : -4;
    a_ix[p] = ox * a.stride - a.pad;
    a_off[p] = (((b * a.H + a_iy[p]) * a.W + a_ix[p]) * a.Cin + 8 * sq) * (int)sizeof(T);
  }
  unsigned b_off[PB];            // byte offset of (row n, channel 8 * sq) inside one tap's [Cout][Cin] slab, or kOob
#pragma unroll
  for (int p = 0; p < PB; ++p) {
    const int n = n0 + srow + RPP * p;
    b_off[p] = n < a.Cout ? w_row_bytes(a, n, sq) : kOob;
  }
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(X), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(Wp), 0, a.w_bytes, 0x00020000);
  // Two register sets: the loads of k-step it + 2 are issued before the MFMAs of step it and only waited for one
  // whole step later (the compiler's counted vmcnt leaves the younger set in flight).  One step of look-ahead left
  // every step waiting out a full memory round trip (1.1 - 1.6 us per step against 0.1 - 0.2 us of MFMAs).
  uint4 ra0[PA], rb0[PB], ra1[PA], rb1[PB];
  // (tap, channel chunk) of the next k-step to load, advanced incrementally (no division per step); past the last step
  // of this split it stays put: that step is loaded again and never used (see the loop below)
  int ld_it = it0, ld_tap = it0 / cchunks, ld_c = it0 - ld_tap * cchunks;
  int ld_ky = ld_tap / 3, ld_kx = ld_tap - 3 * ld_ky;
  bool in_loop = false;
  auto load_step = [&](uint4 (&ra)[PA], uint4 (&rb)[PB]) {
    if ((GA_CONV_ABL & 1) && in_loop) return;
    const int a_step = ((ld_ky * a.W + ld_kx) * a.Cin + ld_c * kKC) * (int)sizeof(T);                  // wave-uniform
    const unsigned b_step = w_step_bytes(a, ld_tap, ld_c);        // kOob + it stays >= 2 GiB
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      const bool in = (unsigned)(a_iy[p] + ld_ky) < (unsigned)a.H && (unsigned)(a_ix[p] + ld_kx) < (unsigned)a.W;
      const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, in ? (unsigned)(a_off[p] + a_step) : kOob, 0, 0);
      ra[p] = uint4{x[0], x[1], x[2], x[3]};
    }
#pragma unroll
    for (int p = 0; p < PB; ++p) {
      const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, b_off[p] + b_step, 0, 0);
      rb[p] = uint4{x[0], x[1], x[2], x[3]};
    }
    if (ld_it + 1 < it1) {
      ++ld_it;
      if (++ld_c == cchunks) {
        ld_c = 0;
        ++ld_tap;
        if (++ld_kx == 3) {
          ld_kx = 0;
          ++ld_ky;
        }
      }
    }
  };
  auto store_step = [&](int buf, const uint4 (&ra)[PA], const uint4 (&rb)[PB]) {
    if ((GA_CONV_ABL & 2) && in_loop) return;
    T* As = lds + buf * kTile;
    T* Bs = As + BM * kLD;
#pragma unroll
    for (int p = 0; p < PA; ++p) *reinterpret_cast<uint4*>(As + (srow + RPP * p) * kLD + 8 * sq) = ra[p];
#pragma unroll
    for (int p = 0; p < PB; ++p) *reinterpret_cast<uint4*>(Bs + (srow + RPP * p) * kLD + 8 * sq) = rb[p];
  };

  f32x16 acc[JN][IM];   // [n block][m block]: D = Wfrag (rows n) x Xfrag (cols m)
#pragma unroll
  for (int j = 0; j < JN; ++j)
#pragma unroll
    for (int i = 0; i < IM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

  const int fr = lane & 31, fh = lane >> 5;   // fragment row inside a 32-block, k half
  auto mma_step = [&](int buf) {
    const T* As = lds + buf * kTile;
    const T* Bs = As + BM * kLD;
#pragma unroll
    for (int kk = 0; kk < kKC / 16; ++kk) {
      uint4 fa[IM], fb[JN];
#pragma unroll
      for (int i = 0; i < IM; ++i) fa[i] = *reinterpret_cast<const uint4*>(As + (wm * WM + i * 32 + fr) * kLD + kk * 16 + fh * 8);
#pragma unroll
      for (int j = 0; j < JN; ++j) fb[j] = *reinterpret_cast<const uint4*>(Bs + (wn * WN + j * 32 + fr) * kLD + kk * 16 + fh * 8);
#if GA_CONV_PRIO
      __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
      for (int j = 0; j < JN; ++j)
#pragma unroll
        for (int i = 0; i < IM; ++i) {
#if GA_CONV_ABL & 4
          asm volatile("" ::"v"(fb[j].x), "v"(fb[j].w), "v"(fa[i].x), "v"(fa[i].w));   // keep the fragment reads alive
#else
          acc[j][i] = Mma32<T>::run(fb[j], fa[i], acc[j][i]);
#endif
        }
#if GA_CONV_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
    }
  };
  // Straight-line loop body: every load group is issued unconditionally (beyond the last step the same step is
  // loaded again and never used) — a branch around a load group makes the compiler assume at the merge that
  // nothing younger is in flight and wait for vmcnt(0) where a counted wait would leave the next set's loads running.
  if (it0 < it1) {
    load_step(ra0, rb0);
    load_step(ra1, rb1);
    store_step(0, ra0, rb0);
  }
  __syncthreads();
  // Two steps per trip, no exit in the middle: a mid-loop break made the accumulators live in different registers on
  // the two paths and the compiler copied all of them (16 v_mov_b64 behind the last MFMA of every step).
  int it = it0;
  in_loop = true;
#define GA_CONV_SYNC() do { if (!(GA_CONV_ABL & 8)) __syncthreads(); } while (0)
  for (; it + 1 < it1; it += 2) {
    // even step: LDS buffer 0 holds step it, set 1 holds step it + 1 (in flight), set 0 is free
    load_step(ra0, rb0);         // step it + 2
    mma_step(0);
    store_step(1, ra1, rb1);
    GA_CONV_SYNC();
    // odd step: buffer 1 holds step it + 1, set 0 holds step it + 2 (in flight), set 1 is free
    load_step(ra1, rb1);         // step it + 3
    mma_step(1);
    store_step(0, ra0, rb0);
    GA_CONV_SYNC();
  }
  if (it < it1) {                // odd number of steps: the last one sits in buffer 0
    mma_step(0);
    __syncthreads();             // the epilogue reuses the buffers: every wave's fragment reads must be done
  }

  conv_epilogue<T, BM, BN, OUT_F32>(acc, lds, Y, part, tickets, bias, residual, a, m0, n0, split);
}

// ---- variant with the input patch kept in LDS (stride 1, tiles made of whole image rows) ---------------------------
// The kernel above stages a [BM pixels][64 channels] tile per tap: the nine taps of one channel chunk re-load and
// re-store (nearly) the same pixels nine times, and the ablation builds showed staging — global loads plus LDS stores,
// the stores alone a quarter of the time — costing as much as the MFMAs and not hiding behind them (a 128 x 64 step
// needs 96 B/clk of LDS stores to keep the matrix pipe fed; ds_write_b128 tops out near 79 B/clk/CU).  Here a tile is
// whole rows of one image (or whole small images); per 64-channel chunk the workgroup stages the rows it needs ONCE —
// (rows + 2) x (W + 2) pixels, zeros around the image — and all nine taps read their pixel fragments from that patch
// at shifted addresses.  A-side staging falls from 9 x BM rows to (rows + 2)(W + 2) rows per chunk (128 pixels of a
// 64-wide image: 1152 -> 264); the weight tile [BN][64] per (chunk, tap) is staged as before.
//   k order: chunk outer, tap inner.  Patch of chunk c + 1: nine 16-byte pieces per thread, all issued at tap 0 of
//   chunk c, stored after tap 8 between two barriers (single patch buffer: LDS stays small enough for two workgroups
//   per CU).  Weights: 3 or 9 register sets in rotation (a divisor of the 9 taps, so every index is static in the
//   unrolled chunk body), 3 or 6 steps in flight; LDS weight buffer = global step parity (runtime).
// 16-byte pieces per thread per chunk: patches of up to 288 pixels for 128-pixel tiles (a 64-wide image: 4 x 66 = 264),
// 224 for 64-pixel tiles (3 x 66 = 198)
// WIDE: one row of a 128-wide map (SDXL's top level: 3 x 130 = 390 pixels) needs 13 pieces = 416 patch rows
constexpr int patch_pieces(int BM, bool wide = false) { return wide ? 13 : (BM >= 128 ? 9 : 7); }

struct PatchGeom {
  int nseg, srows;   // the tile = nseg segments; the patch of a segment holds srows + 2 image rows of W + 2 pixels
};

// Which tiles the patch kernel takes:
//   (1) whole image rows:      BM = k W          -> one segment of k rows
//   (2) whole small images:    BM = k H W        -> k segments of H rows
//   (3) any run of BM consecutive pixels of ONE image (H W a multiple of BM): the run starts at column m0 mod W and wraps
//       around the row ends; its patch holds every image row it touches — at most (W + BM - 2) / W + 1 of them, plus the
//       halo.  This is what maps that are not whole rows of 128 or 64 pixels take: the 96 / 48 / 24-wide levels of the
//       768^2 configuration (BASELINE config 4), which round 2 left on the per-tap kernel at 5.7 - 6.5 x the algorithmic
//       bytes.  A 32-pixel fragment block that wraps a row end reads two patch rows (a few bank conflicts on those blocks).
__host__ __device__ inline bool patch_geometry(int BM, int H, int W, PatchGeom& g, bool wide = false) {
  if (BM % W == 0 && (H * W) % BM == 0) {
    g.nseg = 1;
    g.srows = BM / W;
  } else if (BM % (H * W) == 0) {
    g.nseg = BM / (H * W);
    g.srows = H;
  } else if ((H * W) % BM == 0) {
    g.nseg = 1;
    g.srows = (W + BM - 2) / W + 1;
  } else {
    return false;
  }
  return g.nseg * (g.srows + 2) * (W + 2) * (kKC / 8) <= patch_pieces(BM, wide) * kThreads;
}

template <typename T, int BM, int BN, bool OUT_F32, bool WIDE = false>
__global__ __launch_bounds__(kThreads, WIDE && BN > 64 ? 1 : 2) void conv3x3_patch_kernel(const T* __restrict__ X, const T* __restrict__ Wp,
                                                                    T* __restrict__ Y, float* __restrict__ part,
                                                                    unsigned* __restrict__ tickets,
                                                                    const T* __restrict__ bias,
                                                                    const T* __restrict__ residual, ConvArgs a) {
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int IM = WM / 32, JN = WN / 32;
  constexpr int QP = kKC / 8;                 // 16-byte pieces per row
  constexpr int RPP = kThreads / QP;          // rows staged per pass
  constexpr int PB = BN / RPP;                // weight staging passes
  constexpr int kPatchPieces = patch_pieces(BM, WIDE);
  constexpr int kPatchRows = kPatchPieces * RPP;          // 288 / 224
  constexpr int kPatch = kPatchRows * kLD;                 // elements of the patch buffer
  constexpr int kBt = BN * kLD;                            // one weight buffer
  constexpr int kCtile = BM * (BN + 8);
  constexpr int kLds = kPatch + 2 * kBt > kCtile ? kPatch + 2 * kBt : kCtile;
  __shared__ __attribute__((aligned(16))) T lds[kLds];
  T* patch = lds;
  T* wbuf = lds + kPatch;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int m0, n0, split;
  conv_args_resident(a);
  tile_of_workgroup<BM, BN>(a, m0, n0, split);
  const int cchunks = a.Cin / kKC;
  const int c_begin = split * a.steps_per, c_end = min(cchunks, c_begin + a.steps_per);   // chunks of this split

  // geometry and divisors from the host (launch_tile), as in the DMA form below; sdiv: this kernel also takes the shapes
  // whose dividends the multiply-high form cannot serve
  PatchGeom g{a.pg_nseg, a.pg_srows};
  const int hw = a.H * a.W;
  // geometry (3) (a run that is not made of whole rows): one segment that starts in the middle of a row
  const bool run = a.d_segpx.d == (unsigned)BM && BM != g.srows * a.W;
  int col0 = 0;
  if (run) {
    const int in_image = m0 - sdiv(m0, a.d_hw) * hw;
    col0 = in_image - sdiv(in_image, a.d_w) * a.W;
  }
  const int PW = a.W + 2, seg_px = (int)a.d_segpx.d, seg_rows = g.srows + 2, seg_pw = seg_rows * PW;
  const int npatch = g.nseg * seg_pw;

  constexpr unsigned kOob = 0x80000000u;
  const int srow = tid / QP, sq = tid % QP;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(X), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(Wp), 0, a.w_bytes, 0x00020000);

  // ---- patch staging: piece p of this thread = patch pixel q = srow + RPP * p, channels 8 sq .. 8 sq + 7 of the chunk
  unsigned pa_off[kPatchPieces];   // byte offset of that pixel's channel 8 sq in X, or kOob (halo / beyond the batch)
#pragma unroll
  for (int p = 0; p < kPatchPieces; ++p) {
    const int q = srow + RPP * p;
    const int seg = sdiv(q, a.d_segpw), rem = q - seg * seg_pw;
    const int pr = sdiv(rem, a.d_pw), pc = rem - pr * PW;
    const int mseg = m0 + seg * seg_px;                  // first output pixel of the segment
    const int b = sdiv(mseg, a.d_hw), y0 = sdiv(mseg - b * hw, a.d_w);
    const int iy = y0 + pr - 1, ix = pc - 1;
    const bool ok = q < npatch && mseg < a.M && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
    pa_off[p] = ok ? (unsigned)((((b * a.H + iy) * a.W + ix) * a.Cin + 8 * sq) * (int)sizeof(T)) : kOob;
  }
  uint4 rp[kPatchPieces];
  auto load_patch = [&](int chunk) {
    const unsigned step = (unsigned)(chunk * kKC * (int)sizeof(T));
#pragma unroll
    for (int p = 0; p < kPatchPieces; ++p) {
      const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, pa_off[p] + step, 0, 0);
      rp[p] = uint4{x[0], x[1], x[2], x[3]};
    }
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int p = 0; p < kPatchPieces; ++p)
      *reinterpret_cast<uint4*>(patch + (srow + RPP * p) * kLD + 8 * sq) = rp[p];
  };

  // ---- weight staging (as in the kernel above)
  unsigned b_off[PB];
#pragma unroll
  for (int p = 0; p < PB; ++p) {
    const int n = n0 + srow + RPP * p;
    b_off[p] = n < a.Cout ? w_row_bytes(a, n, sq) : kOob;
  }
  // Weight register sets: NS sets in rotation, NS | 9 so that every index is static in the unrolled chunk body; PD of
  // them in flight (prefetch distance in steps).  A 64-wide weight tile is 2 registers-quads per set: nine sets, six
  // steps ahead; a 128-wide one keeps three sets, three ahead (the accumulators need the registers).
  constexpr int NS = BN <= 64 ? 9 : 3, PD = BN <= 64 ? 6 : 3;
  uint4 rb[NS][PB];
  auto load_w = [&](int chunk, int tap, uint4 (&r)[PB]) {
    const unsigned step = w_step_bytes(a, tap, chunk);
#pragma unroll
    for (int p = 0; p < PB; ++p) {
      const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, b_off[p] + step, 0, 0);
      r[p] = uint4{x[0], x[1], x[2], x[3]};
    }
  };
  auto store_w = [&](int buf, const uint4 (&r)[PB]) {
    T* Bs = wbuf + buf * kBt;
#pragma unroll
    for (int p = 0; p < PB; ++p) *reinterpret_cast<uint4*>(Bs + (srow + RPP * p) * kLD + 8 * sq) = r[p];
  };

  f32x16 acc[JN][IM];
#pragma unroll
  for (int j = 0; j < JN; ++j)
#pragma unroll
    for (int i = 0; i < IM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

  // ---- fragment addresses: this lane's pixel of each 32-block -> top-left pixel of its 3x3 window in the patch
  const int fr = lane & 31, fh = lane >> 5;
  int pix_base[IM];   // element offset into the patch
#pragma unroll
  for (int i = 0; i < IM; ++i) {
    const int pm = wm * WM + i * 32 + lane_pixel(fr, a.lane_rot);
    const int seg = sdiv(pm, a.d_segpx), rem = pm - seg * seg_px + col0;   // col0: the column the tile's first pixel sits in
    const int r = sdiv(rem, a.d_w), c = rem - r * a.W;
    pix_base[i] = ((seg * seg_rows + r) * PW + c) * kLD + fh * 8;
  }
  auto mma_tap = [&](int buf, int ky, int kx) {
    const T* Bs = wbuf + buf * kBt;
    const int tap_off = (ky * PW + kx) * kLD;   // wave-uniform
#pragma unroll
    for (int kk = 0; kk < kKC / 16; ++kk) {
      uint4 fa[IM], fb[JN];
#pragma unroll
      for (int i = 0; i < IM; ++i) fa[i] = *reinterpret_cast<const uint4*>(patch + pix_base[i] + tap_off + kk * 16);
#pragma unroll
      for (int j = 0; j < JN; ++j) fb[j] = *reinterpret_cast<const uint4*>(Bs + (wn * WN + j * 32 + fr) * kLD + kk * 16 + fh * 8);
#pragma unroll
      for (int j = 0; j < JN; ++j)
#pragma unroll
        for (int i = 0; i < IM; ++i) {
#if GA_CONV_ABL & 4
          asm volatile("" ::"v"(fb[j].x), "v"(fb[j].w), "v"(fa[i].x), "v"(fa[i].w));   // keep the fragment reads alive
#else
          acc[j][i] = Mma32<T>::run(fb[j], fa[i], acc[j][i]);
#endif
        }
    }
  };

  if (c_begin < c_end) {
    // prologue: patch of the first chunk, weights of its taps 0 .. PD - 1
    load_patch(c_begin);
#pragma unroll
    for (int t = 0; t < PD; ++t) load_w(c_begin, t, rb[t % NS]);
    store_patch();
    store_w(0, rb[0]);
    __syncthreads();
    int s = 0;   // global step counter of this workgroup: weight buffer = s & 1
    for (int c = c_begin; c < c_end; ++c) {
      const int cn = min(c + 1, c_end - 1);   // next chunk (clamped: loaded again and never used after the last one)
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        // Step s = (chunk c, tap t): its weights sit in LDS buffer s & 1 (stored during step s - 1).  The set that takes
        // the loads of step s + PD last held step s + PD - NS <= s, stored at least one step ago: free.  At batch 1 the
        // weights are a cold stream from HBM and every step of look-ahead shows in the pipeline's time.
        if (!(GA_CONV_ABL & 1)) {
          if (t + PD < 9) load_w(c, t + PD, rb[(t + PD) % NS]);
          else load_w(cn, t + PD - 9, rb[(t + PD) % NS]);
        }
        if (t == 0 && !(GA_CONV_ABL & 16)) load_patch(cn);
        mma_tap(s & 1, t / 3, t % 3);
        if (!(GA_CONV_ABL & 2)) store_w((s + 1) & 1, rb[(t + 1) % NS]);     // step s + 1's weights (loaded PD - 1 steps ago)
        if (!(GA_CONV_ABL & 8)) __syncthreads();
        ++s;
      }
      // every wave is past tap 8: the patch may be replaced
      if (!(GA_CONV_ABL & 16)) store_patch();
      __syncthreads();
    }
  }
  conv_epilogue<T, BM, BN, OUT_F32>(acc, lds, Y, part, tickets, bias, residual, a, m0, n0, split);
}

// ---- the patch variant with the WEIGHTS going global -> LDS directly (buffer_load ... lds) ----------------------------------
// In the kernel above a weight tile makes three hops — global -> 2 or 4 register quads per thread -> ds_write_b128 -> LDS —
// with 3 to 9 register sets rotating to keep several k-steps in flight: 72 VGPRs for a 64-wide tile, an LDS write path that
// tops out near 79 B/clk, and 5.6 of the 32.8 us of the batch-3 320 -> 320 launch (r2 ablation).  Here the weight tile
// [BN][64] of a (chunk, tap) step is written by the DMA engine into a ring of NW slots (no staging registers, no ds_write):
//   * an LDS-DMA wave-instruction writes 1 KiB contiguously (8 rows of 128 bytes): rows cannot be padded, so the slot is
//     XOR-swizzled in 16-byte units — physical column = logical column ^ ((row >> 1) & 7), applied on the SOURCE address
//     and on the fragment reads (a 32-row fragment's ds_read_b128 lane groups then cover all 64 banks);
//   * PD = NW - 1 steps are in flight behind COUNTED s_waitcnt vmcnt(N) and raw s_barrier; the patch loads of the next chunk
//     (ordinary loads to registers, issued at tap 0) sit in the same in-order counter: N is a compile-time constant per
//     unrolled tap.  Past the last step the same step is loaded again into a slot nobody reads, so the counts stay static;
//   * every LDS access inside the loop is inline asm: a C++ LDS load or store makes the compiler wait vmcnt(0) first (any
//     pending LDS-DMA may alias it as far as its wait-count pass can tell) and the ring would drain at every step.
// The patch staging (registers -> padded rows) and the epilogue are the kernel's above.
// Diagnostic build only (-DGA_CONV_STAMPS, `make stamps`; tools/micro/conv_stamps.py): wave 0 of every workgroup of
// conv3x3_patch_dma_kernel reads the shader clock at its phase boundaries and adds up, over its k-steps, the time in front of the
// step barrier, in the refill issue, in the MFMA body and in the patch hand-over.  The fences forbid overlaps the product has:
// read shares.
#if defined(GA_CONV_STAMPS)
__device__ unsigned long long* g_conv_stamps = nullptr;   // [workgroup][12]
struct ConvStamps {
  unsigned long long t[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long last = 0;
  __device__ __forceinline__ unsigned long long now() {
    unsigned long long v = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory");
    __builtin_amdgcn_sched_barrier(0);
#endif
    return v;
  }
  __device__ __forceinline__ unsigned long long real() {
    unsigned long long v = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory");
    __builtin_amdgcn_sched_barrier(0);
#endif
    return v;
  }
  __device__ __forceinline__ void at(int i) { t[i] = last = now(); }
  __device__ __forceinline__ void add(int i) {   // time since the previous stamp goes to sum i
    const unsigned long long v = now();
    t[i] += v - last;
    last = v;
  }
  __device__ __forceinline__ void flush() {
    if (threadIdx.x == 0 && g_conv_stamps != nullptr)
      for (int i = 0; i < 12; ++i) g_conv_stamps[(size_t)blockIdx.x * 12 + i] = t[i];
  }
};
#define GA_CSTAMP(x) x
#else
#define GA_CSTAMP(x) ((void)0)
#endif

__device__ __forceinline__ void conv_lds_read128(u32x4_t& dst, unsigned byte_address) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(byte_address) : "memory");
#else
  dst = u32x4_t{byte_address, 0u, 0u, 0u};
#endif
}
__device__ __forceinline__ void conv_lds_write128(unsigned byte_address, u32x4_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("ds_write_b128 %0, %1" ::"v"(byte_address), "v"(v) : "memory");
#endif
}
// The counted vmcnt waits below assume the ISSUE ORDER written in the source: a refill's DMA pieces, THEN the patch loads.  The patch
// loads are plain (side-effect-free) buffer loads, which the compiler may move: in a variant of this loop (round 4, the step barrier
// moved in front of the last sub-step) it put them BETWEEN the two DMA pieces of the refill — piece 1 became younger than the patch
// loads, the wait for "everything but (PD - 1) refills and the patch loads" no longer covered it, and the n half that piece fills
// was read before it had landed (64 x 64 tile with many k-slices: one launch in two wrong, tools/micro/conv_debug.py).  The
// shipped loop happened to come out in source order; an empty asm that clobbers memory now pins it.
__device__ __forceinline__ void conv_order_fence() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" ::: "memory");
#endif
}
template <int N>
__device__ __forceinline__ void conv_wait_vmcnt() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}
template <int N>
__device__ __forceinline__ void conv_wait_lgkmcnt() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
#endif
}

// DEEP: twice the ring (64 KB of weight slots, one workgroup per CU) for the WEIGHT-BOUND launches — the 16 x 16 and 8 x 8
// levels, where a launch is a 15 - 59 MB weight stream against <= 1024 pixels.  A plan there puts about one workgroup on each CU;
// with the 32 KB ring that is 24 KB of weights in flight per CU, and the stream ran at 1.3 - 1.8 TB/s (bytes in flight / memory
// latency under load, not the MFMAs, set the time: 29.5 MB took 16.8 - 22.4 us cold).
template <typename T, int BM, int BN, bool OUT_F32, bool WIDE = false, bool DEEP = false>
__global__ __launch_bounds__(kThreads, WIDE || DEEP ? 1 : 2) void conv3x3_patch_dma_kernel(
    const T* __restrict__ X, const T* __restrict__ Wp, T* __restrict__ Y, float* __restrict__ part,
    unsigned* __restrict__ tickets, const T* __restrict__ bias, const T* __restrict__ residual, ConvArgs a) {
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int IM = WM / 32, JN = WN / 32;
  constexpr int QP = kKC / 8;                 // 16-byte pieces per row
  constexpr int RPP = kThreads / QP;          // rows staged per pass
  constexpr int kPatchPieces = patch_pieces(BM, WIDE);
  constexpr int kPatchRows = kPatchPieces * RPP;
  constexpr int kPatch = kPatchRows * kLD;                 // elements of the patch buffer (padded rows)
  constexpr int kSlot = BN * kKC;                          // elements of one weight ring slot (128-byte rows, swizzled)
  constexpr int NW = (BN <= 64 ? 4 : 2) * (DEEP ? 2 : 1);  // ring slots: 32 KB (two workgroups per CU), DEEP: 64 KB
  constexpr int PD = NW - 1;                               // steps in flight
  constexpr int IPW = BN / 32;                             // LDS-DMA wave-instructions per slot and wave
  constexpr int kCtile = BM * (BN + 8);
  constexpr int kLds = kPatch + NW * kSlot > kCtile ? kPatch + NW * kSlot : kCtile;
  static_assert((kPatch * (int)sizeof(T)) % 256 == 0, "the ring starts on a bank-row boundary (the swizzle assumes it)");
  __shared__ __attribute__((aligned(1024))) T lds[kLds];
  T* ring = lds + kPatch;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
#if defined(GA_CONV_STAMPS)
  ConvStamps stm;
  stm.t[8] = stm.real();
  stm.at(0);
#endif
  int m0, n0, split;
  conv_args_resident(a);
  tile_of_workgroup<BM, BN>(a, m0, n0, split);
  const int cchunks = a.Cin / kKC;
  const int c_begin = split * a.steps_per, c_end = min(cchunks, c_begin + a.steps_per);   // chunks of this split

  // tile geometry and divisors from the host (launch_tile): no runtime division in this prologue
  PatchGeom g{a.pg_nseg, a.pg_srows};
  const int hw = a.H * a.W;
  const bool run = a.d_segpx.d == (unsigned)BM && BM != g.srows * a.W;   // seg_px = BM only for tiles that are runs of pixels
  int col0 = 0;
  if (run) {
    const int in_image = m0 - fdiv(m0, a.d_hw) * hw;
    col0 = in_image - fdiv(in_image, a.d_w) * a.W;
  }
  const int PW = a.W + 2, seg_px = (int)a.d_segpx.d, seg_rows = g.srows + 2, seg_pw = seg_rows * PW;

  constexpr unsigned kOob = 0x80000000u;
  const int srow = tid / QP, sq = tid % QP;
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(X), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(Wp), 0, a.w_bytes, 0x00020000);
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) void*)lds);

  // ---- patch staging (as above; the LDS stores are asm).  Piece p of this thread = patch pixel q = srow + RPP p: its
  // (segment, patch row, patch column) advance by RPP pixels per piece — carries, no division per piece; 24-bit multiplies
  // (full rate; the host checked pixel count and row bytes < 2^24).  Called AFTER the first weight DMAs are issued: the
  // arithmetic runs under their latency.
  unsigned pa_off[kPatchPieces];
  auto patch_offsets = [&]() {
    int seg = fdiv(srow, a.d_segpw);
    const int rem0 = srow - seg * seg_pw;
    int pr = fdiv(rem0, a.d_pw), pc = rem0 - pr * PW;
    const int b0 = fdiv(m0, a.d_hw), y00 = fdiv(m0 - b0 * hw, a.d_w);   // wave-uniform: image and row of the tile's first pixel
    const unsigned row_bytes = (unsigned)(a.Cin * (int)sizeof(T));
    const int SH = a.up ? a.H >> 1 : a.H, SW = a.up ? a.W >> 1 : a.W;
#pragma unroll
    for (int p = 0; p < kPatchPieces; ++p) {
      const int mseg = m0 + seg * seg_px;
      const int b = b0 + seg;     // several segments = whole images per segment; one segment: seg = 0 for every live piece
      const int iy = y00 + pr - 1, ix = pc - 1;
      const bool ok = seg < g.nseg && mseg < a.M && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      const int sy = a.up ? iy >> 1 : iy, sx = a.up ? ix >> 1 : ix;
      const unsigned pix = __umul24(__umul24((unsigned)b, (unsigned)SH) + (unsigned)sy, (unsigned)SW) + (unsigned)sx;
      pa_off[p] = ok ? __umul24(pix, row_bytes) + (unsigned)(8 * sq * (int)sizeof(T)) : kOob;
      pc += a.pg_dpc;
      pr += a.pg_dpr;
      if (pc >= PW) { pc -= PW; ++pr; }
      if (pr >= seg_rows) { pr -= seg_rows; ++seg; }
      if (pr >= seg_rows) { pr -= seg_rows; ++seg; }
    }
  };
  u32x4_t rp[kPatchPieces];
  auto load_patch = [&](int chunk) {
    const unsigned step = (unsigned)(chunk * kKC * (int)sizeof(T));
#pragma unroll
    for (int p = 0; p < kPatchPieces; ++p) rp[p] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, pa_off[p] + step, 0, 0);
  };
  const unsigned patch_st = lds0 + (unsigned)((srow * kLD + 8 * sq) * (int)sizeof(T));
  auto store_patch = [&]() {
#pragma unroll
    for (int p = 0; p < kPatchPieces; ++p) conv_lds_write128(patch_st + (unsigned)(RPP * p * kLD * (int)sizeof(T)), rp[p]);
  };

  // ---- weight DMA: instruction q of this wave fills rows 8 g .. 8 g + 7 of the slot, g = wave + 4 q; lane l lands at row
  // 8 g + (l >> 3), physical 16-byte column l & 7, and therefore fetches logical column (l & 7) ^ swizzle(row)
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  unsigned w_off[IPW];
#pragma unroll
  for (int q = 0; q < IPW; ++q) {
    const int row = 8 * (wave + 4 * q) + (lane >> 3);
    const int col = (lane & 7) ^ ((row >> 1) & 7);
    w_off[q] = w_row_bytes(a, min(n0 + row, a.Cout - 1), col);   // rows past Cout repeat the last one (never stored)
  }
  auto issue_w = [&](int chunk, int tap, int slot) {
    const unsigned step = w_step_bytes(a, tap, chunk);
#pragma unroll
    for (int q = 0; q < IPW; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)(ring + slot * kSlot + (wave_u + 4 * q) * 8 * kKC),
                                               16, (int)w_off[q], (int)step, 0, 0);
  };

  f32x16 acc[JN][IM];
#pragma unroll
  for (int j = 0; j < JN; ++j)
#pragma unroll
    for (int i = 0; i < IM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

  // ---- fragment addresses (LDS byte addresses)
  const int fr = lane & 31, fh = lane >> 5;
  unsigned pix_adr[IM], b_adr[JN], b_sw[JN];
#pragma unroll
  for (int i = 0; i < IM; ++i) {
    const int pm = wm * WM + i * 32 + lane_pixel(fr, a.lane_rot);
    const int seg = fdiv(pm, a.d_segpx), rem = pm - seg * seg_px + col0;
    const int r = fdiv(rem, a.d_w), c = rem - r * a.W;
    pix_adr[i] = lds0 + (unsigned)((((seg * seg_rows + r) * PW + c) * kLD + fh * 8) * (int)sizeof(T));
  }
#pragma unroll
  for (int j = 0; j < JN; ++j) {
    const int row = wn * WN + j * 32 + fr;
    b_adr[j] = lds0 + (unsigned)((kPatch + row * kKC) * (int)sizeof(T));
    b_sw[j] = (unsigned)(((row >> 1) & 7) ^ fh);
  }
  auto mma_tap = [&](int slot, int ky, int kx) {
    const unsigned tap_off = (unsigned)((ky * PW + kx) * kLD * (int)sizeof(T));   // wave-uniform
    const unsigned slot_off = (unsigned)(slot * kSlot * (int)sizeof(T));
    // fragments of sub-step kk + 1 are requested before the MFMAs of kk (LDS returns in order: a counted lgkmcnt tells when
    // kk's operands are in); the 128 x 128 tile has no registers for a second set under the two-workgroups-per-CU cap
    constexpr bool kAhead = IM * JN <= 2;
    constexpr int NSET = kAhead ? 2 : 1;
    u32x4_t fa[NSET][IM], fb[NSET][JN];
    auto request = [&](int kk, int set) {
#pragma unroll
      for (int i = 0; i < IM; ++i) conv_lds_read128(fa[set][i], pix_adr[i] + tap_off + (unsigned)(kk * 16 * (int)sizeof(T)));
#pragma unroll
      for (int j = 0; j < JN; ++j) conv_lds_read128(fb[set][j], b_adr[j] + slot_off + 16u * ((2u * kk) ^ b_sw[j]));
    };
    if constexpr (kAhead) request(0, 0);
#pragma unroll
    for (int kk = 0; kk < kKC / 16; ++kk) {
      constexpr int dummy = 0;
      (void)dummy;
      if constexpr (kAhead) {
        if (kk + 1 < kKC / 16) {
          request(kk + 1, (kk + 1) & 1);
          conv_wait_lgkmcnt<IM + JN>();
        } else {
          conv_wait_lgkmcnt<0>();
        }
      } else {
        request(kk, 0);
        conv_wait_lgkmcnt<0>();
      }
      __builtin_amdgcn_sched_barrier(0);
      const int set = kAhead ? (kk & 1) : 0;
#pragma unroll
      for (int j = 0; j < JN; ++j)
#pragma unroll
        for (int i = 0; i < IM; ++i)
          acc[j][i] = Mma32<T>::run(__builtin_bit_cast(uint4, fb[set][j]), __builtin_bit_cast(uint4, fa[set][i]), acc[j][i]);
      if constexpr (!kAhead) __builtin_amdgcn_sched_barrier(0);   // the next sub-step's asm reads overwrite the fragment set
    }
  };

  if (c_begin < c_end) {
    // prologue: patch of the first chunk (registers -> LDS), weights of its first PD steps into ring slots 0 .. PD - 1
#pragma unroll
    for (int t = 0; t < PD; ++t) issue_w(c_begin, t, t);   // first: they need the tile's n0 only
    conv_order_fence();
    patch_offsets();
    load_patch(c_begin);
    store_patch();                 // the compiler waits for rp[] itself (the youngest loads: the DMAs above have landed by then)
    conv_wait_lgkmcnt<0>();
    GA_CSTAMP(stm.at(1));
    int s = 0;                     // global step counter of this workgroup: ring slot = s % NW
    for (int c = c_begin; c < c_end; ++c) {
      const int cn = min(c + 1, c_end - 1);   // next chunk (clamped: loaded again and never used after the last one)
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        // younger than step s's DMA at this point: the DMAs of steps s + 1 .. s + PD - 1, and the next chunk's patch loads
        // when they were issued in between (at tap 0, behind that step's DMA): taps 1 .. PD
        constexpr int kPatchInFlight = kPatchPieces;
        if (t >= 1 && t <= PD) conv_wait_vmcnt<(PD - 1) * IPW + kPatchInFlight>();
        else conv_wait_vmcnt<(PD - 1) * IPW>();
        __builtin_amdgcn_s_barrier();   // step s's weights (and, at tap 0, the patch) are in LDS for everyone; step s - 1 is read
        GA_CSTAMP(stm.add(2));
        {  // refill the slot step s - 1 occupied with step s + PD
          const int tt = t + PD;
          if (tt < 9) issue_w(c, tt, (s + PD) % NW);
          else issue_w(cn, tt - 9, (s + PD) % NW);
        }
        conv_order_fence();
        if (t == 0) load_patch(cn);
        conv_order_fence();
        GA_CSTAMP(stm.add(3));
        mma_tap(s % NW, t / 3, t % 3);
        GA_CSTAMP(stm.add(4));
        ++s;
      }
      // every wave is past tap 8 of this chunk after the barrier: the patch may be replaced
      __builtin_amdgcn_s_barrier();
      store_patch();
      conv_wait_lgkmcnt<0>();
      GA_CSTAMP(stm.add(5));
    }
#if defined(GA_CONV_STAMPS)
    stm.t[6] = (unsigned long long)s;
#endif
  }
  conv_wait_vmcnt<0>();
  __syncthreads();
  GA_CSTAMP(stm.at(7));
  conv_epilogue<T, BM, BN, OUT_F32>(acc, lds, Y, part, tickets, bias, residual, a, m0, n0, split);
#if defined(GA_CONV_STAMPS)
  conv_wait_vmcnt<0>();
  stm.at(10);
  stm.t[9] = stm.real();
  stm.flush();
#endif
}

// pre-pack: W [Cout][Cin][3][3] in whatever strides the framework holds (element strides given) ->
//   Wp[t = ky*3+kx][nb = n / 64][cb = c / 64][r = n % 64][k = c % 64]   (N rounded up to 64 rows, the extra rows zero)
//   forward : n = cout, c = cin : W[cout][cin][ky][kx]
//   backward: n = cin, c = cout : W[cout][cin][2-ky][2-kx]   (dX = conv(dY, flipped, transposed))
template <typename T>
__global__ __launch_bounds__(kThreads) void conv_pack_kernel(const T* __restrict__ W, T* __restrict__ Wp, int Cout, int Cin,
                                                             long long s_o, long long s_i, long long s_y, long long s_x,
                                                             int transpose) {
  const long long idx = (long long)blockIdx.x * kThreads + threadIdx.x;
  const int N = transpose ? Cin : Cout, C = transpose ? Cout : Cin;
  const int NB = (N + 63) / 64, CB = C / kKC;
  const long long total = 9LL * NB * 64 * C;
  if (idx >= total) return;
  const int k = (int)(idx & 63), r = (int)((idx >> 6) & 63);
  const long long blk = idx >> 12;
  const int cb = (int)(blk % CB);
  const int nb = (int)((blk / CB) % NB);
  const int t = (int)(blk / ((long long)CB * NB));
  const int n = nb * 64 + r, c = cb * kKC + k;
  const int ky = t / 3, kx = t - 3 * ky;
  const int co = transpose ? c : n, ci = transpose ? n : c;
  const int sy = transpose ? 2 - ky : ky, sx = transpose ? 2 - kx : kx;
  Wp[idx] = n < N ? W[co * s_o + ci * s_i + sy * s_y + sx * s_x] : Traits<T>::from_f32(0.f);
}

struct Plan {
  int bm, bn, splits;
};

// Tile and split choice for shapes without a measured plan (conv_plans.json on the host side holds the measured ones
// for the SD-1.x UNet).  A time estimate in microseconds per candidate:
//   compute  = flops x padding waste x fill / rate(tile): about 480 workgroups fill the chip (just under two per CU);
//              fewer leave CUs idle, more run in rounds of 512; the smaller tiles run at a lower rate (operand reuse);
//   split-K  = the f32 partial slabs written and read back (splits x M x N x 8 bytes at ~4 TB/s) + the sum launch —
//              what makes a split pay at M = 256 (30 MB of slabs for 12 splits) and hurt at M = 49 152 (500 MB for 4: the
//              first rule, a flat 1.5 % per split, chose 4 splits for SDXL's 128x128 level and doubled its time).
// At most 16 splits, at least 3 k-steps per slice; 320 outputs prefer 64-wide tiles (no channel padding).
Plan choose_plan(int M, int N, int steps, int H, int W, int stride) {
  const int cands[3][2] = {{128, 128}, {128, 64}, {64, 64}};
  const double rate[3] = {700.0, 650.0, 570.0};   // TFLOP/s of a well-filled launch per tile shape (profiles/r2_conv_tune.txt)
  const double flops = 2.0 * (double)M * (double)N * (double)steps * kKC;
  Plan best{64, 64, 1};
  double best_us = 1e30;
  for (int ci = 0; ci < 3; ++ci) {
    const int bm = cands[ci][0], bn = cands[ci][1];
    const long long tm = (M + bm - 1) / bm, tn = (N + bn - 1) / bn, tiles = tm * tn;
    const double waste = (double)(tm * bm) * (double)(tn * bn) / ((double)M * (double)N);
    // tiles the patch-in-LDS kernel cannot take fall to the per-tap kernel (9 x the A-side staging): about 0.7 of the rate
    PatchGeom pg;
    const bool patchable = stride == 1 && W >= 8 &&
                           (patch_geometry(bm, H, W, pg) || ((bn <= 64 || bm == 128) && patch_geometry(bm, H, W, pg, true)));
    const double tile_rate = rate[ci] * (patchable ? 1.0 : 0.7);
    const int max_s = steps / 3 < 16 ? (steps / 3 < 1 ? 1 : steps / 3) : 16;
    for (int s = 1; s <= max_s; ++s) {
      const double wgs = (double)tiles * s;
      const double fill = wgs <= 512.0 ? 480.0 / (wgs < 480.0 ? wgs : 480.0) : ((double)((long long)((wgs + 511) / 512)) * 512.0) / wgs;
      const double compute_us = flops * waste * fill / (tile_rate * 1e6);
      const double split_us = s > 1 ? (double)s * (double)M * (double)N * 8.0 / 4.0e6 + 3.0 : 0.0;   // slabs out and back + the hand-off
      const double us = compute_us + split_us;
      if (us < best_us) {
        best_us = us;
        best = Plan{bm, bn, s};
      }
    }
  }
  return best;
}

// Build-time switches of the A/B harness (tools/conv_tune.py builds variant libraries with -D...; the product library takes the
// defaults — round 3 read these from the environment at run time):
//   GA_CONV_DMA  1: weights through the LDS-DMA ring (conv3x3_patch_dma_kernel); 0: the register-staged patch kernel
//   GA_CONV_DEEP 1: the 64 KB weight ring for the weight-bound launches; 0: never
//   GA_CONV_V1   1: every shape on the per-tap staging kernel
#ifndef GA_CONV_DMA
#define GA_CONV_DMA 1
#endif
#ifndef GA_CONV_DEEP
#define GA_CONV_DEEP 1
#endif
#ifndef GA_CONV_V1
#define GA_CONV_V1 0
#endif
constexpr bool use_dma() { return GA_CONV_DMA != 0; }
constexpr bool deep_ring() { return GA_CONV_DEEP != 0; }
constexpr bool force_v1() { return GA_CONV_V1 != 0; }

template <typename T, int BM, int BN>
int launch_tile(const T* X, const T* Wp, T* Y, float* ws, unsigned* tickets, const T* bias, const T* residual,
                const ConvArgs& a_in, int splits, hipStream_t s) {
  ConvArgs a = a_in;
  a.lane_rot = 0;
  a.splits = splits;
  a.tm = (a.M + BM - 1) / BM;
  a.tn = (a.Cout + BN - 1) / BN;
  if (splits > 1 && (long long)splits * a.tm * a.tn * BM * BN * 4 >= (1LL << 31)) return GA_ERR_SHAPE;   // 32-bit slab offsets
  {  // bytes that reach the fabric if each XCD fetches what its run of workgroups shares once
    const double xb = (double)a.B * a.H * a.W * a.Cin, wb = (a.pad ? 9.0 : 1.0) * a.Cin * a.Cout;
    const double m_first = wb + xb * (a.tn * splits < 8 ? a.tn * splits : 8), n_first = xb + wb * (a.tm < 8 ? a.tm : 8);
    a.n_fastest = n_first < m_first ? 1 : 0;
  }
  dim3 grid((unsigned)(a.tm * a.tn * splits));
  PatchGeom pg{1, 1};
  // 8x8 maps: the register-staged patch kernel never took them (the halo makes the patch 100 pixels for 64, the depth splits
  // only by chunks), and the 32 KB DMA ring measured the same as the per-tap kernel (15.4 vs 15.1 us): these launches are a
  // 29.5 - 59 MB weight stream — what they need is bytes in flight, i.e. the DEEP ring below
  const bool geom_ok = a.pad == 1 && a.stride == 1 && (a.W >= 16 || (a.W >= 8 && use_dma() && deep_ring())) && !force_v1();
  const bool patch = geom_ok && patch_geometry(BM, a.H, a.W, pg);
  const bool patch_wide = geom_ok && !patch && (BN <= 64 || BM == 128) && patch_geometry(BM, a.H, a.W, pg, true);
  // Tiles that are RUNS of pixels (start mid-row, wrap around the row ends: the 96 / 48 / 24-wide maps of the 768^2
  // configuration) stay on the register-staged patch kernel: the DMA-ring form measured 6 - 20 % slower on the 48-wide level
  // and up to 17 % on the 24-wide one (profiles/r3_conv_tune_sd21.txt; equal at 96 and 12), and equal-or-faster only where a
  // tile is whole rows or whole images (every level of the 512^2 configuration)
  const bool run_geom = BM % a.W != 0 && BM % (a.H * a.W) != 0;
  const bool dma_wanted = use_dma() && !(run_geom && a.W >= 16);   // below 16 only the DMA form has a patch variant at all (see geom_ok)
  bool fast_ok = true;
  {  // everything the kernels would otherwise divide for (FastDiv: exact for the dividends named here)
    const unsigned long long wgs = (unsigned long long)a.tm * a.tn * splits;
    bool map_ok = true;
    a.d_tm = make_fastdiv(a.tm, wgs, map_ok);
    a.d_tn = make_fastdiv(a.tn, wgs, map_ok);
    if (!map_ok) return GA_ERR_SHAPE;   // more than 2^32 / max(tm, tn) workgroups: no such launch exists below the 2^31-element limits
    a.pg_nseg = pg.nseg;
    a.pg_srows = pg.srows;
    const long long seg_px = run_geom ? BM : (long long)pg.srows * a.W, seg_pw = (long long)(pg.srows + 2) * (a.W + 2);
    a.d_w = make_fastdiv(a.W, (unsigned long long)a.H * a.W + seg_px + a.W, fast_ok);
    a.d_hw = make_fastdiv((long long)a.H * a.W, (unsigned long long)a.M + 2ull * BM, fast_ok);
    a.d_pw = make_fastdiv(a.W + 2, (unsigned long long)seg_pw + 512, fast_ok);
    a.d_segpw = make_fastdiv(seg_pw, 1024, fast_ok);
    a.d_segpx = make_fastdiv(seg_px, 2ull * BM, fast_ok);
    a.pg_dpr = 32 / (a.W + 2);
    a.pg_dpc = 32 % (a.W + 2);
    if ((long long)a.B * a.H * a.W >= (1 << 24) || (long long)a.Cin * 2 >= (1 << 24)) fast_ok = false;   // 24-bit multiplies
    bool tap_ok = true;   // the per-tap kernel: its own two divisors (sdiv: it divides properly where these cannot serve)
    a.d_howo = make_fastdiv((long long)a.Ho * a.Wo, (unsigned long long)a.M + 2ull * BM, tap_ok);
    a.d_wo = make_fastdiv(a.Wo, (unsigned long long)a.Ho * a.Wo, tap_ok);
  }
  // the DMA patch kernel divides by multiplication only: a shape whose dividends overflow that (batch x pixels^2 >= 2^32: sixteen
  // 128 x 128 maps in one launch) takes the register-staged kernels, which divide properly
  const bool dma = dma_wanted && fast_ok;
  if (a.up && !((patch || patch_wide) && dma)) return GA_ERR_SHAPE;   // only the patch-DMA kernel gathers from the half-size map
  if (patch_wide) {   // the 13-piece instantiation: one row of a 128-wide map, or a run of a 96- / 48-wide map (geometry 3)
    a.steps_per = (a.Cin / kKC + splits - 1) / splits;
    if constexpr (BN <= 64 || BM == 128) {
      if (dma) {
        if (splits == 1)
          hipLaunchKernelGGL((conv3x3_patch_dma_kernel<T, BM, BN, false, true>), grid, dim3(kThreads), 0, s, X, Wp, Y,
                             (float*)nullptr, (unsigned*)nullptr, bias, residual, a);
        else
          hipLaunchKernelGGL((conv3x3_patch_dma_kernel<T, BM, BN, true, true>), grid, dim3(kThreads), 0, s, X, Wp, Y, ws,
                             tickets, bias, residual, a);
      } else if (splits == 1) {
        hipLaunchKernelGGL((conv3x3_patch_kernel<T, BM, BN, false, true>), grid, dim3(kThreads), 0, s, X, Wp, Y,
                           (float*)nullptr, (unsigned*)nullptr, bias, residual, a);
      } else {
        hipLaunchKernelGGL((conv3x3_patch_kernel<T, BM, BN, true, true>), grid, dim3(kThreads), 0, s, X, Wp, Y, ws, tickets,
                           bias, residual, a);
      }
      return check_launch();
    }
  }
  if (patch) {
    a.steps_per = (a.Cin / kKC + splits - 1) / splits;   // this variant splits the depth by channel chunks
    a.lane_rot = a.W == 16 ? 2 : 0;
  }
  if (patch && dma) {
    // weight-bound launch (few pixels against tens of MB of weights) with about one workgroup per CU: the deep ring
    const bool deep = a.M <= 1024 && (long long)a.tm * a.tn * splits <= 384 && deep_ring();
    if (deep) {
      if (splits == 1)
        hipLaunchKernelGGL((conv3x3_patch_dma_kernel<T, BM, BN, false, false, true>), grid, dim3(kThreads), 0, s, X, Wp, Y,
                           (float*)nullptr, (unsigned*)nullptr, bias, residual, a);
      else
        hipLaunchKernelGGL((conv3x3_patch_dma_kernel<T, BM, BN, true, false, true>), grid, dim3(kThreads), 0, s, X, Wp, Y, ws,
                           tickets, bias, residual, a);
    } else if (splits == 1) {
      hipLaunchKernelGGL((conv3x3_patch_dma_kernel<T, BM, BN, false>), grid, dim3(kThreads), 0, s, X, Wp, Y, (float*)nullptr,
                         (unsigned*)nullptr, bias, residual, a);
    } else {
      hipLaunchKernelGGL((conv3x3_patch_dma_kernel<T, BM, BN, true>), grid, dim3(kThreads), 0, s, X, Wp, Y, ws, tickets, bias,
                         residual, a);
    }
    return check_launch();
  }
  if (splits == 1) {
    if (patch)
      hipLaunchKernelGGL((conv3x3_patch_kernel<T, BM, BN, false>), grid, dim3(kThreads), 0, s, X, Wp, Y, (float*)nullptr,
                         (unsigned*)nullptr, bias, residual, a);
    else
      hipLaunchKernelGGL((conv3x3_kernel<T, BM, BN, false>), grid, dim3(kThreads), 0, s, X, Wp, Y, (float*)nullptr,
                         (unsigned*)nullptr, bias, residual, a);
  } else {
    if (patch)
      hipLaunchKernelGGL((conv3x3_patch_kernel<T, BM, BN, true>), grid, dim3(kThreads), 0, s, X, Wp, Y, ws, tickets, bias,
                         residual, a);
    else
      hipLaunchKernelGGL((conv3x3_kernel<T, BM, BN, true>), grid, dim3(kThreads), 0, s, X, Wp, Y, ws, tickets, bias,
                         residual, a);
  }
  return check_launch();
}

template <typename T>
int conv_t(const void* X, const void* Wp, void* Y, float* ws, unsigned* tickets, const void* bias, const void* residual,
           ConvArgs a, int bm, int bn, int splits, hipStream_t s) {
  a.steps = (a.pad ? 9 : 1) * a.Cin / kKC;
  a.steps_per = (a.steps + splits - 1) / splits;
  const T* x = (const T*)X;
  const T* w = (const T*)Wp;
  const T* b = (const T*)bias;
  const T* r = (const T*)residual;
  if (bm == 128 && bn == 128) return launch_tile<T, 128, 128>(x, w, (T*)Y, ws, tickets, b, r, a, splits, s);
  if (bm == 128 && bn == 64) return launch_tile<T, 128, 64>(x, w, (T*)Y, ws, tickets, b, r, a, splits, s);
  if (bm == 64 && bn == 64) return launch_tile<T, 64, 64>(x, w, (T*)Y, ws, tickets, b, r, a, splits, s);
  return GA_ERR_SHAPE;
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// f32 slab floats of a split-K launch: every slice of every (padded) tile keeps BM x BN accumulators
long long conv_workspace_floats(long long M, int N, int bm, int bn, int splits) {
  if (splits <= 1) return 0;
  return (long long)splits * ((M + bm - 1) / bm) * ((N + bn - 1) / bn) * bm * bn;
}

}  // namespace

extern "C" long long ga_splitk_workspace_floats(int64_t M, int N, int bm, int bn, int splits) {
  return conv_workspace_floats(M, N, bm, bn, splits);
}

extern "C" int ga_conv3x3_plan(int B, int H, int W, int Cin, int Cout, int stride, int* bm, int* bn, int* splits,
                               long long* workspace_floats) {
  if (!bm || !bn || !splits || !workspace_floats) return GA_ERR_NULL;
  if (B < 1 || H < 1 || W < 1 || Cin < kKC || Cin % kKC != 0 || Cout < 8 || Cout % 8 != 0 || (stride != 1 && stride != 2))
    return GA_ERR_SHAPE;
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const Plan p = choose_plan(B * Ho * Wo, Cout, 9 * Cin / kKC, H, W, stride);
  *bm = p.bm;
  *bn = p.bn;
  *splits = p.splits;
  *workspace_floats = conv_workspace_floats(B * Ho * Wo, Cout, p.bm, p.bn, p.splits);
  return GA_OK;
}

extern "C" long long ga_conv3x3_packed_elems(int N, int C) { return 9LL * ((N + 63) / 64) * 64 * C; }

extern "C" int ga_conv3x3_pack_weights(const void* W, void* Wp, int Cout, int Cin, int64_t stride_o, int64_t stride_i,
                                       int64_t stride_y, int64_t stride_x, int transpose_flip, int dtype,
                                       ga_stream_t stream) {
  if (!W || !Wp) return GA_ERR_NULL;
  if (Cout < 1 || Cin < 1 || (transpose_flip ? Cout : Cin) % kKC != 0) return GA_ERR_SHAPE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const long long total = ga_conv3x3_packed_elems(transpose_flip ? Cin : Cout, transpose_flip ? Cout : Cin);
  dim3 grid((unsigned)((total + kThreads - 1) / kThreads));
  switch (dtype) {
    case GA_F16:
      hipLaunchKernelGGL(conv_pack_kernel<_Float16>, grid, dim3(kThreads), 0, s, (const _Float16*)W, (_Float16*)Wp, Cout, Cin,
                         (long long)stride_o, (long long)stride_i, (long long)stride_y, (long long)stride_x, transpose_flip);
      break;
    case GA_BF16:
      hipLaunchKernelGGL(conv_pack_kernel<bf16_t>, grid, dim3(kThreads), 0, s, (const bf16_t*)W, (bf16_t*)Wp, Cout, Cin,
                         (long long)stride_o, (long long)stride_i, (long long)stride_y, (long long)stride_x, transpose_flip);
      break;
    default:
      return GA_ERR_DTYPE;
  }
  return check_launch();
}

extern "C" int ga_conv3x3_gn_blocks(int H, int W, int Cout, int groups, int bm, int bn) {
  // partial blocks per image ga_conv3x3_nhwc_gn writes (0: it does not serve the shape): whole m tiles per image, 8 <= channels
  // per group <= bn, at most 128 blocks (what ga_group_norm_apply folds)
  if (H < 1 || W < 1 || groups < 1 || Cout % groups != 0 || (bm != 64 && bm != 128) || (bn != 64 && bn != 128)) return 0;
  const int hw = H * W, cg = Cout / groups;
  if (hw % bm != 0 || cg < 8 || cg > bn || 2 * (hw / bm) > 128) return 0;
  return 2 * (hw / bm);
}

static int conv3x3_nhwc_impl(const void* X, const void* Wp, void* Y, float* workspace, unsigned* tickets, const void* bias,
                             const void* residual, int B, int H, int W, int Cin, int Cout, int stride, int bm, int bn,
                             int splits, int dtype, ga_stream_t stream, float* gn_partials, const void* gn_chan_bias,
                             int gn_groups);

extern "C" int ga_conv3x3_nhwc(const void* X, const void* Wp, void* Y, float* workspace, unsigned* tickets, const void* bias,
                               const void* residual, int B, int H, int W, int Cin, int Cout, int stride, int bm, int bn,
                               int splits, int dtype, ga_stream_t stream) {
  return conv3x3_nhwc_impl(X, Wp, Y, workspace, tickets, bias, residual, B, H, W, Cin, Cout, stride, bm, bn, splits, dtype, stream,
                           nullptr, nullptr, 0);
}

extern "C" int ga_conv3x3_nhwc_gn(const void* X, const void* Wp, void* Y, float* workspace, unsigned* tickets, const void* bias,
                                  const void* residual, int B, int H, int W, int Cin, int Cout, int bm, int bn, int splits,
                                  int dtype, ga_stream_t stream, float* gn_partials, const void* gn_chan_bias, int gn_groups) {
  if (!gn_partials) return GA_ERR_NULL;
  if (ga_conv3x3_gn_blocks(H, W, Cout, gn_groups, bm, bn) == 0) return GA_ERR_UNSUPPORTED;
  if (gn_chan_bias && !al16(gn_chan_bias)) return GA_ERR_ALIGN;
  return conv3x3_nhwc_impl(X, Wp, Y, workspace, tickets, bias, residual, B, H, W, Cin, Cout, 1, bm, bn, splits, dtype, stream,
                           gn_partials, gn_chan_bias, gn_groups);
}

static int conv3x3_nhwc_impl(const void* X, const void* Wp, void* Y, float* workspace, unsigned* tickets, const void* bias,
                             const void* residual, int B, int H, int W, int Cin, int Cout, int stride, int bm, int bn,
                             int splits, int dtype, ga_stream_t stream, float* gn_partials, const void* gn_chan_bias,
                             int gn_groups) {
  if (!X || !Wp || !Y) return GA_ERR_NULL;
  if (B < 1 || H < 1 || W < 1 || Cin < kKC || Cin % kKC != 0 || Cout < 8 || Cout % 8 != 0 || (stride != 1 && stride != 2))
    return GA_ERR_SHAPE;
  if (splits < 1 || splits > 64 || (splits > 1 && (!workspace || !tickets))) return GA_ERR_SHAPE;
  if (!al16(X) || !al16(Wp) || !al16(Y) || (bias && !al16(bias)) || (residual && !al16(residual))) return GA_ERR_ALIGN;
  ConvArgs a;
  a.gn_partials = gn_partials; a.gn_cbias = gn_chan_bias;
  a.gn_G = gn_groups; a.gn_cg = gn_groups ? Cout / gn_groups : 0; a.gn_tpi = gn_groups ? H * W / bm : 0;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.stride = stride;
  a.Ho = (H - 1) / stride + 1;
  a.Wo = (W - 1) / stride + 1;
  a.M = B * a.Ho * a.Wo;
  a.pad = 1;
  const long long xb = (long long)B * H * W * Cin * 2, wb = ga_conv3x3_packed_elems(Cout, Cin) * 2;
  if (xb >= (1LL << 31) || wb >= (1LL << 31) || (long long)a.M * Cout >= (1LL << 31)) return GA_ERR_SHAPE;   // 32-bit byte offsets
  a.x_bytes = (unsigned)xb;
  a.w_bytes = (unsigned)wb;
  a.w_tap = (int)(wb / 18); a.w_blk = 64 * Cin; a.w_row = 64; a.w_chunk = 64 * kKC;   // the blocked pack
  a.up = 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case GA_F16: return conv_t<_Float16>(X, Wp, Y, workspace, tickets, bias, residual, a, bm, bn, splits, s);
    case GA_BF16: return conv_t<bf16_t>(X, Wp, Y, workspace, tickets, bias, residual, a, bm, bn, splits, s);
    default: return GA_ERR_DTYPE;
  }
}

extern "C" int ga_conv3x3_up2x_nhwc(const void* X, const void* Wp, void* Y, float* workspace, unsigned* tickets,
                                    const void* bias, const void* residual, int B, int H, int W, int Cin, int Cout, int bm,
                                    int bn, int splits, int dtype, ga_stream_t stream) {
  if (!X || !Wp || !Y) return GA_ERR_NULL;
  if (B < 1 || H < 1 || W < 1 || Cin < kKC || Cin % kKC != 0 || Cout < 8 || Cout % 8 != 0) return GA_ERR_SHAPE;
  if (splits < 1 || splits > 64 || (splits > 1 && (!workspace || !tickets))) return GA_ERR_SHAPE;
  if (!al16(X) || !al16(Wp) || !al16(Y) || (bias && !al16(bias)) || (residual && !al16(residual))) return GA_ERR_ALIGN;
  ConvArgs a;
  a.gn_partials = nullptr; a.gn_cbias = nullptr; a.gn_cg = a.gn_G = a.gn_tpi = 0;
  a.B = B; a.H = 2 * H; a.W = 2 * W; a.Cin = Cin; a.Cout = Cout; a.stride = 1;
  a.Ho = a.H; a.Wo = a.W;
  a.M = B * a.Ho * a.Wo;
  a.pad = 1;
  const long long xb = (long long)B * H * W * Cin * 2, wb = ga_conv3x3_packed_elems(Cout, Cin) * 2;
  if (xb >= (1LL << 31) || wb >= (1LL << 31) || (long long)a.M * Cout >= (1LL << 31)) return GA_ERR_SHAPE;
  a.x_bytes = (unsigned)xb;
  a.w_bytes = (unsigned)wb;
  a.w_tap = (int)(wb / 18); a.w_blk = 64 * Cin; a.w_row = 64; a.w_chunk = 64 * kKC;
  a.up = 1;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case GA_F16: return conv_t<_Float16>(X, Wp, Y, workspace, tickets, bias, residual, a, bm, bn, splits, s);
    case GA_BF16: return conv_t<bf16_t>(X, Wp, Y, workspace, tickets, bias, residual, a, bm, bn, splits, s);
    default: return GA_ERR_DTYPE;
  }
}

/* Y[M][N] = X[M][K] * W[N][K]^T (+ bias[N]) (+ residual[M][N]): the Linear layers and 1x1 convolutions of the UNet (weights
 * in the framework's own [out][in] layout, no packing) on the same pipelined MFMA kernel, as a one-tap convolution. */
extern "C" int ga_gemm_nt(const void* X, const void* W, void* Y, float* workspace, unsigned* tickets, const void* bias,
                          const void* residual, int64_t M, int K, int N, int bm, int bn, int splits, int dtype,
                          ga_stream_t stream) {
  if (!X || !W || !Y) return GA_ERR_NULL;
  if (M < 1 || K < kKC || K % kKC != 0 || N < 8 || N % 8 != 0) return GA_ERR_SHAPE;
  if (splits < 1 || splits > 64 || (splits > 1 && (!workspace || !tickets))) return GA_ERR_SHAPE;
  if (!al16(X) || !al16(W) || !al16(Y) || (bias && !al16(bias)) || (residual && !al16(residual))) return GA_ERR_ALIGN;
  const long long xb = (long long)M * K * 2, wb = (long long)N * K * 2;
  if (xb >= (1LL << 31) || wb >= (1LL << 31) || (long long)M * N >= (1LL << 31)) return GA_ERR_SHAPE;
  ConvArgs a;
  a.gn_partials = nullptr; a.gn_cbias = nullptr; a.gn_cg = a.gn_G = a.gn_tpi = 0;
  a.B = 1; a.H = (int)M; a.W = 1; a.Cin = K; a.Cout = N; a.stride = 1;
  a.Ho = (int)M; a.Wo = 1; a.M = (int)M;
  a.pad = 0;
  a.x_bytes = (unsigned)xb;
  a.w_bytes = (unsigned)wb;
  a.w_tap = 0; a.w_blk = 64 * K; a.w_row = K; a.w_chunk = kKC;   // row-major [N][K]
  a.up = 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case GA_F16: return conv_t<_Float16>(X, W, Y, workspace, tickets, bias, residual, a, bm, bn, splits, s);
    case GA_BF16: return conv_t<bf16_t>(X, W, Y, workspace, tickets, bias, residual, a, bm, bn, splits, s);
    default: return GA_ERR_DTYPE;
  }
}

#if defined(GA_CONV_STAMPS)
extern "C" int ga_conv_set_stamps(void* buffer) {   // [workgroups of the next launches][12] u64, or NULL
  unsigned long long* p = static_cast<unsigned long long*>(buffer);
  return hipMemcpyToSymbol(HIP_SYMBOL(g_conv_stamps), &p, sizeof(p)) == hipSuccess ? GA_OK : GA_ERR_LAUNCH;
}
#endif
