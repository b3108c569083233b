// Linear layers of the UNet's transformer blocks (to_q / to_k / to_v / to_out, proj_in / proj_out, the GEGLU feed-forward:
// diffusers 0.12.1 CrossAttention / BasicTransformerBlock called from utils/ptp_utils.py:70-91 and
// pipeline_guided_attention.py:647-738) as ONE kernel family with the element-wise neighbours folded in:
//
//   Y[m][n] = epilogue( sum_k X[m][k] * W[n][k] )            X [M][K] (row stride ldx), W [N][K] (the framework's layout)
//
//   LayerNorm in front   LN(x) W^T = rstd[m] * (x W'^T - mean[m] * colsum[n]) + shift[n],  W' = gamma o W,
//                        colsum[n] = sum_k W'[n][k], shift[n] = sum_k beta[k] W[n][k] + bias[n]  (host, once per weight):
//                        the GEMM runs on the RAW residual stream and the normalisation is two FMAs per output element.
//                        mean / rstd come from per-(row, column-tile) partial sums that the PRODUCING call's epilogue
//                        left behind (row_partials_out) — the row never makes a pass of its own.
//   bias + residual      the attention / feed-forward output projections land on the residual stream directly.
//   GEGLU                W = [h rows | gate rows]; a workgroup takes BN/2 h columns and the matching gate columns and
//                        writes h * gelu(gate): the [M][2F] projection never exists (no-grad passes) or is written
//                        beside the result for the backward (grad passes).
//
// Why an own kernel: at guidance batch 1 these are 160 launches per UNet pass of 5 - 40 k-steps each (M = 4096 ... 64
// tokens) — prologue, epilogue and memory latency, not MFMA time.  Structure:
//   * operands go global -> LDS directly (buffer_load ... lds, 16 bytes per lane, no staging registers, no ds_write),
//     into a ring of NSTAGE k-steps of 64; NSTAGE - 1 steps stay in flight behind counted s_waitcnt vmcnt(N) and raw
//     s_barrier (one barrier per k-step);  __syncthreads() would drain the ring (it waits vmcnt(0) with an LDS-DMA pending);
//   * an LDS-DMA wave-instruction writes 1 KiB contiguously (8 rows of 128 bytes), so rows cannot be padded: the image is
//     XOR-swizzled in 16-byte units, physical column = logical column ^ ((row >> 1) & 7), applied on the SOURCE address
//     and on the fragment reads (the ds_read_b128 lane groups of a 32-row fragment then cover all 64 banks);
//   * 256 threads = 2 x 2 waves, v_mfma_f32_32x32x16; the weight fragment is the A operand, the token fragment the B
//     operand: a lane ends with 4 consecutive output features of one token per register quad, and the tile leaves through
//     LDS as whole 16-byte row pieces;
//   * split-K for the small-M shapes with the reduction INSIDE the launch: every slice stores its f32 accumulators
//     write-through (sc1) in its own thread order, drains, and one lane takes a ticket; the slice whose ticket is last
//     re-reads the others with sc1 loads IN SLICE ORDER (bitwise reproducible), runs the epilogue and returns the ticket
//     word to zero.  No second launch, no atomics on data.
#include "attn_common.h"

using namespace ga;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kBK = 64;          // depth of one k-step (elements): 128-byte LDS rows
constexpr int kThreads = 256;
#define GA_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <typename T>
struct Mma32L;
template <>
struct Mma32L<_Float16> {
  __device__ static __forceinline__ f32x16 run(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};
template <>
struct Mma32L<bf16_t> {
  __device__ static __forceinline__ f32x16 run(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};

struct LinArgs {
  int M, N, K;               // N = rows of W (GEGLU: 2F)
  int ldx, ldy, ld_res, ld_pre;
  int tm, tn, splits, steps, steps_per, n_fastest;
  FastDiv d_tm, d_tn;        // tile mapping by multiply-high (lin_tile)
  int F;                     // GEGLU: features of the result (N / 2); 0 otherwise
  int ln_parts;              // LayerNorm fold: partial sums per row; 0 = no fold
  float ln_eps, ln_inv_k;
  unsigned x_bytes, w_bytes;
};

// The scalar head of the argument list (see linear_kernel).  Passed as SEPARATE kernel parameters — only leading scalar /
// pointer parameters are preloaded, an aggregate ends the sequence — and re-assembled by the launch stub below.
struct LinHead {
  int M, ldx;
  unsigned kn;          // K | N << 16
  int tm;
  unsigned tn_splits;   // tn | splits << 16
  unsigned steps;       // steps | steps_per << 16 | n_fastest << 31
  unsigned dtm_m, dtn_m;
};

struct LinPtrs {
  const void* bias;          // [N] T or null
  const void* residual;      // [M][ld_res] T or null
  void* preact;              // GEGLU: optional [M][ld_pre] T, the projection before the gate
  const float* ln_partials;  // [M][ln_parts][2]
  const float* ln_colsum;    // [N]
  const float* ln_shift;     // [N]
  float* ln_stats_out;       // optional [M][2] (mean, rstd)
  float* row_partials_out;   // optional [M][tn][2]
  float* slabs;              // split-K: [splits][tm * tn][BM * BN] f32, thread order
  unsigned* tickets;         // split-K: [tm * tn], zero on entry and on exit
  float* gn_partials;        // optional: GroupNorm partial sums of the stored result for the consuming norm (see lin_epilogue)
  int gn_cg, gn_G, gn_hw, gn_tpi;   // channels per group, groups, rows (pixels) per image, m tiles per image
};

// workgroup -> (m tile, n tile, k slice): an XCD (workgroup ids equal mod 8) owns a contiguous run of the logical order, so
// the tiles that share a weight slice (or a token slab) meet in one L2.  Bijective for any grid size.
template <int BM, int BN>
__device__ __forceinline__ void lin_tile(const LinArgs& a, int& mt, int& nt, int& split) {
  const int total = a.tm * a.tn * a.splits, q = total >> 3, r = total & 7;   // == gridDim.x, without the hidden-argument load
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  // divisions by multiply-high with the host's reciprocals (two runtime integer divisions were ~100 instructions in front of
  // every workgroup's first load); selects instead of two branches assigning mt / nt in turn (see conv3x3.hip)
  const bool nf = a.n_fastest != 0;
  const FastDiv d1 = nf ? a.d_tn : a.d_tm, d2 = nf ? a.d_tm : a.d_tn;
  const int t1 = nf ? a.tn : a.tm, t2 = nf ? a.tm : a.tn;
  const int rest = fdiv(logical, d1), i1 = logical - rest * t1;
  split = fdiv(rest, d2);
  const int i2 = rest - split * t2;
  mt = nf ? i2 : i1;
  nt = nf ? i1 : i2;
}

// Inline asm with GPU register constraints only exists in the device pass: the host pass parses kernel bodies too, and a
// constraint it does not know silently voids the kernel's definition there (the launch stub then stays undefined).
__device__ __forceinline__ void lds_read128(u32x4& dst, unsigned byte_address) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(byte_address) : "memory");
#else
  dst = u32x4{byte_address, 0u, 0u, 0u};
#endif
}
// the value passes through an empty volatile asm: what is computed from it cannot be scheduled above this point
__device__ __forceinline__ void opaque(u32x4& v) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(v));
#endif
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}
template <int N>
__device__ __forceinline__ void wait_lgkmcnt() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
#endif
}

// ---- diagnostic build only (make stamps -> tools/micro/libga_stamps.so, tools/micro/lin_stamps.py): wave 0 of every workgroup
// reads the shader clock at the phase boundaries of linear_kernel and leaves the values in a buffer of their own.  No stamp
// executes in the product library (the macro is never defined there).
#if defined(GA_LIN_STAMPS)
__device__ unsigned long long* g_lin_stamps = nullptr;   // [workgroup][10]
struct LinStamps {
  unsigned long long t[10];
  __device__ __forceinline__ void at(int i) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long v;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    t[i] = v;
#endif
  }
  __device__ __forceinline__ void real(int i) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long v;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    t[i] = v;
#endif
  }
  __device__ __forceinline__ void flush() {
    if (threadIdx.x == 0 && g_lin_stamps != nullptr)
      for (int i = 0; i < 10; ++i) g_lin_stamps[(size_t)blockIdx.x * 10 + i] = t[i];
  }
};
#define GA_STAMP(st, i) (st).at(i)
#else
struct LinStamps {};
#define GA_STAMP(st, i) ((void)0)
#endif

// gelu(g) = g/2 (1 + erf(g / sqrt 2)); erff() made the epilogue longer than the tile's matrix work (the 12288 x 320 x 2560
// call: 58 us with erff).
typedef float f32x2 __attribute__((ext_vector_type(2)));
// Two gelu at a time on the packed-f32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32: two lanes' worth per issue):
// erf(x) = x P(x^2) / Q(x^2) on [-4, 4] (the rational approximation XLA / Eigen use for f32: |err| <= 4.5e-7, constant beyond
// |x| = 4 where 1 - |erf| < 2e-8), one v_rcp per element and no v_exp.  The first form here — Abramowitz-Stegun 7.1.26: one v_rcp,
// one v_exp and scalar FMAs per element — cost about 84 issue cycles per element and wave against about 45 for this one; with
// five k-steps of MFMAs per tile the GEGLU epilogue (32 gelu per thread and tile) is as long as the tile's matrix work.
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 g) {
  const f32x2 lim = {4.0f, 4.0f};
  f32x2 x = g * 0.70710678118654752f;
  x = __builtin_elementwise_min(__builtin_elementwise_max(x, -lim), lim);
  const f32x2 x2 = x * x;
  auto c = [](float v) { return f32x2{v, v}; };
  f32x2 p = c(-2.72614225801306e-10f);
  p = __builtin_elementwise_fma(p, x2, c(2.77068142495902e-08f));
  p = __builtin_elementwise_fma(p, x2, c(-2.10102402082508e-06f));
  p = __builtin_elementwise_fma(p, x2, c(-5.69250639462346e-05f));
  p = __builtin_elementwise_fma(p, x2, c(-7.34990630326855e-04f));
  p = __builtin_elementwise_fma(p, x2, c(-2.95459980854025e-03f));
  p = __builtin_elementwise_fma(p, x2, c(-1.60960333262415e-02f));
  f32x2 q = c(-1.45660718464996e-05f);
  q = __builtin_elementwise_fma(q, x2, c(-2.13374055278905e-04f));
  q = __builtin_elementwise_fma(q, x2, c(-1.68282697438203e-03f));
  q = __builtin_elementwise_fma(q, x2, c(-7.37332916720468e-03f));
  q = __builtin_elementwise_fma(q, x2, c(-1.42647390514189e-02f));
  const f32x2 rq = {__builtin_amdgcn_rcpf(q.x), __builtin_amdgcn_rcpf(q.y)};   // v_rcp_f32 (1 ulp); __frcp_rn is a ten-instruction IEEE division
  const f32x2 erf = x * p * rq;
  const f32x2 hg = g * 0.5f;
  return __builtin_elementwise_fma(hg, erf, hg);
}

// What the epilogue reads from memory besides the accumulators — bias or the LayerNorm fold's (shift, colsum) per output
// column, the residual rows — is requested BEHIND THE FIRST RING LOADS and sits in registers when the main loop ends.  Loaded
// where it is used (the first form of this kernel) the epilogue was a chain of dependent round trips at the end of every
// launch: four bias loads, each behind its own `if (column < N)` and waited for on its own, then one residual load per
// output vector — 3-5 us of pure latency in launches of 6-10 us.  Addresses are clamped instead of predicated (columns /
// rows past the edge are computed and never stored).
template <typename T, int BM, int BN, bool GEGLU, bool LN>
struct LinPrefetch {
  static constexpr int JN = BN / 64;
  static constexpr int OUTC = GEGLU ? BN / 2 : BN;
  static constexpr int VPR = OUTC / 8;                       // 16-byte vectors per output row
  static constexpr int NV = BM * VPR / kThreads;             // output vectors per thread
  static_assert(BM * VPR % kThreads == 0, "every thread stores the same number of vectors");
  typename Traits<T>::frag bias[LN ? 1 : JN][LN ? 1 : 4];
  f32x4 shift[LN ? JN : 1][LN ? 4 : 1], colsum[LN ? JN : 1][LN ? 4 : 1];
  u32x4 res[GEGLU ? 1 : NV];
  bool has_bias, has_res;

  __device__ __forceinline__ void load(const LinArgs& a, const LinPtrs& p, int m0, int n0) {
    constexpr int WN = BN / 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave & 1, fh = lane >> 5;
    const T* bp = static_cast<const T*>(p.bias);
    has_bias = !LN && bp != nullptr;
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int c = wn * WN + j * 32 + 8 * qd + 4 * fh;
        int n;                                               // row of W this column came from, clamped into the matrix
        if (GEGLU) n = c < BN / 2 ? min(n0 + c, a.F - 4) : a.F + min(n0 + c - BN / 2, a.F - 4);
        else n = min(n0 + c, a.N - 4);
        if constexpr (LN) {
          shift[j][qd] = *reinterpret_cast<const f32x4*>(p.ln_shift + n);
          colsum[j][qd] = *reinterpret_cast<const f32x4*>(p.ln_colsum + n);
        } else {
          if (has_bias) bias[j][qd] = load_frag<T>(bp + n);
        }
      }
    const T* rp = static_cast<const T*>(p.residual);
    has_res = !GEGLU && rp != nullptr;
    if constexpr (!GEGLU) {
      if (has_res) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
          const int v = tid + k * kThreads, r = v / VPR, cv = (v - r * VPR) * 8;
          const int m = min(m0 + r, a.M - 1), n = min(n0 + cv, a.N - 8);
          res[k] = *reinterpret_cast<const u32x4*>(rp + (size_t)m * a.ld_res + n);
        }
      }
    }
  }
};

// Epilogue of both kernels: bias / LayerNorm algebra in registers, the tile through LDS (`Cs`, BM x (BN + 8) elements that no
// pending LDS-DMA targets), out as whole 16-byte row pieces with GEGLU / the residual applied and the row partial sums taken.
// acc[j][i][r]: tile column c = wn * WN + j * 32 + 8 * (r >> 2) + 4 * fh + (r & 3), tile row wm * WM + i * 32 + fr.
template <typename T, int BM, int BN, bool GEGLU, bool LN>
__device__ __forceinline__ void lin_epilogue(f32x16 (&acc)[BN / 64][BM / 64], T* Cs, T* __restrict__ Y, const LinArgs& a,
                                             const LinPtrs& p, int m0, int n0, int nt, const float (&ln_mean)[BM / 64],
                                             const float (&ln_rstd)[BM / 64], const LinPrefetch<T, BM, BN, GEGLU, LN>& pre,
                                             LinStamps& st) {
  constexpr int WM = BM / 2, WN = BN / 2, IM = WM / 32, JN = WN / 32;
  constexpr int LDC = BN + 8;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, fr = lane & 31, fh = lane >> 5;
#pragma unroll
  for (int i = 0; i < IM; ++i)
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int c = wn * WN + j * 32 + 8 * qd + 4 * fh;
        float add[4] = {0.f, 0.f, 0.f, 0.f}, cs[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (LN) {
#pragma unroll
          for (int r = 0; r < 4; ++r) add[r] = pre.shift[j][qd][r], cs[r] = pre.colsum[j][qd][r];
        } else {
          if (pre.has_bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r) add[r] = Traits<T>::to_f32(pre.bias[j][qd][r]);
          }
        }
        typename Traits<T>::frag f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[j][i][4 * qd + r];
          if constexpr (LN) v = ln_rstd[i] * (v - ln_mean[i] * cs[r]) + add[r];
          else v += add[r];
          f[r] = Traits<T>::from_f32(v);
        }
        store_frag<T>(Cs + (wm * WM + i * 32 + fr) * LDC + c, f);
      }
  if constexpr (LN) {
    if (p.ln_stats_out != nullptr && nt == 0 && wn == 0 && fh == 0) {
#pragma unroll
      for (int i = 0; i < IM; ++i) {
        const int m = m0 + wm * WM + i * 32 + fr;
        if (m < a.M) *reinterpret_cast<float2*>(p.ln_stats_out + (size_t)m * 2) = float2{ln_mean[i], ln_rstd[i]};
      }
    }
  }
  __syncthreads();
  GA_STAMP(st, 5);
  constexpr int OUTC = GEGLU ? BN / 2 : BN;       // columns of Y this tile writes
  constexpr int VPR = OUTC / 8;                   // 16-byte vectors per output row
  constexpr int NV = BM * VPR / kThreads;
  static_assert((VPR & (VPR - 1)) == 0 && VPR <= 16, "the row reduction below adds inside a DPP row");
  const int n_out = GEGLU ? a.F : a.N;
  // GroupNorm statistics for the norm that consumes Y (proj_out + residual in front of a ResnetBlock's norm1): per-thread sums of
  // the values AS STORED over this thread's rows — its NV vectors share one column run (kThreads % VPR == 0) — folded through
  // LDS in a fixed order below, exactly as conv_epilogue does (conv3x3.hip)
  const bool gn = !GEGLU && p.gn_partials != nullptr;
  float gsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, gsq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int v = tid + k * kThreads;
    const int r = v / VPR, cv = (v - r * VPR) * 8;
    const int m = m0 + r, n = n0 + cv;
    const bool ok = m < a.M && n < n_out;
    uint4 val = *reinterpret_cast<const uint4*>(Cs + r * LDC + cv);
    T* e = reinterpret_cast<T*>(&val);
    if constexpr (GEGLU) {
      if (ok && p.preact != nullptr) {
        T* pre_row = static_cast<T*>(p.preact) + (size_t)m * a.ld_pre;
        *reinterpret_cast<uint4*>(pre_row + n) = val;
        *reinterpret_cast<uint4*>(pre_row + a.F + n) = *reinterpret_cast<const uint4*>(Cs + r * LDC + BN / 2 + cv);
      }
      const uint4 gv = *reinterpret_cast<const uint4*>(Cs + r * LDC + BN / 2 + cv);
      const T* ge = reinterpret_cast<const T*>(&gv);
#pragma unroll
      for (int q = 0; q < 8; q += 2) {
        const f32x2 gl = gelu_erf2(f32x2{Traits<T>::to_f32(ge[q]), Traits<T>::to_f32(ge[q + 1])});
        e[q] = Traits<T>::from_f32(Traits<T>::to_f32(e[q]) * gl.x);
        e[q + 1] = Traits<T>::from_f32(Traits<T>::to_f32(e[q + 1]) * gl.y);
      }
    } else {
      if (pre.has_res) {
        u32x4 rv = pre.res[k];
        opaque(rv);   // the unpacking stays here: hoisted to the loads it made the prologue wait for them
        const T* re = reinterpret_cast<const T*>(&rv);
#pragma unroll
        for (int q = 0; q < 8; ++q) e[q] = Traits<T>::from_f32(Traits<T>::to_f32(e[q]) + Traits<T>::to_f32(re[q]));
      }
    }
    if (ok) *reinterpret_cast<uint4*>(Y + (size_t)m * a.ldy + n) = val;
    if constexpr (!GEGLU) {
      if (gn && ok) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float x = Traits<T>::to_f32(e[q]);
          gsum[q] += x;
          gsq[q] += x * x;
        }
      }
    }
    if (p.row_partials_out != nullptr) {
      // (sum, sum of squares) of the values AS STORED over this tile's columns of the row: VPR adjacent lanes hold one row
      float s1 = 0.f, s2 = 0.f;
      if (ok) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float x = Traits<T>::to_f32(e[q]);
          s1 += x;
          s2 += x * x;
        }
      }
      s1 = group_sum<VPR>(s1);   // DPP adds (ga_common.h); the shuffles this replaces were 6 dependent ds_bpermute round trips per vector
      s2 = group_sum<VPR>(s2);
      if ((v & (VPR - 1)) == 0 && m < a.M)
        *reinterpret_cast<float2*>(p.row_partials_out + ((size_t)m * a.tn + nt) * 2) = float2{s1, s2};
    }
  }
  if constexpr (!GEGLU) {
    if (gn) {
      // this thread's 8 channels lie in at most two groups (8 <= channels per group): gA takes the first `split` of them
      const int cv = (tid & (VPR - 1)) * 8;
      const int c_abs = n0 + cv, gA = c_abs / p.gn_cg, split = min(8, (gA + 1) * p.gn_cg - c_abs);
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
      float4* lds4 = reinterpret_cast<float4*>(Cs);
      lds4[tid] = r;
      __syncthreads();
      const int n_end = min(n0 + BN, a.N);
      const int g_first = n0 / p.gn_cg, g_last = (n_end - 1) / p.gn_cg;
      const int g = g_first + tid;
      if (g <= g_last) {
        const int c_lo = max(g * p.gn_cg, n0), c_hi = min((g + 1) * p.gn_cg, n_end);      // this tile's channels of group g
        float sa = 0.f, sq = 0.f;
        for (int vv = (c_lo - n0) >> 3; vv <= (c_hi - 1 - n0) >> 3; ++vv) {
          const bool first = (n0 + 8 * vv) / p.gn_cg == g;   // g is this vector's gA, otherwise its gA + 1
          for (int pr = 0; pr < kThreads / VPR; ++pr) {       // fixed order: the same bits whoever runs first
            const float4 e4 = lds4[pr * VPR + vv];
            sa += first ? e4.x : e4.z;
            sq += first ? e4.y : e4.w;
          }
        }
        // slot 2 t of m tile t: the n tile a group STARTS in; slot 2 t + 1: the n tile it continues into (zero when it does not)
        const int b = m0 / p.gn_hw, t = (m0 - b * p.gn_hw) / BM;
        float2* out = reinterpret_cast<float2*>(p.gn_partials) + ((size_t)b * 2 * p.gn_tpi + 2 * t) * p.gn_G + g;
        const bool starts = g * p.gn_cg >= n0, ends = (g + 1) * p.gn_cg <= n_end;
        if (starts) {
          out[0] = float2{sa, sq};
          if (ends) out[p.gn_G] = float2{0.f, 0.f};
        } else {
          out[p.gn_G] = float2{sa, sq};
        }
      }
    }
  }
}

// GEGLU: BN columns of the tile = BN/2 h features followed by the BN/2 gate features of the same output columns
template <typename T, int BM, int BN, int NSTAGE, bool GEGLU, bool LN>
__global__ __launch_bounds__(kThreads, (BM + BN) * 128 * NSTAGE <= 80 * 1024 ? 2 : 1) void linear_kernel(
    const T* __restrict__ X, const T* __restrict__ W, T* __restrict__ Y, int h_M, int h_ldx, unsigned h_kn, int h_tm,
    unsigned h_tn_splits, unsigned h_steps, unsigned h_dtm_m, unsigned h_dtn_m, LinArgs a_in, LinPtrs p) {
  const LinHead h{h_M, h_ldx, h_kn, h_tm, h_tn_splits, h_steps, h_dtm_m, h_dtn_m};
  LinStamps st;
#if defined(GA_LIN_STAMPS)
  st.real(8);
#endif
  GA_STAMP(st, 0);
  // Everything in front of the first operand load comes from `X, W, Y, h`: 14 dwords of scalar arguments that the hardware
  // PRELOADS into SGPRs at wave launch (-amdgpu-kernarg-preload-count, Makefile) — the kernel's first instructions used to
  // be three dependent scalar loads of the argument block (one cache miss, two hits) in front of the tile mapping.  The
  // rest of the arguments (strides of the outputs, LayerNorm constants, pointers of the epilogue) is loaded while the
  // ring fills.
  LinArgs a;
  a.M = h.M; a.ldx = h.ldx; a.K = (int)(h.kn & 0xffffu); a.N = (int)(h.kn >> 16);
  a.tm = h.tm; a.tn = (int)(h.tn_splits & 0xffffu); a.splits = (int)(h.tn_splits >> 16);
  a.steps = (int)(h.steps & 0xffffu); a.steps_per = (int)((h.steps >> 16) & 0x7fffu); a.n_fastest = (int)(h.steps >> 31);
  a.d_tm = FastDiv{(unsigned)a.tm, h.dtm_m}; a.d_tn = FastDiv{(unsigned)a.tn, h.dtn_m};
  a.F = GEGLU ? a.N / 2 : 0;
  a.x_bytes = (unsigned)(((a.M - 1) * a.ldx + a.K) * 2); a.w_bytes = (unsigned)(a.N * a.K * 2);
  a.ldy = a_in.ldy; a.ld_res = a_in.ld_res; a.ld_pre = a_in.ld_pre;
  a.ln_parts = a_in.ln_parts; a.ln_eps = a_in.ln_eps; a.ln_inv_k = a_in.ln_inv_k;
  constexpr int WM = BM / 2, WN = BN / 2, IM = WM / 32, JN = WN / 32;
  constexpr int kStage = (BM + BN) * kBK;                  // elements per ring slot
  constexpr int IPS = (BM + BN) / 32;                      // LDS-DMA wave-instructions per slot and wave (8 rows each)
  constexpr int LDC = BN + 8;                              // output staging row stride
  constexpr int kLds = NSTAGE * kStage > BM * LDC ? NSTAGE * kStage : BM * LDC;
  __shared__ __attribute__((aligned(1024))) T lds[kLds];   // the ONE LDS object of the kernel (ring, flags, output tile)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, fr = lane & 31, fh = lane >> 5;
  int mt, nt, split;
  lin_tile<BM, BN>(a, mt, nt, split);
  const int m0 = mt * BM;
  const int n0 = GEGLU ? nt * (BN / 2) : nt * BN;          // first output column of the tile
  const int it0 = split * a.steps_per, it1 = min(a.steps, it0 + a.steps_per);
  const int nsteps = it1 - it0;

  // ---- LDS-DMA source addresses: instruction q of this wave fills rows 8 g .. 8 g + 7 of the slot, g = wave + 4 q; lane l
  // lands at row 8 g + (l >> 3), physical 16-byte column l & 7, and therefore fetches logical column (l & 7) ^ swizzle(row)
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(X), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(W), 0, a.w_bytes, 0x00020000);
  unsigned voff[IPS];
#pragma unroll
  for (int q = 0; q < IPS; ++q) {
    const int row = 8 * (wave + 4 * q) + (lane >> 3);
    const int col = (lane & 7) ^ ((row >> 1) & 7);
    if (row < BM) {
      const int m = min(m0 + row, a.M - 1);                // rows past M repeat the last one (never stored)
      voff[q] = (unsigned)((m * a.ldx + 8 * col) * (int)sizeof(T));
    } else {
      const int r = row - BM;
      int n;
      if (GEGLU) n = r < BN / 2 ? min(n0 + r, a.F - 1) : a.F + min(n0 + r - BN / 2, a.F - 1);
      else n = min(n0 + r, a.N - 1);
      voff[q] = (unsigned)((n * a.K + 8 * col) * (int)sizeof(T));
    }
  }
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto issue = [&](int step, int slot) {                   // k-step `step` of this slice into ring slot `slot`
    const unsigned koff = (unsigned)((it0 + step) * kBK * (int)sizeof(T));
#pragma unroll
    for (int q = 0; q < IPS; ++q) {
      T* dst = lds + slot * kStage + (wave_u + 4 * q) * 8 * kBK;
      if (8 * (wave_u + 4 * q) < BM)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, GA_LDS_PTR(dst), 16, (int)voff[q], (int)koff, 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, GA_LDS_PTR(dst), 16, (int)voff[q], (int)koff, 0, 0);
    }
  };

  // ---- fragment addresses (LDS byte addresses inside slot 0): row * 128 + 16 * ((2 kk + fh) ^ swizzle(row)).  The reads
  // are inline asm: a C++ LDS load makes the compiler wait vmcnt(0) first (any pending LDS-DMA may alias it as far as its
  // wait-count pass can tell, and the counted waits below are invisible to it) — the ring would drain at every k-step.
  unsigned a_adr[IM], a_sw[IM], b_adr[JN], b_sw[JN];
  const unsigned lds0 = (unsigned)(uintptr_t)GA_LDS_PTR(lds);
#pragma unroll
  for (int i = 0; i < IM; ++i) {
    const int row = wm * WM + i * 32 + fr;
    a_adr[i] = lds0 + row * (kBK * (int)sizeof(T));
    a_sw[i] = (unsigned)(((row >> 1) & 7) ^ fh);
  }
#pragma unroll
  for (int j = 0; j < JN; ++j) {
    const int row = BM + wn * WN + j * 32 + fr;
    b_adr[j] = lds0 + row * (kBK * (int)sizeof(T));
    b_sw[j] = (unsigned)(((row >> 1) & 7) ^ fh);
  }

  f32x16 acc[JN][IM];
#pragma unroll
  for (int j = 0; j < JN; ++j)
#pragma unroll
    for (int i = 0; i < IM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

  // ---- main loop: ring of NSTAGE slots, NSTAGE - 1 k-steps in flight
  constexpr int PRE = NSTAGE - 1;
#pragma unroll
  for (int s = 0; s < PRE; ++s)
    if (s < nsteps) issue(s, s);
  LinPrefetch<T, BM, BN, GEGLU, LN> pre;
  pre.load(a, p, m0, n0);
  // ---- LayerNorm fold: this lane's rows' statistics from the producer's partial sums.  Requested behind the first ring loads
  // (in front of them they were a dependent round trip before the first operand load), four parts per row in flight (one
  // load, one wait per part was up to 20 serial round trips at 1280 channels); summed in part order as before.
  float ln_mean[IM], ln_rstd[IM];
  if constexpr (LN) {
    const float2* part[IM];
    float s1[IM], s2[IM];
#pragma unroll
    for (int i = 0; i < IM; ++i) {
      const int m = min(m0 + wm * WM + i * 32 + fr, a.M - 1);
      part[i] = reinterpret_cast<const float2*>(p.ln_partials) + (size_t)m * a.ln_parts;
      s1[i] = s2[i] = 0.f;
    }
    for (int q0 = 0; q0 < a.ln_parts; q0 += 4) {
      float2 v[IM][4];
#pragma unroll
      for (int i = 0; i < IM; ++i)
#pragma unroll
        for (int u = 0; u < 4; ++u) v[i][u] = part[i][min(q0 + u, a.ln_parts - 1)];
#pragma unroll
      for (int i = 0; i < IM; ++i)
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (q0 + u < a.ln_parts) {
            s1[i] += v[i][u].x;
            s2[i] += v[i][u].y;
          }
    }
#pragma unroll
    for (int i = 0; i < IM; ++i) {
      const float mean = s1[i] * a.ln_inv_k;
      ln_mean[i] = mean;
      ln_rstd[i] = rsqrtf(fmaxf(s2[i] * a.ln_inv_k - mean * mean, 0.f) + a.ln_eps);
    }
  }

  GA_STAMP(st, 1);
  for (int it = 0; it < nsteps; ++it) {
    // my loads of step `it` have landed when at most the younger steps' instructions are outstanding
    const int younger = min(nsteps - 1 - it, PRE - 1);
    if (younger >= 3) wait_vmcnt<3 * IPS>();
    else if (younger == 2) wait_vmcnt<2 * IPS>();
    else if (younger == 1) wait_vmcnt<IPS>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();   // everyone's part of step `it` is in LDS; everyone is done reading step it - 1
#if defined(GA_LIN_STAMPS)
    if (it == 0) st.at(2);
#endif
    if (it + PRE < nsteps) issue(it + PRE, (it + PRE) % NSTAGE);   // refills the slot step it - 1 occupied
    const unsigned slot_off = (unsigned)((it % NSTAGE) * kStage * (int)sizeof(T));
    // fragments of sub-step kk + 1 are requested before the MFMAs of kk; LDS returns in order, so "all but the reads just
    // issued" (a counted lgkmcnt) means the operands of kk are in their registers
    u32x4 fa[2][IM], fb[2][JN];
    auto request = [&](int kk, int set) {
#pragma unroll
      for (int i = 0; i < IM; ++i)
        lds_read128(fa[set][i], a_adr[i] + slot_off + 16u * ((2u * kk) ^ a_sw[i]));
#pragma unroll
      for (int j = 0; j < JN; ++j)
        lds_read128(fb[set][j], b_adr[j] + slot_off + 16u * ((2u * kk) ^ b_sw[j]));
    };
    request(0, 0);
#pragma unroll
    for (int kk = 0; kk < kBK / 16; ++kk) {
      if (kk + 1 < kBK / 16) {
        request(kk + 1, (kk + 1) & 1);
        wait_lgkmcnt<IM + JN>();
      } else {
        wait_lgkmcnt<0>();
      }
      __builtin_amdgcn_sched_barrier(0);   // the MFMAs below must not be hoisted above the wait (register-only: "memory" does not order them)
#pragma unroll
      for (int j = 0; j < JN; ++j)
#pragma unroll
        for (int i = 0; i < IM; ++i)
          acc[j][i] = Mma32L<T>::run(__builtin_bit_cast(uint4, fb[kk & 1][j]), __builtin_bit_cast(uint4, fa[kk & 1][i]), acc[j][i]);
    }
  }
  static_assert(NSTAGE >= 2 && NSTAGE <= 5, "the counted waits above cover up to four k-steps in flight");
  wait_lgkmcnt<0>();
  __builtin_amdgcn_s_barrier();   // every wave's fragment reads are done: the ring memory is free for the epilogue
  GA_STAMP(st, 3);

  // ---- split-K: publish, take a ticket; only the last slice of a tile goes on
  if (a.splits > 1) {
    const int tile = nt * a.tm + mt;
    const __amdgpu_buffer_rsrc_t srsrc =
        __builtin_amdgcn_make_buffer_rsrc(p.slabs, 0, 0x7ffffff0, 0x00020000);
    constexpr int QUADS = JN * IM * 4;
    const unsigned per_slice = (unsigned)(a.tm * a.tn) * (BM * BN * 4u);
    const unsigned tbase = (unsigned)tile * (BM * BN * 4u) + (unsigned)tid * 16u;
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
      for (int i = 0; i < IM; ++i)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const int q = (j * IM + i) * 4 + qd;
          const u32x4 v = {__float_as_uint(acc[j][i][4 * qd]), __float_as_uint(acc[j][i][4 * qd + 1]),
                           __float_as_uint(acc[j][i][4 * qd + 2]), __float_as_uint(acc[j][i][4 * qd + 3])};
          __builtin_amdgcn_raw_buffer_store_b128(v, srsrc, tbase + q * (kThreads * 16u), split * per_slice, 16);   // sc1
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
      for (int i = 0; i < IM; ++i) keep_live(acc[j][i]);   // the stored registers stay untouched until the stores are done
    __syncthreads();
    int* flag = reinterpret_cast<int*>(lds);
    if (tid == 0) {
      const unsigned old = __hip_atomic_fetch_add(p.tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      flag[0] = old == (unsigned)a.splits - 1;
      if (old == (unsigned)a.splits - 1) __hip_atomic_store(p.tickets + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const int last = flag[0];
    __syncthreads();
    if (!last) return;
    // fixed summation order 0, 1, ..., splits - 1 whoever arrived last (the own slice comes back from memory like the others)
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
      for (int i = 0; i < IM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
    // SB slices' slabs in flight at a time (one slice per round trip was up to six dependent round trips here); slots past the
    // last slice re-read it and add zero.  Still summed in slice order.
    constexpr int SB = QUADS <= 4 ? 4 : (QUADS <= 8 ? 2 : 1);
    for (int s0 = 0; s0 < a.splits; s0 += SB) {
      u32x4 v[SB][QUADS];
#pragma unroll
      for (int sb = 0; sb < SB; ++sb) {
        const unsigned soff = (unsigned)min(s0 + sb, a.splits - 1) * per_slice;
#pragma unroll
        for (int q = 0; q < QUADS; ++q)
          v[sb][q] = __builtin_amdgcn_raw_buffer_load_b128(srsrc, tbase + q * (kThreads * 16u), soff, 16);         // sc1
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

  float ln_m[IM], ln_r[IM];
#pragma unroll
  for (int i = 0; i < IM; ++i) {
    ln_m[i] = LN ? ln_mean[i] : 0.f;
    ln_r[i] = LN ? ln_rstd[i] : 1.f;
  }
  GA_STAMP(st, 4);
  lin_epilogue<T, BM, BN, GEGLU, LN>(acc, lds, Y, a, p, m0, n0, nt, ln_m, ln_r, pre, st);
#if defined(GA_LIN_STAMPS)
  st.at(6);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  st.at(7);
  st.real(9);
  st.flush();
#endif
}

template <typename T, int BM, int BN, int NSTAGE>
int launch_lin(const T* X, const T* W, T* Y, LinArgs a, const LinPtrs& p, hipStream_t s) {
  const int outc = a.F ? BN / 2 : BN, n_out = a.F ? a.F : a.N;
  a.tm = (a.M + BM - 1) / BM;
  a.tn = (n_out + outc - 1) / outc;
  a.steps = a.K / kBK;
  a.steps_per = (a.steps + a.splits - 1) / a.splits;
  {  // bytes that reach the fabric if each XCD fetches what its run of workgroups shares once (as in conv3x3.hip)
    const double xb = (double)a.M * a.K, wb = (double)a.N * a.K;
    const double m_first = wb + xb * (a.tn * a.splits < 8 ? a.tn * a.splits : 8), n_first = xb + wb * (a.tm < 8 ? a.tm : 8);
    a.n_fastest = n_first < m_first ? 1 : 0;
  }
  {
    bool ok = true;
    const unsigned long long wgs = (unsigned long long)a.tm * a.tn * a.splits;
    a.d_tm = make_fastdiv(a.tm, wgs, ok);
    a.d_tn = make_fastdiv(a.tn, wgs, ok);
    if (!ok) return GA_ERR_SHAPE;   // workgroups x tiles >= 2^32: no such launch below the 2^31-byte operand limits
  }
  const dim3 grid((unsigned)(a.tm * a.tn * a.splits));
  const bool ln = a.ln_parts > 0;
  // the packed head (LinHead): 16-bit fields — K, N <= 65535 elements, <= 65535 column tiles / k-steps (checked here)
  if (a.K > 0xffff || a.N > 0xffff || a.tn > 0xffff || a.splits > 0xffff || a.steps > 0xffff || a.steps_per > 0x7fff) return GA_ERR_SHAPE;
  const LinHead h{a.M, a.ldx, (unsigned)a.K | ((unsigned)a.N << 16), a.tm, (unsigned)a.tn | ((unsigned)a.splits << 16),
                  (unsigned)a.steps | ((unsigned)a.steps_per << 16) | ((unsigned)(a.n_fastest != 0) << 31), a.d_tm.m, a.d_tn.m};
#define GA_LIN_LAUNCH(G, L)                                                                                                \
  hipLaunchKernelGGL((linear_kernel<T, BM, BN, NSTAGE, G, L>), grid, dim3(kThreads), 0, s, X, W, Y, h.M, h.ldx, h.kn, h.tm, \
                     h.tn_splits, h.steps, h.dtm_m, h.dtn_m, a, p)
  if (a.F) {
    if (ln) GA_LIN_LAUNCH(true, true);
    else GA_LIN_LAUNCH(true, false);
  } else {
    if (ln) GA_LIN_LAUNCH(false, true);
    else GA_LIN_LAUNCH(false, false);
  }
#undef GA_LIN_LAUNCH
  return check_launch();
}

// ring depth per tile: `stages` = 0 takes the first listed.  The shallower rings fit two workgroups per CU (LDS <= 80 KB):
// with one wave per SIMD nothing covers a wave's LDS-DMA issue (8 pieces per k-step cost about as much issue time as its
// 16 MFMAs), with two the partner's MFMAs do — what the large batch-3 shapes want; the deep rings hide more memory
// latency per workgroup — what the short small-M shapes want.
template <typename T>
int lin_t(const void* X, const void* W, void* Y, const LinArgs& a, const LinPtrs& p, int bm, int bn, int stages, hipStream_t s) {
  const T* x = (const T*)X;
  const T* w = (const T*)W;
  if (bm == 128 && bn == 128) {
    if (stages == 2) return launch_lin<T, 128, 128, 2>(x, w, (T*)Y, a, p, s);
    if (stages == 0 || stages == 3) return launch_lin<T, 128, 128, 3>(x, w, (T*)Y, a, p, s);
  } else if ((bm == 128 && bn == 64) || (bm == 64 && bn == 128)) {
    const bool tall = bm == 128;
    if (stages == 3) return tall ? launch_lin<T, 128, 64, 3>(x, w, (T*)Y, a, p, s) : launch_lin<T, 64, 128, 3>(x, w, (T*)Y, a, p, s);
    if (stages == 0 || stages == 4)
      return tall ? launch_lin<T, 128, 64, 4>(x, w, (T*)Y, a, p, s) : launch_lin<T, 64, 128, 4>(x, w, (T*)Y, a, p, s);
  } else if (bm == 64 && bn == 64) {
    if (stages == 5) return launch_lin<T, 64, 64, 5>(x, w, (T*)Y, a, p, s);   // K = 320: the whole depth in flight, 2 x 80 KB per CU
    if (stages == 0 || stages == 4) return launch_lin<T, 64, 64, 4>(x, w, (T*)Y, a, p, s);
  }
  return GA_ERR_SHAPE;
}

// =====================================================================================================================
// linear_stream_kernel — the same contraction, LayerNorm folded in front [+ GEGLU], for launches with MANY output tiles (the
// feed-forward and QKV / to_q GEMMs of the batch-3 and batch-2 passes: 300 - 1900 tiles of 128 x 128) as ONE persistent
// 512-thread workgroup per CU.  What the clock stamps of linear_kernel showed on these shapes (tools/micro/lin_stamps.py,
// profiles/r4_linear_stamps_before.txt; 12288 x 320 x 2560 + GEGLU, 9.8 us per tile, two tiles per CU at a time): a third is
// prologue (tile map, addresses, LayerNorm statistics, the first operand round trip), a third the five k-steps (ONE step in
// flight in the 2-slot ring that fits twice per CU: every step pays a memory round trip; 8 LDS-DMA pieces per wave and step
// cost more issue time than the step's 16 MFMAs), a third the epilogue (GEGLU through 16-bit LDS images).  Here:
//  * one workgroup per CU takes the LDS: a 4-slot ring (3 k-steps = 96 KB in flight) that keeps STREAMING ACROSS TILE BOUNDARIES
//    — while a tile's epilogue runs, the next tile's first steps are landing; per-tile set-up is a few address computations;
//  * 8 waves (2 per SIMD), each 32 tokens x 64 columns: 4 LDS-DMA pieces per wave and step instead of 8, and a partner wave's
//    MFMAs under every wave's DMA issue and epilogue arithmetic;
//  * GEGLU in registers: a wave owns 32 h columns AND their 32 gate columns, so h * gelu(gate) is formed from the f32
//    accumulators (the projection is never rounded to 16 bits in between; half the staging traffic);
//  * EVERY memory operation of the loop is an LDS-DMA (operands, and per tile: the LayerNorm shift / colsum of its columns, the
//    rows' partial sums), every LDS access inline asm: vmcnt retires in order, so one ordinary load waited for inside the loop
//    would drain the ring (and hipcc waits vmcnt(0) in front of any LDS access it can see while an LDS-DMA is pending).  The
//    per-tile constants are requested when the tile's FIRST k-step is consumed and read >= 3 steps later: the ring's own counted
//    waits have covered them by then.
// Built, measured and removed (profiles/r4_linear_stream_stamps_deferred_epilogue.txt): the GEGLU epilogue of tile j deferred into
// tile j + 1's k-steps from a second accumulator set, the two waves of a SIMD taking turns (one's gelu arithmetic under the other's
// MFMAs) — 41.9 against 37.3 us on 12288 x 320 x 2560: per k-step the kernel waits 1200 cycles for its 32 KB of operands (27 bytes
// per clock and CU from L2) whatever sits beside the MFMAs; the epilogue was never the bound, the operand stream is.
// Serves: LayerNorm fold (always), optional GEGLU, no bias / residual / split-K / pre-activation copy / statistics output — the
// no-grad passes' forms; K >= 320 (five k-steps: see the waits), 2 <= partial sums per row <= 20.
constexpr int kSThreads = 512;
constexpr int kSStage = 256 * kBK * 2;                 // bytes of one ring slot: 128 token rows + 128 weight rows of 128 bytes
constexpr int kSNst = 4, kSPre = kSNst - 1;
constexpr int kSOutStride = 144;                       // staging row: 64 outputs + one 16-byte vector (8-byte writes of 16
                                                       // consecutive rows and 16-byte reads fall on distinct banks)
constexpr int kSPartsMax = 20;
constexpr int kSStageOff = kSNst * kSStage;            // output staging, overlaid by the rows' LayerNorm partial sums
constexpr int kSStageBytes = 128 * 16 * (kSPartsMax / 2);   // 20480 >= 128 * kSOutStride
constexpr int kSCstOff = kSStageOff + kSStageBytes;    // shift[128] | colsum[128] of the tile's columns (f32)
constexpr int kSStatOff = kSCstOff + 1024;             // (mean, rstd)[128] of the tile's rows
constexpr int kSLds = kSStatOff + 1024;
static_assert(128 * kSOutStride <= kSStageBytes && kSLds <= 160 * 1024, "LDS map of linear_stream_kernel");

__device__ __forceinline__ void lds_read64(f32x2& dst, unsigned byte_address) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("ds_read_b64 %0, %1" : "=v"(dst) : "v"(byte_address) : "memory");
#else
  dst = f32x2{(float)byte_address, 0.f};
#endif
}
__device__ __forceinline__ void lds_read128f(f32x4& dst, unsigned byte_address) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(byte_address) : "memory");
#else
  dst = f32x4{(float)byte_address, 0.f, 0.f, 0.f};
#endif
}
__device__ __forceinline__ void lds_write64(unsigned byte_address, f32x2 v) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("ds_write_b64 %0, %1" ::"v"(byte_address), "v"(v) : "memory");
#endif
}

template <typename T>
__device__ __forceinline__ f32x2 pack4(float a, float b, float c, float d) {   // four results as 8 bytes of T
  typename Traits<T>::frag f;
  f[0] = Traits<T>::from_f32(a);
  f[1] = Traits<T>::from_f32(b);
  f[2] = Traits<T>::from_f32(c);
  f[3] = Traits<T>::from_f32(d);
  return __builtin_bit_cast(f32x2, f);
}

template <typename T, bool GEGLU>
__global__ __launch_bounds__(kSThreads, 2) void linear_stream_kernel(
    const T* __restrict__ X, const T* __restrict__ W, T* __restrict__ Y, int h_M, int h_ldx, unsigned h_kn, int h_tm,
    unsigned h_tn_splits, unsigned h_steps, unsigned h_dtm_m, unsigned h_dtn_m, LinArgs a_in, LinPtrs p) {
  // the 14 preloaded scalar dwords of linear_kernel; `splits` carries the grid size (one workgroup per CU)
  const int M = h_M, ldx = h_ldx, K = (int)(h_kn & 0xffffu), N = (int)(h_kn >> 16);
  const int tm = h_tm, tn = (int)(h_tn_splits & 0xffffu), G = (int)(h_tn_splits >> 16);
  const int steps = (int)(h_steps & 0xffffu);
  const bool n_fastest = (h_steps >> 31) != 0;
  const FastDiv d_tm{(unsigned)tm, h_dtm_m}, d_tn{(unsigned)tn, h_dtn_m};
  const int F = GEGLU ? N / 2 : 0, n_out = GEGLU ? F : N;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[kSLds];   // the ONE LDS object of the kernel
  const unsigned lds0 = (unsigned)(uintptr_t)GA_LDS_PTR(lds);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, fr = lane & 31, fh = lane >> 5;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);

  // ---- this workgroup's tiles: logical tile L = j * G + rb, rb = its place in an XCD-contiguous order of the G workgroups
  // (workgroup ids equal mod 8 share an XCD: each XCD works on a contiguous run of every band of G tiles — shared operand panels
  // meet in one L2)
  const int total_tiles = tm * tn;
  int rb;
  {
    const int q = G >> 3, r = G & 7, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    rb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int my_tiles = rb < total_tiles ? (total_tiles - rb + G - 1) / G : 0;
  const int total = my_tiles * steps;                       // k-steps this workgroup streams
  auto tile_origin = [&](int j, int& m0, int& n0) {
    const int L = j * G + rb;
    int mt, nt;
    if (n_fastest) {
      mt = fdiv(L, d_tn);
      nt = L - mt * tn;
    } else {
      nt = fdiv(L, d_tm);
      mt = L - nt * tm;
    }
    m0 = mt * 128;
    n0 = nt * (GEGLU ? 64 : 128);
  };
  // feature (row of W) behind tile column c (0 .. 127), clamped into the matrix: GEGLU tiles hold 64 h columns, then their gates
  auto feature = [&](int n0, int c, int last) {
    if (GEGLU) return c < 64 ? min(n0 + c, F - 1 - last) : F + min(n0 + c - 64, F - 1 - last);
    return min(n0 + c, N - 1 - last);
  };

  // ---- operand stream: piece q of a wave fills rows 8 g .. 8 g + 7 of a slot, g = wave + 8 q (g < 16: tokens, else weights);
  // lane l lands at row 8 g + (l >> 3), physical 16-byte column l & 7, and fetches logical column (l & 7) ^ swizzle(row)
  const __amdgpu_buffer_rsrc_t xrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(X), 0, (unsigned)(((M - 1) * ldx + K) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(W), 0, (unsigned)(N * K * 2), 0x00020000);
  unsigned voff[4];
  int ij = 0, ik = 0, gi = 0;                               // issue cursor: tile, k-step, steps issued
  auto set_issue_tile = [&](int j) {
    int m0, n0;
    tile_origin(j, m0, n0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = 8 * (wave + 8 * q) + (lane >> 3);
      const int col = (lane & 7) ^ ((row >> 1) & 7);
      if (q < 2) voff[q] = (unsigned)((min(m0 + row, M - 1) * ldx + 8 * col) * 2);
      else voff[q] = (unsigned)((feature(n0, row - 128, 0) * K + 8 * col) * 2);
    }
  };
  // piece q (of 4) of the next k-step of the stream into its ring slot; `advance` moves the cursor behind the fourth
  auto issue_piece = [&](int q) {
    const unsigned koff = (unsigned)(ik * kBK * 2);
    unsigned char* dst = lds + (gi & (kSNst - 1)) * kSStage + (wave_u + 8 * q) * 1024;
    if (q < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, GA_LDS_PTR(dst), 16, (int)voff[q], (int)koff, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, GA_LDS_PTR(dst), 16, (int)voff[q], (int)koff, 0, 0);
  };
  auto advance = [&]() {
    ++gi;
    if (++ik == steps) {
      ik = 0;
      if (++ij < my_tiles) set_issue_tile(ij);
    }
  };
  auto issue = [&]() {
#pragma unroll
    for (int q = 0; q < 4; ++q) issue_piece(q);
    advance();
  };
  // per-tile constants, by LDS-DMA with per-lane source addresses: wave 0 brings shift | colsum of the tile's 128 columns (one
  // piece: lanes 0-31 shift, 32-63 colsum, four floats each), the partial sums of the 128 rows go out as pieces of 64 rows x
  // two parts (a 16-byte load from an 8-byte aligned address), dealt over the waves.  With an odd part count the last piece
  // starts one part early instead of reading past the row (the statistics below skip the part it repeats).
  const int parts = a_in.ln_parts, chunks = (parts + 1) >> 1;
  auto issue_constants = [&](int m0, int n0) {
    if (wave_u == ((2 * chunks) & 7)) {   // the wave behind the ones that carry the partial pieces
      const int c = 4 * (lane & 31);
      const float* src = (lane < 32 ? p.ln_shift : p.ln_colsum) + feature(n0, c, 3);
      __builtin_amdgcn_global_load_lds(src, GA_LDS_PTR(lds + kSCstOff), 16, 0, 0);
    }
    for (int pi = wave_u; pi < 2 * chunks; pi += 8) {
      const int c = pi >> 1, rh = pi & 1;
      const int m = min(m0 + 64 * rh + lane, M - 1);
      const float* src = p.ln_partials + ((size_t)m * parts + min(2 * c, parts - 2)) * 2;
      __builtin_amdgcn_global_load_lds(src, GA_LDS_PTR(lds + kSStageOff + (c * 128 + 64 * rh) * 16), 16, 0, 0);
    }
  };
  // (mean, rstd) of the tile's rows from the partial sums: every wave takes 16 rows, four lanes a row (each lane the pieces
  // equal to its index mod 4), the four are added by quad permutes.  (The first form — threads 0 .. 127 a row each, piece by
  // piece — kept two of the eight waves busy for 0.85 us per tile while the other six waited at the next barrier.)
  auto row_statistics = [&]() {
    const int row = 16 * wave + (lane >> 2), sub = lane & 3;
    f32x4 v[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) lds_read128f(v[u], lds0 + kSStageOff + (unsigned)((min(sub + 4 * u, chunks - 1) * 128 + row) * 16));
    wait_lgkmcnt<0>();
    __builtin_amdgcn_sched_barrier(0);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int c = sub + 4 * u;
      const bool in = c < chunks, first = in && 2 * c + 1 < parts;   // odd count, last piece: its first part belongs to the piece before
      s1 += (first ? v[u][0] : 0.f) + (in ? v[u][2] : 0.f);
      s2 += (first ? v[u][1] : 0.f) + (in ? v[u][3] : 0.f);
    }
    s1 = group_sum<4>(s1);
    s2 = group_sum<4>(s2);
    const float mean = s1 * a_in.ln_inv_k;
    const float rstd = rsqrtf(fmaxf(s2 * a_in.ln_inv_k - mean * mean, 0.f) + a_in.ln_eps);
    if (sub == 0) lds_write64(lds0 + kSStatOff + (unsigned)(row * 8), f32x2{mean, rstd});
  };

  // ---- fragment addresses inside slot 0 (see linear_kernel): tokens = B operand, weights = A operand
  unsigned a_adr, a_sw, b_adr[2], b_sw[2];
  {
    const int row = wm * 32 + fr;
    a_adr = lds0 + row * 128;
    a_sw = (unsigned)(((row >> 1) & 7) ^ fh);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int wrow = 128 + j * 64 + wn * 32 + fr;
      b_adr[j] = lds0 + wrow * 128;
      b_sw[j] = (unsigned)(((wrow >> 1) & 7) ^ fh);
    }
  }
  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

#if defined(GA_LIN_STAMPS)
  LinStamps stm;   // [0] start [1] prologue done [2..5] SUMS over the stream: wait + barrier, k-step bodies, statistics, epilogues [7] end
  unsigned long long acc_t[4] = {0, 0, 0, 0}, tp;
  stm.real(8);
  stm.at(0);
#define GA_SSTAMP(i) do { stm.at(6); acc_t[i] += stm.t[6] - tp; tp = stm.t[6]; } while (0)
#else
#define GA_SSTAMP(i) ((void)0)
#endif
  if (my_tiles > 0) set_issue_tile(0);
#pragma unroll
  for (int s = 0; s < kSPre; ++s)
    if (gi < total) issue();
  int cj = 0, ck = 0, m0c = 0, n0c = 0;                     // consume cursor
#if defined(GA_LIN_STAMPS)
  stm.at(1);
  tp = stm.t[1];
#endif
  for (int g = 0; g < total; ++g) {
    // my pieces of step g have landed when at most the younger steps' pieces are outstanding (the few constant pieces issued in
    // between only make this wait for a little more than it needs)
    const int younger = min(total - 1 - g, kSPre - 1);
    if (younger >= 2) wait_vmcnt<8>();
    else if (younger == 1) wait_vmcnt<4>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();   // step g is in LDS for everyone; everyone is done with step g - 1 (and with the last epilogue)
    GA_SSTAMP(0);
    if (ck == 0) {
      tile_origin(cj, m0c, n0c);
      issue_constants(m0c, n0c);
    }
    // The refill of the slot step g - 1 occupied goes out BETWEEN this step's MFMAs, one piece behind each of the four
    // sub-steps: the eight waves leave the barrier together, and eight waves issuing 32 pieces at once, then computing at once,
    // took the sum of the two (the first form of this kernel: 41 us on 12288 x 320 x 2560 against 44 for the per-tile kernel).
    const bool refill = gi < total;
    if (ck == 3) row_statistics();  // the constants went out three steps ago, in front of a step this wave has now waited for
    GA_SSTAMP(2);
    const unsigned slot_off = (unsigned)((g & (kSNst - 1)) * kSStage);
    u32x4 fa[2], fb[2][2];
    auto request = [&](int kk, int set) {
      lds_read128(fa[set], a_adr + slot_off + 16u * ((2u * kk) ^ a_sw));
#pragma unroll
      for (int j = 0; j < 2; ++j) lds_read128(fb[set][j], b_adr[j] + slot_off + 16u * ((2u * kk) ^ b_sw[j]));
    };
    request(0, 0);
#pragma unroll
    for (int kk = 0; kk < kBK / 16; ++kk) {
      if (kk + 1 < kBK / 16) {
        request(kk + 1, (kk + 1) & 1);
        wait_lgkmcnt<3>();
      } else {
        wait_lgkmcnt<0>();
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[j] = Mma32L<T>::run(__builtin_bit_cast(uint4, fb[kk & 1][j]), __builtin_bit_cast(uint4, fa[kk & 1]), acc[j]);
      if (refill) issue_piece(kk);
    }
    if (refill) advance();
    GA_SSTAMP(1);
    if (++ck < steps) continue;

    // ---- epilogue of tile cj (the ring keeps landing the next tile meanwhile)
    ck = 0;
    ++cj;
    f32x2 st;                                               // (mean, rstd) of this lane's token row
    lds_read64(st, lds0 + kSStatOff + (wm * 32 + fr) * 8);
    f32x4 sh[2][4], cs[2][4];                               // shift / colsum of this lane's columns: [h | gate or half][quad]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const unsigned c = (unsigned)(j * 64 + wn * 32 + 8 * qd + 4 * fh);
        lds_read128f(sh[j][qd], lds0 + kSCstOff + 4u * c);
        lds_read128f(cs[j][qd], lds0 + kSCstOff + 512u + 4u * c);
      }
    wait_lgkmcnt<0>();
    __builtin_amdgcn_sched_barrier(0);
    const float mean = st[0], rstd = st[1];
    const unsigned out_adr = lds0 + kSStageOff + (unsigned)((wm * 32 + fr) * kSOutStride + (wn * 32 + 4 * fh) * 2);
    constexpr int PASSES = GEGLU ? 1 : 2;
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
      if (pass > 0) {
        __builtin_amdgcn_s_barrier();                       // the first half has left the staging rows
      }
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if constexpr (GEGLU) {
            const float hv = rstd * (acc[0][4 * qd + r] - mean * cs[0][qd][r]) + sh[0][qd][r];
            const float gv = rstd * (acc[1][4 * qd + r] - mean * cs[1][qd][r]) + sh[1][qd][r];
            o[r] = gv;
            acc[0][4 * qd + r] = hv;
          } else {
            o[r] = rstd * (acc[pass][4 * qd + r] - mean * cs[pass][qd][r]) + sh[pass][qd][r];
          }
        }
        if constexpr (GEGLU) {
          const f32x2 g01 = gelu_erf2(f32x2{o[0], o[1]}), g23 = gelu_erf2(f32x2{o[2], o[3]});
          o[0] = acc[0][4 * qd] * g01.x;
          o[1] = acc[0][4 * qd + 1] * g01.y;
          o[2] = acc[0][4 * qd + 2] * g23.x;
          o[3] = acc[0][4 * qd + 3] * g23.y;
        }
        lds_write64(out_adr + 16u * qd, pack4<T>(o[0], o[1], o[2], o[3]));
      }
      wait_lgkmcnt<0>();
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int v = tid + k * kSThreads, r = v >> 3, cv = (v & 7) * 8;
        u32x4 val;
        lds_read128(val, lds0 + kSStageOff + (unsigned)(r * kSOutStride + cv * 2));
        wait_lgkmcnt<0>();
        __builtin_amdgcn_sched_barrier(0);
        const int m = m0c + r, n = n0c + pass * 64 + cv;
        if (m < M && n < n_out) *reinterpret_cast<u32x4*>(Y + (size_t)m * a_in.ldy + n) = val;
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    GA_SSTAMP(3);
  }
#if defined(GA_LIN_STAMPS)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  stm.at(7);
  stm.real(9);
  for (int i = 0; i < 4; ++i) stm.t[2 + i] = acc_t[i];
  stm.t[6] = (unsigned long long)my_tiles;
  stm.flush();
#endif
}

template <typename T>
int launch_stream(const T* X, const T* W, T* Y, LinArgs a, const LinPtrs& p, hipStream_t s) {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8)
      return GA_ERR_LAUNCH;
    cus = n;
  }
  const int outc = a.F ? 64 : 128, n_out = a.F ? a.F : a.N;
  a.tm = (a.M + 127) / 128;
  a.tn = (n_out + outc - 1) / outc;
  a.steps = a.K / kBK;
  const long long tiles = (long long)a.tm * a.tn;
  if (tiles > 0x7fffffffLL / 256 || a.K > 0xffff || a.N > 0xffff || a.tn > 0xffff || a.steps > 0xffff) return GA_ERR_SHAPE;
  const int G = tiles < cus ? (int)tiles : cus;
  {
    const double xb = (double)a.M * a.K, wb = (double)a.N * a.K;
    const double m_first = wb + xb * (a.tn < 8 ? a.tn : 8), n_first = xb + wb * (a.tm < 8 ? a.tm : 8);
    a.n_fastest = n_first < m_first ? 1 : 0;
  }
  bool ok = true;
  const unsigned long long span = (unsigned long long)tiles + (unsigned long long)G;
  a.d_tm = make_fastdiv(a.tm, span, ok);
  a.d_tn = make_fastdiv(a.tn, span, ok);
  if (!ok) return GA_ERR_SHAPE;
  const LinHead h{a.M, a.ldx, (unsigned)a.K | ((unsigned)a.N << 16), a.tm, (unsigned)a.tn | ((unsigned)G << 16),
                  (unsigned)a.steps | ((unsigned)(a.n_fastest != 0) << 31), a.d_tm.m, a.d_tn.m};
  if (a.F)
    hipLaunchKernelGGL((linear_stream_kernel<T, true>), dim3((unsigned)G), dim3(kSThreads), 0, s, X, W, Y, h.M, h.ldx, h.kn, h.tm,
                       h.tn_splits, h.steps, h.dtm_m, h.dtn_m, a, p);
  else
    hipLaunchKernelGGL((linear_stream_kernel<T, false>), dim3((unsigned)G), dim3(kSThreads), 0, s, X, W, Y, h.M, h.ldx, h.kn, h.tm,
                       h.tn_splits, h.steps, h.dtm_m, h.dtn_m, a, p);
  return check_launch();
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

#if defined(GA_LIN_STAMPS)
extern "C" int ga_lin_set_stamps(void* buffer) {   // [workgroups of the largest launch][10] u64, or NULL
  unsigned long long* p = static_cast<unsigned long long*>(buffer);
  return hipMemcpyToSymbol(HIP_SYMBOL(g_lin_stamps), &p, sizeof(p)) == hipSuccess ? GA_OK : GA_ERR_LAUNCH;
}
#endif

extern "C" int ga_linear_workspace(int64_t M, int N, int bm, int bn, int splits, int geglu, long long* slab_floats, int* tiles) {
  if (!slab_floats || !tiles) return GA_ERR_NULL;
  if (M < 1 || N < 8 || splits < 1 || (bm != 64 && bm != 128) || (bn != 64 && bn != 128)) return GA_ERR_SHAPE;
  const int outc = geglu ? bn / 2 : bn, n_out = geglu ? N / 2 : N;
  const long long tm = (M + bm - 1) / bm, tn = (n_out + outc - 1) / outc;
  *tiles = (int)(tm * tn);
  *slab_floats = splits > 1 ? (long long)splits * tm * tn * bm * bn : 0;
  return GA_OK;
}

extern "C" int ga_linear_gn_blocks(int hw, int N, int groups, int bm, int bn) {
  /* partial blocks per image a ga_linear_fused call with gn_partials writes (2 slots per m tile: the n tile a group starts in and
   * the one it continues into), or 0 when the shape is not served: m tiles must not straddle images, a group's channels must
   * lie in at most two n tiles and a 16-byte vector in at most two groups */
  if (hw < 1 || N < 8 || groups < 1 || N % groups != 0 || (bm != 64 && bm != 128) || (bn != 64 && bn != 128)) return 0;
  const int cg = N / groups;
  if (hw % bm != 0 || cg < 8 || cg > bn || groups > 64) return 0;
  const int blocks = 2 * (hw / bm);
  return blocks <= 128 ? blocks : 0;
}

extern "C" int ga_linear_fused(const void* X, int64_t ldx, const void* W, void* Y, int64_t ldy, const ga_linear_epilogue_t* ep,
                               float* slabs, unsigned* tickets, int64_t M, int K, int N, int bm, int bn, int splits,
                               int stages, int dtype, ga_stream_t stream) {
  if (!X || !W || !Y || !ep) return GA_ERR_NULL;
  if (M < 1 || K < kBK || K % kBK != 0 || N < 8 || N % 8 != 0 || ldx < K || ldx % 8 != 0 || ldy % 8 != 0) return GA_ERR_SHAPE;
  if (splits < 1 || splits > 64 || K / kBK < splits || (splits > 1 && (!slabs || !tickets))) return GA_ERR_SHAPE;
  const int n_out = ep->geglu ? N / 2 : N;
  if (ep->geglu && (N % 16 != 0 || ep->residual)) return GA_ERR_SHAPE;
  if (ldy < n_out || (ep->residual && (ep->ld_res < n_out || ep->ld_res % 8 != 0))) return GA_ERR_SHAPE;
  if (ep->preact && (!ep->geglu || ep->ld_pre < N || ep->ld_pre % 8 != 0)) return GA_ERR_SHAPE;
  if (ep->ln_partials && (ep->ln_parts < 1 || !ep->ln_colsum || !ep->ln_shift)) return GA_ERR_NULL;
  if (!al16(X) || !al16(W) || !al16(Y) || (ep->bias && !al16(ep->bias)) || (ep->residual && !al16(ep->residual)) ||
      (ep->preact && !al16(ep->preact)) || (ep->ln_colsum && !al16(ep->ln_colsum)) || (ep->ln_shift && !al16(ep->ln_shift)))
    return GA_ERR_ALIGN;
  const long long xb = ((long long)(M - 1) * ldx + K) * 2, wb = (long long)N * K * 2;
  if (xb >= (1LL << 31) || wb >= (1LL << 31)) return GA_ERR_SHAPE;   // 32-bit byte offsets in the buffer loads
  LinArgs a;
  a.M = (int)M; a.N = N; a.K = K;
  a.ldx = (int)ldx; a.ldy = (int)ldy; a.ld_res = (int)ep->ld_res; a.ld_pre = (int)ep->ld_pre;
  a.splits = splits;
  a.F = ep->geglu ? N / 2 : 0;
  a.ln_parts = ep->ln_partials ? ep->ln_parts : 0;
  a.ln_eps = ep->ln_eps;
  a.ln_inv_k = 1.0f / (float)K;
  a.x_bytes = (unsigned)xb;
  a.w_bytes = (unsigned)wb;
  if (splits > 1) {
    long long need; int tiles;
    if (ga_linear_workspace(M, N, bm, bn, splits, ep->geglu, &need, &tiles) != GA_OK) return GA_ERR_SHAPE;
    if (need * 4 >= (1LL << 31)) return GA_ERR_SHAPE;
  }
  LinPtrs p;
  p.bias = ep->bias; p.residual = ep->residual; p.preact = ep->preact;
  p.ln_partials = ep->ln_partials; p.ln_colsum = ep->ln_colsum; p.ln_shift = ep->ln_shift;
  p.ln_stats_out = ep->ln_stats_out; p.row_partials_out = ep->row_partials_out;
  p.slabs = slabs; p.tickets = tickets;
  p.gn_partials = nullptr; p.gn_cg = p.gn_G = p.gn_hw = p.gn_tpi = 0;
  if (ep->gn_partials) {
    if (ep->geglu || stages == GA_LINEAR_STREAM) return GA_ERR_UNSUPPORTED;
    if (ga_linear_gn_blocks(ep->gn_hw, N, ep->gn_groups, bm, bn) == 0 || M % ep->gn_hw != 0) return GA_ERR_SHAPE;
    p.gn_partials = ep->gn_partials; p.gn_G = ep->gn_groups; p.gn_cg = N / ep->gn_groups; p.gn_hw = ep->gn_hw;
    p.gn_tpi = ep->gn_hw / bm;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (stages == GA_LINEAR_STREAM) {
    // the persistent one-workgroup-per-CU form (linear_stream_kernel): what it serves, it serves with the 128 x 128 tile only
    if (bm != 128 || bn != 128 || splits != 1 || !ep->ln_partials || ep->bias || ep->residual || ep->preact || ep->ln_stats_out ||
        ep->row_partials_out || K / kBK < 5 || ep->ln_parts < 2 || ep->ln_parts > kSPartsMax || ldy > 0x7fffffff)
      return GA_ERR_UNSUPPORTED;
    if ((long long)M * ep->ln_parts * 8 >= (1LL << 31)) return GA_ERR_SHAPE;
    switch (dtype) {
      case GA_F16: return launch_stream<_Float16>((const _Float16*)X, (const _Float16*)W, (_Float16*)Y, a, p, s);
      case GA_BF16: return launch_stream<bf16_t>((const bf16_t*)X, (const bf16_t*)W, (bf16_t*)Y, a, p, s);
      default: return GA_ERR_DTYPE;
    }
  }
  switch (dtype) {
    case GA_F16: return lin_t<_Float16>(X, W, Y, a, p, bm, bn, stages, s);
    case GA_BF16: return lin_t<bf16_t>(X, W, Y, a, p, bm, bn, stages, s);
    default: return GA_ERR_DTYPE;
  }
}
