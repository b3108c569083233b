/* ga_hip.h — C ABI of libga_hip.so: the MI355X (gfx950) kernels behind the guided-attention hot path.
 *
 * The reference (jackBonadies/Guided-Attention) is pure Python and has no FFI: the path sits behind
 * three Python protocols (attention-processor, controller/AttentionStore, pipeline methods).  This
 * header is the C boundary a maintainer would bind underneath those protocols (ctypes stub in
 * INTEGRATION.md).  Each entry point names the reference code it replaces (file:line relative to
 * the reference checkout).
 *
 * Conventions (all entry points):
 *   - return 0 (GA_OK) or a negative ga_status; never throw, never allocate, never synchronise;
 *   - work is enqueued on `stream` (a hipStream_t; NULL = the default stream);
 *   - every tensor pointer is a DEVICE pointer to contiguous memory in the stated layout, owned by
 *     the caller; small descriptor structs (`ga_token_t`, `ga_loss_params_t`, pointer lists) are HOST
 *     memory and are consumed before the call returns;
 *   - dtype is a ga_dtype; "T" below means that element type.  Accumulation is always f32.
 */
#ifndef GA_HIP_H
#define GA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever an exported signature changes or an export is added; a binder checks ga_version() against the header it
 * was written for BEFORE its first call (guided-attention_amd/_lib.py:load does) — a stale binding passes pointers in the
 * wrong positions.  History:
 *   120  0.1.2  strict bbox mode, paint-with-words entry points
 *   180  0.1.8  ga_linear_epilogue_t gained gn_partials / gn_groups / gn_hw at its END; new: ga_linear_gn_blocks
 *   170  0.1.7  new: ga_cat_channels_gn, ga_cat_channels_gn_blocks, ga_cat_group_norm_fwd, ga_group_norm_one_launch
 *   160  0.1.6  new: ga_conv3x3_thin_in, ga_conv3x3_thin_out, ga_conv3x3_thin_pack, ga_conv3x3_thin_packed_elems,
 *               ga_conv3x3_thin_supported (the UNet's conv_in / conv_out and their adjoints)
 *   150  0.1.5  new: ga_conv3x3_nhwc_gn, ga_conv3x3_gn_blocks, ga_group_norm_apply, ga_group_norm_two_launch
 *   140  0.1.4  ga_linear_fused: stages = GA_LINEAR_STREAM (persistent form); no signature changed
 *   130  0.1.3  (round 3, bumped late) ga_conv3x3_nhwc / ga_gemm_nt gained `tickets` behind `workspace`, ga_group_norm_bwd
 *               gained `g_res` before `dx`; new: ga_aggregate_loss_fwd, ga_linear_fused, ga_linear_workspace,
 *               ga_splitk_workspace_floats, ga_conv3x3_up2x_nhwc, ga_cat_channels, ga_conv3x3_packed_elems */
#define GA_VERSION 180

typedef void* ga_stream_t; /* hipStream_t */

typedef enum { GA_F16 = 0, GA_BF16 = 1, GA_F32 = 2 } ga_dtype;

typedef enum {
  GA_OK = 0,
  GA_ERR_NULL = -1,        /* a required pointer is NULL */
  GA_ERR_SHAPE = -2,       /* a size is out of the supported range */
  GA_ERR_DTYPE = -3,       /* unknown dtype */
  GA_ERR_ALIGN = -4,       /* a pointer is not 16-byte aligned / head_dim not a multiple of 8 */
  GA_ERR_LAUNCH = -5,      /* hipLaunchKernel reported an error */
  GA_ERR_UNSUPPORTED = -6  /* valid request this build does not implement */
} ga_status;

int ga_version(void);
const char* ga_strerror(int status);

/* ---------------------------------------------------------------------------------------------
 * K1  attention-store capture, forward.
 * Replaces utils/ptp_utils.py:77-86 (head split, get_attention_scores, store, bmm(P,V), head merge)
 * and utils/ptp_utils.py:97-146 (scores = scale*Q K^T, softmax over keys, cast back).
 *
 *   Q  [B][N][H][D]  T   the to_q projection as it comes out of the linear layer (no head transpose)
 *   K,V[B][Kt][H][D] T   the to_k / to_v projections of the text (or any) context, Kt <= 128
 *   O  [B][N][H][D]  T   softmax(scale Q K^T) V, already in "batch_to_head_dim" layout
 *   P  [B*H][N][Kt]  T   the attention probabilities exactly as the reference hands them to the
 *                        controller (head-major batch); NULL = do not materialise them
 * D must be a multiple of 8 and <= 256.
 */
int ga_attn_capture_fwd(const void* Q, const void* K, const void* V, void* O, void* P,
                        int B, int H, int N, int Kt, int D, float scale, int dtype, ga_stream_t stream);

/* K1 backward (the reference relies on torch autograd through baddbmm/softmax/bmm).
 *   dO [B][N][H][D] T ; dP (optional) direct gradient on P, element (bh, n, k) at
 *   dP[bh*dP_stride_bh + n*dP_stride_n + k] (strides in elements; dP_stride_bh = 0 broadcasts one
 *   [N][Kt] map over all B*H head-maps — the shape dLoss/dA takes after aggregate_attention);
 *   dQ [B][N][H][D] T.   P is recomputed from Q,K (bit-identical to the forward).
 *   dK/dV (gradients w.r.t. the context projections) are not needed on this path (the text
 *   embedding carries no gradient: pipeline_guided_attention.py:466) and must be NULL.
 */
int ga_attn_capture_bwd(const void* Q, const void* K, const void* V, const void* dO,
                        const void* dP, int64_t dP_stride_bh, int64_t dP_stride_n,
                        void* dQ, void* dK, void* dV,
                        int B, int H, int N, int Kt, int D, float scale, int dtype, ga_stream_t stream);

/* K1 with the paint-with-words bias (utils/ptp_utils.py:113-138; off by default, curHyperParams
 * "paint_with_words_stop" / "paint_with_words_weight"): the reference adds
 *     mask[n][k] * 0.4 * attention_scores.max() * log(1 + sigma_t)
 * to the scaled scores of the 77-token cross-attention before the softmax.  Split into what the device needs:
 *   ga_attn_scores_max      : packed[0] = max over ALL scaled scores of the call and its position,
 *                             (order-preserving bits of the f32 value << 32) | flat index into [B*H][N][Kt];
 *                             the caller zeroes packed[0] first (device memory, 8 bytes);
 *   ga_attn_capture_fwd_biased / _bwd_biased : as ga_attn_capture_fwd / _bwd with scores + bias[n][k] * coef[0]
 *                             (bias [N][Kt] T, shared by every batch entry and head; coef a DEVICE scalar so that
 *                             nothing synchronises); the backward also accumulates d loss / d coef = sum dS * bias into
 *                             bias_grad[0] (f32 atomics; may be NULL) — the gradient the reference's autograd sends
 *                             on through `.max()`, which the host adds to dQ at the maximum's position.
 * Kt <= 80 (the reference applies the mask to the text context only).
 * Ties: when several scores equal the maximum bit for bit, `packed` names the one with the HIGHEST flat index and the
 * whole d loss / d coef gradient goes to that position; torch's full-reduction `.max()` backward (what the reference
 * runs) spreads it evenly over the tied positions.  The two agree whenever the maximum is unique (every fixture and
 * pipeline test); a tie needs two identical (query, key) pairs.  tests/test_kernels_gpu.py pins this rule.
 */
int ga_attn_scores_max(const void* Q, const void* K, int B, int H, int N, int Kt, int D, float scale, int dtype,
                       unsigned long long* packed, ga_stream_t stream);
int ga_attn_capture_fwd_biased(const void* Q, const void* K, const void* V, void* O, void* P, const void* bias,
                               const float* coef, int B, int H, int N, int Kt, int D, float scale, int dtype,
                               ga_stream_t stream);
int ga_attn_capture_bwd_biased(const void* Q, const void* K, const void* V, const void* dO, const void* dP,
                               int64_t dP_stride_bh, int64_t dP_stride_n, void* dQ, const void* bias, const float* coef,
                               float* bias_grad, int B, int H, int N, int Kt, int D, float scale, int dtype,
                               ga_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K2  aggregate_attention (utils/ptp_utils.py:273-289, select = 0): the mean over every head-map of
 * every stored tensor with npix pixels, summed in list order.
 *   maps[i]  device pointer to [heads[i]][npix][Kt] T   (host array of n_maps pointers, n_maps <= 128)
 *   A        [npix][Kt] f32
 */
int ga_aggregate_maps(const void* const* maps, const int* heads, int n_maps, int npix, int Kt,
                      float* A, int dtype, ga_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K3+K4  the Gaussian-smoothed box loss.
 * Replaces pipeline_guided_attention.py:201-296 (_compute_max_attention_per_index),
 * utils/gaussian_smoothing.py:21-71, utils/helpers.py:164-173,215-277 (inside_box,
 * calculate_bounding_box_losses, strict and non-strict) and pipeline_guided_attention.py:359-451
 * (_compute_loss / get_centering_loss / group_losses_by_sumprompt).
 */
typedef enum { GA_TOK_COOR = 0, GA_TOK_BOX = 1 } ga_token_kind;

typedef struct {
  int32_t token;   /* position in the Kt-token prompt (BOS = 0), i.e. a key of config.token_dict */
  int32_t kind;    /* ga_token_kind */
  double geom[4];  /* BOX: x, y, width, height; COOR: x, y — fractions of the image (helpers.Rect size 1) */
  float weight;    /* 1, or 1/len(sub-prompt) when sub_prompt_avg_within */
  float _pad;
} ga_token_t;

typedef struct {
  float inside_scale;   /* curHyperParams["inside_loss_scale"]            */
  float outside_scale;  /* curHyperParams["outside_loss_scale"] (the x3 of pipeline:427 is applied inside) */
  float center_weight;  /* curHyperParams.get("bb_center_weight", .05); <= 0 disables the term */
  float sigma;          /* Gaussian sigma */
  double shrink;        /* curHyperParams["shrink_factor"] */
  int32_t ksize;        /* Gaussian kernel size (odd, <= 7; the reference only runs 3) */
  int32_t smooth;       /* smooth_attentions */
  int32_t strict;       /* curHyperParams["strict"]: the weighted hinge form of helpers.py:216-264 */
  int32_t _pad;
} ga_loss_params_t;

#define GA_TERMS 8 /* per token: max, col, row, inside, outside, token_loss, unscaled, sum(M) */

/*   A      [res*res][Kt] f32  aggregated maps (row-major pixels, token fastest)
 *   first,last  text-token slice [first,last) that is re-softmaxed (x100): 1 and Kt-1, or the first
 *          EOT index for SD-2.1 (pipeline:209-219)
 *   terms  [T][GA_TERMS] f32, loss [1] f32 (sum over tokens of weight*token_loss)
 * Limits: res <= 64, T <= 32, T*res*res <= 24576.
 */
int ga_smooth_loss_fwd(const float* A, int res, int Kt, int first, int last,
                       const ga_token_t* tokens, int T, const ga_loss_params_t* hp,
                       float* terms, float* loss, ga_stream_t stream);

/* Backward of the above w.r.t. A (recomputed from A; nothing is saved by the forward).
 *   dloss  [1] f32 device scalar multiplying the gradient, or NULL for 1
 *   dA     [res*res][Kt] f32
 *   dP_bcast (optional) [res*res][Kt] T = bcast_scale * dA, the map that ga_attn_capture_bwd
 *          broadcasts over the head-maps (bcast_scale = 1 / number of aggregated head-maps)
 */
int ga_smooth_loss_bwd(const float* A, int res, int Kt, int first, int last,
                       const ga_token_t* tokens, int T, const ga_loss_params_t* hp,
                       const float* dloss, float* dA, void* dP_bcast, float bcast_scale, int dtype,
                       ga_stream_t stream);

/* K2 + K3 + K4 in one launch: ga_aggregate_maps followed by ga_smooth_loss_fwd on its result
 * (utils/ptp_utils.py:279-289 -> pipeline_guided_attention.py:217-296, one launch instead of two on the serial path
 * of every refinement iteration).  Every workgroup averages 256 elements of A and stores them; the workgroup that
 * finishes last evaluates the loss on the complete map.
 *   A       [res*res][Kt] f32, written (kept: the backward and the diagnostics read it)
 *   ticket  one zero-initialised 32-bit word of device memory owned by the caller; the kernel leaves it zero.
 *           Launches sharing a ticket word must be stream-ordered.
 * Results are bit-identical to the two separate calls. */
int ga_aggregate_loss_fwd(const void* const* maps, const int* heads, int n_maps, int res, int Kt,
                          int first, int last, const ga_token_t* tokens, int T,
                          const ga_loss_params_t* hp, float* A, float* terms, float* loss,
                          unsigned* ticket, int dtype, ga_stream_t stream);

/* Gaussian weights exactly as utils/gaussian_smoothing.py:21-47 builds them (host helper; w[ksize*ksize]). */
int ga_gaussian_weights(int ksize, float sigma, float* w);

/* ---------------------------------------------------------------------------------------------
 * K5  latent update (pipeline_guided_attention.py:466-469): out = latents - step * grad, with the
 * log line's mean(|grad|) (pipeline:467) fused when absmean != NULL ([1] f32).  out may alias latents.
 */
int ga_latent_axpy(const void* latents, const void* grad, float step, void* out, float* absmean,
                   int64_t n, int dtype, ga_stream_t stream);

/* K6  out = a*x + b*y (re-noise back to level t, pipeline_guided_attention.py:1048-1053). */
int ga_latent_axpby(const void* x, const void* y, float a, float b, void* out, int64_t n, int dtype,
                    ga_stream_t stream);

/* Classifier-free-guidance combine + DDIM step (eta = 0), pipeline_guided_attention.py:1022-1029:
 *   eps = eps_uncond + g*(eps_text - eps_uncond); x0 = (x - sqrt(1-a_t) eps)/sqrt(a_t);
 *   prev = sqrt(a_prev) x0 + sqrt(1-a_prev) eps.   x0_out may be NULL.  prev may alias x. */
int ga_cfg_ddim_step(const void* eps_uncond, const void* eps_text, float guidance, const void* x,
                     float alpha_t, float alpha_prev, void* prev, void* x0_out, int64_t n, int dtype,
                     ga_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Tiled self-attention for long key sequences (the encoder_hidden_states = None case of the processor,
 * utils/ptp_utils.py:66-93 with :97-146): O = softmax(scale Q K^T) V without materialising P.
 *   Q,K,V,O,dO,dQ,dK,dV [B][N][H][D] T (projection layout); LSE, delta [B*H][N] f32 (LSE is written by the
 *   forward, in the log2 domain, and read by the backward; delta is scratch the backward fills).
 *   ld_qkv = row stride (elements) of Q, K, V and of dQ, dK, dV: 0 (= H*D) for separate projections, 3*H*D when
   they are the three column slices of one fused QKV projection (pass the slice base pointers); O / dO are dense.
   D % 8 == 0, D <= 160 (f32: D <= 80).  Backward = 2 launches (dQ, which also fills delta; then dK+dV), no atomics.
 */
int ga_self_attn_fwd(const void* Q, const void* K, const void* V, void* O, float* LSE,
                     int B, int H, int N, int D, int ld_qkv, float scale, int dtype, ga_stream_t stream);
int ga_self_attn_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO,
                     const float* LSE, float* delta, void* dQ, void* dK, void* dV,
                     int B, int H, int N, int D, int ld_qkv, float scale, int dtype, ga_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * UNet host helper: GroupNorm (+ fused SiLU) on channels-last activations, forward and backward to the
 * input (large levels: 2 launches each way — per-block partial sums, then an apply pass whose workgroups fold the partials themselves; <= 256 pixels: 1 launch).  Stands in for the GroupNorm -> SiLU pairs of the diffusers UNet blocks the reference runs in
 * pipeline_guided_attention.py:583-743 (diffusers 0.12.1 ResnetBlock2D / Transformer2DModel).
 *   x, y, dy, dx [B][HW][C] T (NHWC); gamma, beta [C] T; stats [B][G][2] f32 (mean, rstd), written by the
 *   forward and read by the backward; workspace GA_GN_WORKSPACE_FLOATS(B, G) f32 scratch.  C/G must be even, G <= 64,
 *   C <= 2560.  chan_bias (optional, [B][C] T): the layer normalises x + chan_bias[b][c] — the ResnetBlock's
 *   time-embedding term folded into the norm instead of a separate broadcast-add pass; it receives no gradient.
 *   gamma/beta gradients are not produced (weights are frozen on this path).
 *   g_res (optional, like dx): the gradient that reaches x through its OTHER consumer (the block's skip connection);
 *   dx = group-norm backward + g_res in the same pass instead of a separate accumulation launch.
 */
#define GA_GN_WORKSPACE_FLOATS(B, G) ((B) * 257 * (G) * 2)
int ga_group_norm_fwd(const void* x, const void* chan_bias, const void* gamma, const void* beta, void* y, float* stats,
                      float* workspace, int B, int HW, int C, int G, float eps, int act_silu, int dtype,
                      ga_stream_t stream);
/* The apply launch of the large-level forward on partial sums taken elsewhere (`partials` [B][blocks][G][2] f32, blocks <= 128:
 * what ga_conv3x3_nhwc_gn leaves): y, stats as ga_group_norm_fwd.  16-bit types, C % 8 == 0, C / G >= 8.
 * ga_group_norm_two_launch: 1 when ga_group_norm_fwd would take two launches for the shape (where the pair above saves one). */
int ga_group_norm_apply(const void* x, const void* chan_bias, const void* gamma, const void* beta, void* y, float* stats,
                        const float* partials, int blocks, int B, int HW, int C, int G, float eps, int act_silu, int dtype,
                        ga_stream_t stream);
int ga_group_norm_two_launch(int HW, int C, int G, int dtype);
int ga_group_norm_bwd(const void* x, const void* chan_bias, const void* dy, const void* gamma, const void* beta,
                      const float* stats, const void* g_res, void* dx, float* workspace, int B, int HW, int C, int G,
                      int act_silu, int dtype, ga_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * UNet host helpers: element-wise epilogues of the transformer feed-forward and the ResnetBlock (diffusers 0.12.1
 * GEGLU.forward / ResnetBlock2D.forward, run by the reference inside pipeline_guided_attention.py:583-743).
 *   ga_geglu_fwd : x [rows][2F] T (the GEGLU projection output: h = x[:, :F], gate = x[:, F:]) -> y [rows][F] T,
 *                  y = h * gelu(gate), exact (erf) GELU evaluated in f32.
 *   ga_geglu_bwd : dx [rows][2F] = (dy * gelu(gate) | dy * h * gelu'(gate)).
 *   ga_bias_residual_add : out = y + bias[c] + residual, rows x C, bias [C] optional (NULL); out may alias y or
 *                  residual.
 * F (C) must be a multiple of 16 / sizeof(T); pointers 16-byte aligned.
 */
int ga_geglu_fwd(const void* x, void* y, int64_t rows, int F, int dtype, ga_stream_t stream);
int ga_geglu_bwd(const void* x, const void* dy, void* dx, int64_t rows, int F, int dtype, ga_stream_t stream);
int ga_bias_residual_add(const void* y, const void* bias, const void* residual, void* out, int64_t rows, int C,
                         int dtype, ga_stream_t stream);

/* UNet host helper: out[m][0 .. C1) = a[m][:], out[m][C1 .. C1 + C2) = b[m][:] for `rows` rows — the channel concatenation
 * of the running activation with a skip connection on channels-last tensors (diffusers 0.12.1 CrossAttnUpBlock2D /
 * UpBlock2D `torch.cat([hidden_states, res_hidden_states], dim=1)`, run by the reference inside
 * pipeline_guided_attention.py:583-743).  a [rows][C1], b [rows][C2], out [rows][C1 + C2], all dense; elem_bytes 2 or 4;
 * C1, C2 multiples of 16 / elem_bytes; pointers 16-byte aligned; rows * (C1 + C2) * elem_bytes < 32 GiB. */
int ga_cat_channels(const void* a, const void* b, void* out, int64_t rows, int C1, int C2, int elem_bytes,
                    ga_stream_t stream);
/* The same concatenation that ALSO leaves the GroupNorm statistics of its result for the norm layer that consumes it (the
 * UpBlock's resnet.norm1): a [B][HW][C1], b [B][HW][C2] -> out [B][HW][C1 + C2] and per (image, pixel block, group) partial
 * (sum, sum of squares) [B][blocks][G][2] f32, blocks = ga_cat_channels_gn_blocks(HW, C1 + C2, G, dtype) (0: not served — that
 * norm is a single launch already, or off the wide path: use ga_cat_channels).  ga_group_norm_apply(x = out, NULL, ...,
 * partials, blocks) then normalises without the statistics launch.  16-bit types, C1 % 8 == C2 % 8 == 0. */
/* For the norms that are ONE launch (ga_group_norm_one_launch: a group's slab of <= 20 480 elements — the 16 x 16 and 8 x 8 levels)
 * the concatenation and the norm are one launch: y [B][HW][C1 + C2] = [silu](group_norm(cat([a, b]))) and cat [B][HW][C1 + C2]
 * itself (the ResnetBlock's shortcut GEMM and the backward read it); statistics as ga_group_norm_fwd leaves them.
 * GA_ERR_UNSUPPORTED for shapes whose norm takes two launches (ga_cat_channels_gn serves those). */
int ga_group_norm_one_launch(int HW, int C, int G, int dtype);
int ga_cat_group_norm_fwd(const void* a, const void* b, void* cat, const void* gamma, const void* beta, void* y, float* stats,
                          int B, int HW, int C1, int C2, int G, float eps, int act_silu, int dtype, ga_stream_t stream);
int ga_cat_channels_gn_blocks(int HW, int C, int G, int dtype);
int ga_cat_channels_gn(const void* a, const void* b, void* out, float* partials, int B, int HW, int C1, int C2, int G, int dtype,
                       ga_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * UNet host helper: 3x3 convolution, pad 1, stride 1 or 2, on channels-last activations as an implicit GEMM on MFMA
 * (diffusers 0.12.1 ResnetBlock2D / Upsample2D / Downsample2D convolutions, run by the reference inside
 * pipeline_guided_attention.py:583-743 through cuDNN).  16-bit types; Cin % 64 == 0 (one k-step = 64 channels of one
 * tap), Cout % 8 == 0; anything else returns GA_ERR_SHAPE.
 *   ga_conv3x3_pack_weights : W [Cout][Cin][3][3] with the given ELEMENT strides -> the blocked pack
 *                             Wp [9][ceil(N / 64)][C / 64][64][64] (tap, 64-row block, 64-channel chunk, row, channel;
 *                             rows past N are zero): what a workgroup fetches per k-step is one contiguous 8 KB run.
 *                             Wp holds ga_conv3x3_packed_elems(N, C) elements.  transpose_flip = 0: N = Cout, C = Cin
 *                             (forward).  transpose_flip = 1: N = Cin, C = Cout, taps mirrored — with it the same kernel
 *                             computes the backward to the input of a stride-1 convolution from the upstream gradient.
 *                             C % 64 == 0.  Once per weight version.
 *   ga_conv3x3_plan         : tile (bm x bn) and split-K factor for a shape, and the f32 workspace it needs
 *                             (ga_splitk_workspace_floats(B * Ho * Wo, Cout, bm, bn, splits); 0 when splits == 1).
 *                             Pure host function.
 *   ga_conv3x3_nhwc         : Y [B][Ho][Wo][Cout] = conv(X [B][H][W][Cin], Wp) (+ bias[Cout]) (+ residual like Y);
 *                             bias / residual may be NULL; Ho = (H - 1) / stride + 1.  splits > 1 runs the GEMM depth
 *                             in `splits` slices per tile: each slice stores its f32 accumulators to `workspace`, the slice
 *                             that arrives last sums them in slice order (bitwise reproducible) and writes Y — ONE
 *                             launch.  `tickets`: ceil(M / bm) * ceil(Cout / bn) 32-bit words, ZERO on entry; the kernel
 *                             leaves them zero.  Launches sharing workspace / tickets must be stream-ordered.
 *   ga_conv3x3_up2x_nhwc    : Y [B][2H][2W][Cout] = conv(nearest-neighbour 2x up-sampling of X [B][H][W][Cin], Wp), stride 1
 *                             (diffusers 0.12.1 Upsample2D.forward: F.interpolate(scale_factor=2, mode="nearest") + conv):
 *                             the patch gather reads pixel (y >> 1, x >> 1); the up-sampled map is never written.  Plan as
 *                             for a [B][2H][2W] input.  GA_ERR_SHAPE when the shape is not served by the patch kernel (the
 *                             caller up-samples itself and calls ga_conv3x3_nhwc).
 */
long long ga_splitk_workspace_floats(int64_t M, int N, int bm, int bn, int splits);
long long ga_conv3x3_packed_elems(int N, int C);
int ga_conv3x3_pack_weights(const void* W, void* Wp, int Cout, int Cin, int64_t stride_o, int64_t stride_i,
                            int64_t stride_y, int64_t stride_x, int transpose_flip, int dtype, ga_stream_t stream);
int ga_conv3x3_plan(int B, int H, int W, int Cin, int Cout, int stride, int* bm, int* bn, int* splits,
                    long long* workspace_floats);
int ga_conv3x3_nhwc(const void* X, const void* Wp, void* Y, float* workspace, unsigned* tickets, const void* bias,
                    const void* residual, int B, int H, int W, int Cin, int Cout, int stride, int bm, int bn, int splits,
                    int dtype, ga_stream_t stream);
/* The same convolution (stride 1) that ALSO leaves the GroupNorm statistics of its stored result for the norm layer that consumes
 * it: per (image, m tile, group) partial (sum, sum of squares) [B][blocks][groups][2] f32, blocks = ga_conv3x3_gn_blocks(...)
 * (0: shape not served — m tiles must not straddle images, 8 <= Cout / groups <= bn), written by the epilogue in a fixed order.
 * gn_chan_bias (optional [B][Cout] T): the term that norm adds to its input (the ResnetBlock's time embedding, diffusers
 * ResnetBlock2D: `hidden_states + temb` in front of norm2) — the sums are those of result + term; Y is unchanged.
 * ga_group_norm_apply(x = Y, ..., partials, blocks) then normalises without the statistics launch of ga_group_norm_fwd. */
int ga_conv3x3_gn_blocks(int H, int W, int Cout, int groups, int bm, int bn);
int ga_conv3x3_nhwc_gn(const void* X, const void* Wp, void* Y, float* workspace, unsigned* tickets, const void* bias,
                       const void* residual, int B, int H, int W, int Cin, int Cout, int bm, int bn, int splits, int dtype,
                       ga_stream_t stream, float* gn_partials, const void* gn_chan_bias, int gn_groups);

int ga_conv3x3_up2x_nhwc(const void* X, const void* Wp, void* Y, float* workspace, unsigned* tickets, const void* bias,
                         const void* residual, int B, int H, int W, int Cin, int Cout, int bm, int bn, int splits, int dtype,
                         ga_stream_t stream);

/* The UNet's two edge convolutions, 3x3, padding 1, stride 1, one side four channels wide (diffusers 0.12.1
 * UNet2DConditionModel.conv_in: latents -> block_out_channels[0], conv_out: the reverse; called through the UNet forward from
 * pipeline_guided_attention.py:647-738; the reference runs them through cuDNN).  Each is the other's adjoint, so the pair also
 * serves their backward-to-input passes (conv_in's carries the guidance gradient to the latents, pipeline…:_update_latent).
 * Not matrix-core work: one pass over the wide tensor bounds both (HBM).  16-bit types; W % 16 == 0.
 *   ga_conv3x3_thin_supported : 1 when (H, W, Cin, Cout) is served: Cin == 4 and Cout % 4 == 0 (64 ... 4096), or Cout == 4 and
 *                               Cin % 64 == 0 (64 ... 320).
 *   ga_conv3x3_thin_pack      : W [Cout][Cin][3][3] with the given ELEMENT strides -> pairs of input channels per 32-bit word,
 *                               [tap][input pair][output] (ga_conv3x3_thin_packed_elems(Cout, Cin) elements of T).
 *                               transpose_flip = 1: the adjoint's weights (outputs and inputs exchanged, taps mirrored), e.g.
 *                               conv_in's [320][4][3][3] packed for ga_conv3x3_thin_out computes conv_in's backward to its input.
 *   ga_conv3x3_thin_in        : X [B][4][H][W] (NCHW, dense: the latents as the pipeline holds them) -> Y [B][H][W][Cout] (+ bias)
 *   ga_conv3x3_thin_out       : X [B][H][W][Cin] -> Y [B][4][H][W] (NCHW, dense) (+ bias [4] or NULL)
 */
long long ga_conv3x3_thin_packed_elems(int Cout, int Cin);
int ga_conv3x3_thin_supported(int H, int W, int Cin, int Cout);
int ga_conv3x3_thin_pack(const void* W, void* Wp, int Cout, int Cin, long long stride_o, long long stride_i, long long stride_y,
                         long long stride_x, int transpose_flip, int dtype, ga_stream_t stream);
int ga_conv3x3_thin_in(const void* X, const void* Wp, const void* bias, void* Y, int B, int H, int W, int Cout, int dtype,
                       ga_stream_t stream);
int ga_conv3x3_thin_out(const void* X, const void* Wp, const void* bias, void* Y, int B, int H, int W, int Cin, int dtype,
                        ga_stream_t stream);

/* Linear layers / 1x1 convolutions of the UNet (diffusers 0.12.1 CrossAttention.to_q/to_k/to_v/to_out, FeedForward,
 * Transformer2DModel.proj_in/proj_out, ResnetBlock2D.conv_shortcut — called from pipeline_guided_attention.py:647-738
 * and utils/ptp_utils.py:70-91 through torch.nn.functional.linear / conv2d) on the convolution's pipelined MFMA kernel
 * as a one-tap convolution:  Y[M][N] = X[M][K] * W[N][K]^T (+ bias[N]) (+ residual[M][N]).
 * W is the framework's own [out_features][in_features] layout (no packing).  16-bit types; K % 64 == 0, N % 8 == 0;
 * bm x bn in {128x128, 128x64, 64x64}; splits > 1: workspace (ga_splitk_workspace_floats) and tickets as for ga_conv3x3_nhwc.
 */
int ga_gemm_nt(const void* X, const void* W, void* Y, float* workspace, unsigned* tickets, const void* bias,
               const void* residual, int64_t M, int K, int N, int bm, int bn, int splits, int dtype, ga_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Linear layers of the transformer blocks with their element-wise neighbours folded in
 * (diffusers 0.12.1 CrossAttention.to_q/to_k/to_v/to_out, Transformer2DModel.proj_in/proj_out, FeedForward:
 * utils/ptp_utils.py:70-91 and the blocks called from pipeline_guided_attention.py:647-738):
 *
 *     Y[m][n] = epilogue( sum_k X[m][k] * W[n][k] )      X [M][K] T (row stride ldx), W [N][K] T, Y row stride ldy
 *
 *   bias        [N] T or NULL            added before anything else (ignored with ln_partials: ln_shift carries it)
 *   residual    [M][ld_res] T or NULL    added to the result (the residual stream)
 *   geglu       1: W = [F h rows | F gate rows], Y [M][F] = (h + bias_h) * gelu(g + bias_g)  (diffusers GEGLU);
 *               preact (optional) [M][ld_pre] T receives the [M][2F] projection for the backward
 *   ln_partials [M][ln_parts][2] f32 or NULL: (sum, sum of squares) of X's rows in ln_parts pieces (what an earlier call
 *               left in row_partials_out).  The call then computes LayerNorm(X) W0^T + bias0 as
 *                   rstd[m] * (sum_k X[m][k] W[n][k] - mean[m] * ln_colsum[n]) + ln_shift[n]
 *               where the caller passes W = gamma o W0, ln_colsum[n] = sum_k W[n][k] (f32),
 *               ln_shift[n] = sum_k beta[k] W0[n][k] + bias0[n] (f32), eps = ln_eps.
 *   ln_stats_out     optional [M][2] f32 (mean, rstd) of the rows (what ga_add_layer_norm_bwd reads)
 *   row_partials_out optional [M][tn][2] f32, tn = ceil(N_out / bn): (sum, sum of squares) of the STORED result rows per
 *               column tile — the ln_partials of the next call (ln_parts = tn)
 * bm x bn in {64,128} x {64,128} is the tile (geglu: bn/2 result columns per tile); splits > 1 divides K over
 * workgroups: each slice stores its f32 accumulators to `slabs`, the slice that arrives last sums them in slice order
 * (bitwise reproducible) and runs the epilogue — one launch.  ga_linear_workspace gives the sizes: `slabs` f32
 * [slab_floats], `tickets` [tiles] 32-bit words that are ZERO on entry (the kernel leaves them zero); launches that
 * share them must be stream-ordered.  `stages` = k-steps of the LDS ring (0: the tile's default; 128x128: 2 or 3,
 * 128x64 / 64x128: 3 or 4, 64x64: 4 or 5): shallow rings fit two workgroups per CU.  K % 64 == 0, N % 8 == 0, 16-bit dtypes.
 * `stages` = GA_LINEAR_STREAM with the 128x128 tile: the PERSISTENT form for launches with many output tiles — one 512-thread
 * workgroup per CU streams its tiles through a 4-slot ring that stays full across tile boundaries, GEGLU formed in registers
 * from the f32 accumulators.  It serves the no-grad forms only: ln_partials required (2 <= ln_parts <= 20), no bias / residual /
 * preact / ln_stats_out / row_partials_out, splits = 1, K >= 320; anything else returns GA_ERR_UNSUPPORTED. */
#define GA_LINEAR_STREAM 8
typedef struct {
  const void* bias;
  const void* residual;
  int64_t ld_res;
  int geglu;
  void* preact;
  int64_t ld_pre;
  const float* ln_partials;
  int ln_parts;
  float ln_eps;
  const float* ln_colsum;
  const float* ln_shift;
  float* ln_stats_out;
  float* row_partials_out;
  /* GroupNorm statistics of the STORED result for the norm layer that consumes it (the transformer's proj_out + residual in
   * front of the next ResnetBlock's norm1 / conv_norm_out), as ga_conv3x3_nhwc_gn leaves them: the rows are gn_hw pixels per
   * image, gn_partials [M / gn_hw][ga_linear_gn_blocks(gn_hw, N, gn_groups, bm, bn)][gn_groups][2] f32.  NULL: none.
   * Not with geglu; gn_hw % bm == 0, 8 <= N / gn_groups <= bn. */
  float* gn_partials;
  int gn_groups;
  int gn_hw;
} ga_linear_epilogue_t;

int ga_linear_gn_blocks(int hw, int N, int groups, int bm, int bn);
int ga_linear_workspace(int64_t M, int N, int bm, int bn, int splits, int geglu, long long* slab_floats, int* tiles);
int ga_linear_fused(const void* X, int64_t ldx, const void* W, void* Y, int64_t ldy, const ga_linear_epilogue_t* ep,
                    float* slabs, unsigned* tickets, int64_t M, int K, int N, int bm, int bn, int splits, int stages,
                    int dtype, ga_stream_t stream);

/* Residual add + LayerNorm (diffusers 0.12.1 BasicTransformerBlock.forward: x = attn(norm(x)) + x; norm_next(x)):
 *   fwd: x_new = a + x (rounded to T), y = LayerNorm(x_new) * gamma + beta, stats [rows][2] f32 = (mean, rstd).
 *        a == NULL: plain LayerNorm of x (x_new is not written and may be NULL).  stats may be NULL (inference).
 *   bwd: dx = LayerNorm-backward(dy; x, stats, gamma) + g_res, where x is the row the forward normalised (x_new, or x
 *        in the plain mode) and g_res (optional) is the gradient arriving at x_new from its other consumer — the
 *        accumulation autograd would run as a separate add.  gamma / beta gradients are not produced (frozen UNet).
 * rows x C row-major, C a multiple of 16 / sizeof(T), C <= 4096 (f32: 2048); pointers 16-byte aligned.
 */
int ga_add_layer_norm_fwd(const void* a, const void* x, const void* gamma, const void* beta, void* x_new, void* y,
                          float* stats, int64_t rows, int C, float eps, int dtype, ga_stream_t stream);
int ga_add_layer_norm_bwd(const void* x, const float* stats, const void* gamma, const void* dy, const void* g_res,
                          void* dx, int64_t rows, int C, int dtype, ga_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GA_HIP_H */
