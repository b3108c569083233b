"""Linear layers of the transformer blocks on `ga_linear_fused` (csrc/linear.hip), with their neighbours folded in and
the autograd glue the guidance backward needs (reference: the diffusers 0.12.1 CrossAttention / BasicTransformerBlock /
Transformer2DModel forwards called from utils/ptp_utils.py:70-91 and pipeline_guided_attention.py:647-738).

    linear(x, W, b, residual, want_partials)         y = x W^T + b (+ residual); optionally the per-(row, column-tile) partial
                                                     sums of the STORED result — what a following LayerNorm fold consumes
    ln_linear(x, partials, norm, W, b, geglu)        y = LayerNorm(x) W^T + b, optionally through GEGLU: the GEMM runs on the
                                                     raw rows, the normalisation is applied in its epilogue (csrc/linear.hip)

Weights are frozen on this path (only the latents are differentiated, pipeline_guided_attention.py:466): the backward
produces dX only — dX = dY W by the library GEMM, then the existing LayerNorm / GEGLU backward kernels.
The gamma-scaled weights, their column sums and the beta / bias shifts are built once per (weight, norm) pair and
re-built when a version counter moves."""
import weakref

import torch
import torch.nn.functional as F

from . import ops
from ._lib import GaError


def supported(x, in_features, out_features):
    return (x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and in_features % 64 == 0 and out_features % 16 == 0
            and x.stride(-1) == 1)


def block_folds(x):
    """Whether a transformer block on the token tensor x (B, N, C) should take the folded form: served at all, and not one of
    the (tokens, channels) cases where the library form of the whole block was measured faster (ops.library_block)."""
    return supported(x, x.shape[-1], x.shape[-1]) and not ops.library_block(x.numel() // x.shape[-1], x.shape[-1])


def _transposed(weight):
    """W^T (in_features, out_features) contiguous, cached per weight version: dX = dY W is then the SAME NT kernel with W^T as
    its weight (the frozen UNet's Linear weights cost 0.7 GB twice — irrelevant beside 288 GB of HBM)."""
    key = (weight.data_ptr(), weight._version, tuple(weight.shape), weight.dtype)
    hit = _wt_cache.get(key)
    if hit is None or hit[0]() is None:
        with torch.no_grad():
            wt = weight.t().contiguous()
        if len(_wt_cache) > 2048:
            for dead in [k for k, (ref, _) in _wt_cache.items() if ref() is None]:
                del _wt_cache[dead]
        # the entry lives as long as the tensor that OWNS the storage: a 1x1 convolution's weight arrives here as a fresh
        # (out, in) view of the parameter on every call — tied to that view, the entry was dead by the next call and every
        # guidance backward transposed 27 weight matrices again (round 3: 14 strided-copy launches per backward pass)
        owner = weight._base if weight._base is not None else weight
        hit = _wt_cache[key] = (weakref.ref(owner), wt)
    return ops.keep_alive(hit[1])


_wt_cache = {}


def grad_input(gy, weight):
    """dX = dY W for a frozen Linear (weight (N, K)): ga_linear_fused on the cached transpose where the kernel serves the
    shape (N % 64 == 0: N is the depth of this product), the library GEMM otherwise."""
    gy2 = gy if gy.stride(-1) == 1 else gy.contiguous()
    if supported(gy2, weight.shape[0], weight.shape[1]) and weight.shape[1] % 8 == 0:
        return ops.linear_fused(gy2, _transposed(weight), None)["y"]
    return torch.matmul(gy, weight)


class _Linear(torch.autograd.Function):
    """y = x W^T + b (+ residual) [, row partial sums of y].  Differentiable w.r.t. x and the residual."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, want_partials, gn=None, box=None):
        """gn = (groups, hw) and a list `box`: the epilogue also takes the GroupNorm statistics of y for the norm that consumes it;
        what it took ((partials, blocks) or None) leaves through `box` (forward-only side data without a gradient)."""
        if ctx.needs_input_grad[1] or (bias is not None and ctx.needs_input_grad[2]):
            raise GaError("Linear weight gradients are not part of the guided-attention path (frozen UNet)")
        out = ops.linear_fused(x, weight, bias, residual=residual, want_row_partials=want_partials, gn=gn)
        if box is not None:
            box.append(out["gn"])
        ctx.weight, ctx.has_res = weight, residual is not None
        partials = out["row_partials"]
        if partials is None:
            partials = x.new_empty(0)
        ctx.mark_non_differentiable(partials)
        ctx.set_materialize_grads(False)     # no zero tensors for outputs nothing differentiates
        return out["y"], partials

    @staticmethod
    def backward(ctx, gy, _gp):
        if gy is None:
            return None, None, None, None, None, None, None
        gx = grad_input(gy, ctx.weight) if ctx.needs_input_grad[0] else None
        return gx, None, None, (gy if ctx.has_res else None), None, None, None


def linear(x, weight, bias=None, residual=None, want_partials=False, gn_for=None):
    """-> (y, partials or None).  gn_for = (groups, hw) of the GroupNorm (no channel bias) that consumes y viewed as (B, hw, N)
    images: where that norm would take two launches the epilogue takes its statistics and y carries them (`_ga_gn_tokens`:
    (partials, blocks, groups) — the caller re-attaches them to the NCHW view it hands to the norm)."""
    if gn_for is not None and ops.gn_two_launch(gn_for[1], weight.shape[0], gn_for[0], x.dtype):
        box = []
        y, partials = _Linear.apply(x, weight, bias, residual, want_partials, gn_for, box)
        if box and box[0] is not None:
            y._ga_gn_tokens = (box[0][0], box[0][1], gn_for[0])
        return y, (partials if want_partials else None)
    y, partials = _Linear.apply(x, weight, bias, residual, want_partials)
    return y, (partials if want_partials else None)


def _folded(weight, bias, norm):
    """(gamma o W rounded to the activation type, its f32 column sums, beta . W^T + bias in f32) of a Linear behind a
    LayerNorm — cached on the norm module per (weight storage, versions)."""
    key = (weight.data_ptr(), weight._version, tuple(weight.shape), None if bias is None else (bias.data_ptr(), bias._version),
           norm.weight._version, norm.bias._version, weight.dtype)
    cache = norm.__dict__.setdefault("_ga_folded", {})
    entry = cache.get(key)
    # good only for the weight object it was made from (a freed weight's address may be handed to another tensor)
    owner = weight._base if weight._base is not None else weight
    hit = entry[1] if entry is not None and entry[0]() is owner else None
    if hit is None:
        with torch.no_grad():
            wg = (weight.float() * norm.weight.float()[None, :]).to(weight.dtype).contiguous()
            colsum = wg.float().sum(1).contiguous()
            shift = weight.float() @ norm.bias.float()
            if bias is not None:
                shift = shift + bias.float()
            hit = (wg, colsum, shift.contiguous())
        if len(cache) > 8:
            # Entries of weights that are gone, and entries a newer version of the same weight has replaced.  Captured hipGraphs
            # read wg / colsum / shift by raw pointer — a runner keeps what it captured against alive itself
            # (ops.keepalive_scope), so dropping an entry here can never free memory a live graph replays on.
            def stale(k, ref):
                w = ref()
                return w is None or w._version != k[1] or norm.weight._version != k[4] or norm.bias._version != k[5]
            for k in [k for k, (ref, _) in cache.items() if stale(k, ref)]:
                del cache[k]
        cache[key] = (weakref.ref(owner), hit)
    ops.keep_alive(*hit)
    return hit


class _LNLinear(torch.autograd.Function):
    """(y, x) = ([GEGLU](LayerNorm(x) W^T + b), x) with the statistics taken from `partials`.  Differentiable w.r.t. x.
    The second output is x itself: the caller hands THAT to x's other consumer (the residual input of the projection that
    closes the sub-block), so that both gradients arrive here and the LayerNorm backward kernel adds them in its own pass
    (`g_res` of ga_add_layer_norm_bwd) — autograd's accumulation would be one more launch per sub-block."""

    @staticmethod
    def forward(ctx, x, partials, norm, weight, bias, geglu):
        need = ctx.needs_input_grad[0]
        wg, colsum, shift = _folded(weight, bias, norm)
        out = ops.linear_fused(x, wg, None, geglu=geglu, want_preact=geglu and need,
                               ln=(partials, colsum, shift, norm.eps), want_ln_stats=need)
        if need:
            ctx.save_for_backward(x, out["ln_stats"], out["preact"] if geglu else None)
            ctx.norm, ctx.weight, ctx.geglu = norm, weight, geglu
        ctx.set_materialize_grads(False)
        return out["y"], x.view_as(x)

    @staticmethod
    def backward(ctx, gy, g_pass):
        x, stats, preact = ctx.saved_tensors
        if gy is None:
            return g_pass, None, None, None, None, None
        if ctx.geglu:
            gy = ops.geglu_backward(preact, gy)
        g_ln = grad_input(gy, ctx.weight)                        # gradient at the LayerNorm's output (gamma applied inside)
        gx = ops._ln_bwd(x.contiguous(), stats, ctx.norm.weight, g_ln, g_pass)
        return gx, None, None, None, None, None


def ln_linear(x, partials, norm, weight, bias=None, geglu=False):
    """-> (y, x): use the returned x for the residual connection (see _LNLinear)."""
    return _LNLinear.apply(x, partials, norm, weight, bias, geglu)


def conv1x1_weight(conv):
    """(Cout, Cin) view of a 1x1 Conv2d weight (any memory format): free."""
    w = conv.weight
    return w.reshape(w.shape[0], w.shape[1])
