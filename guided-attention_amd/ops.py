"""Python-side operators over the C ABI of libga_hip.so: thin argument marshalling plus the
`torch.autograd.Function` glue that puts the HIP forward/backward kernels into the autograd graph
the pipeline differentiates (reference: pipeline_guided_attention.py:466 autograd.grad(loss, latents)).
"""
import ctypes
import weakref
from collections import OrderedDict

import torch

from . import _lib
from ._lib import GaError, check, dtype_code, load, require_cuda, stream_ptr


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


# Optional launch census (bench.py): which shapes each capture kernel was launched with, and how often.
# Per-launch HIP events are useless while the host is the bottleneck (the event pair brackets host gaps),
# so bench.py replays each recorded shape back-to-back between two events instead (replay_launch_us).
_CENSUS = None


def start_census():
    global _CENSUS
    _CENSUS = {}


def stop_census():
    global _CENSUS
    rec, _CENSUS = _CENSUS, None
    return rec or {}


def _count(key):
    if _CENSUS is not None:
        _CENSUS[key] = _CENSUS.get(key, 0) + 1


class census_scope:
    """Collect the launches made inside the block into `.launches` (used while a hipGraph is captured, so that
    each replay can be accounted with `add_census`)."""

    def __enter__(self):
        global _CENSUS
        self._outer, _CENSUS = _CENSUS, {}
        return self

    def __exit__(self, *exc):
        global _CENSUS
        self.launches, _CENSUS = _CENSUS, self._outer


def add_census(launches):
    if _CENSUS is not None:
        for key, n in launches.items():
            _CENSUS[key] = _CENSUS.get(key, 0) + n


class no_gc:
    """No automatic garbage collection inside the block (collect once on entry): a cycle collection that runs while a stream is
    capturing may destroy an older hipGraph / stream / event, which aborts the process — every capture of this package sits
    inside one."""

    def __enter__(self):
        import gc
        self.was = gc.isenabled()
        gc.collect()
        gc.disable()

    def __exit__(self, *exc):
        if self.was:
            import gc
            gc.enable()


_KEEP = None   # while a hipGraph is being captured: every cached tensor a captured launch reads by raw pointer


class keepalive_scope:
    """Collect into `.tensors` every derived weight tensor (folded LayerNorm weights, transposes, convolution packs) handed
    out inside the block.  A GraphRunner captures inside one and keeps the list: its graphs read those tensors by raw pointer,
    so they must live exactly as long as the runner does, whatever the caches that made them evict later."""

    def __enter__(self):
        global _KEEP
        self._outer, _KEEP = _KEEP, []
        return self

    def __exit__(self, *exc):
        global _KEEP
        self.tensors, _KEEP = _KEEP, self._outer


def keep_alive(*tensors):
    if _KEEP is not None:
        _KEEP.extend(t for t in tensors if t is not None)
    return tensors[0] if len(tensors) == 1 else tensors


_side_streams = {}


def side_stream(device=None):
    """ONE side stream per device for warm-up runs, captures and micro-replays: the split-K scratch is kept per (device,
    stream) (linear_workspace), so every fresh torch.cuda.Stream() would cost another 256 MB of slabs."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    s = _side_streams.get(idx)
    if s is None:
        s = _side_streams[idx] = torch.cuda.Stream(device=idx)
    return s


def replay_launch_us(key, iters=100):
    """Average duration (us) of one launch of a recorded kernel shape, inputs resident in HBM: `iters` launches
    captured into one hipGraph (no host in the loop: a Python-driven loop is launch-bound at ~10 us per call and
    hides the kernel time), replayed between two HIP events on the launch stream.  For the self-attention and
    GroupNorm backward the figure covers the launches of one backward call (2 resp. up to 3 kernels)."""
    kind, B, H, N, Kt, D, flag, dt = key
    dtype = {"torch.float16": torch.float16, "torch.bfloat16": torch.bfloat16, "torch.float32": torch.float32}[dt]
    dev = torch.device("cuda", torch.cuda.current_device())
    lib = load()
    fn = None
    if kind == "conv3x3":          # key = (kind, B, Cin, H*W, stride, Cout, epilogue?, dtype); the UNet's maps are square
        cin, hw, stride, cout = H, N, Kt, D
        side_len = int(round(hw ** 0.5))
        x = torch.randn(B, cin, side_len, hw // side_len, device=dev, dtype=dtype).contiguous(memory_format=torch.channels_last)
        # In the pipeline a convolution's weights are cold (1.7 GB of UNet weights cycle through a 256 MB Infinity Cache
        # between two uses) while its input was just written.  A back-to-back replay on ONE weight tensor would time an
        # L2-resident weight stream the product never sees: rotate over enough copies to exceed the Infinity Cache.
        n_copies = max(1, min(24, -(-320 * 2 ** 20 // (9 * cout * cin * 2))))
        wps = [torch.randn(9, cout, cin, device=dev, dtype=dtype) * (9 * cin) ** -0.5 for _ in range(n_copies)]
        turn = [0]
        ho, wo = (side_len - 1) // stride + 1, (hw // side_len - 1) // stride + 1
        bias = torch.randn(cout, device=dev, dtype=dtype) if flag else None
        res = (torch.randn(B, cout, ho, wo, device=dev, dtype=dtype).contiguous(memory_format=torch.channels_last)
               if flag else None)
        bm, bn, splits, _ = conv3x3_plan(B, side_len, hw // side_len, cin, cout, stride)
        y = torch.empty(B, cout, ho, wo, device=dev, dtype=dtype).contiguous(memory_format=torch.channels_last)
        code = dtype_code(x)

        def fn():
            wp = wps[turn[0] % n_copies]
            turn[0] += 1
            ws, tickets = splitk_workspace(dev, B * ho * wo, cout, bm, bn, splits)   # the scratch of the stream fn runs on
            check(lib.ga_conv3x3_nhwc(_ptr(x), _ptr(wp), _ptr(y), _ptr(ws), _ptr(tickets), _ptr(bias), _ptr(res), B, side_len,
                                      hw // side_len, cin, cout, stride, bm, bn, splits, code, stream_ptr()), "replay conv")
    elif kind in ("conv3x3_thin_in", "conv3x3_thin_out"):   # key = (kind, B, Cin, H*W, 1, Cout, bias?, dtype)
        cin, hw, cout = H, N, D
        side_len = int(round(hw ** 0.5))
        w = torch.randn(cout, cin, 3, 3, device=dev, dtype=dtype) * (9 * cin) ** -0.5
        wp = conv3x3_thin_packed_weights(w, False)
        bias = torch.randn(cout, device=dev, dtype=dtype) if flag else None
        x = torch.randn(B, cin, side_len, hw // side_len, device=dev, dtype=dtype)
        if cin != 4:
            x = x.contiguous(memory_format=torch.channels_last)

        def fn():
            conv3x3_thin(x, wp, cout, bias)
    elif kind in ("geglu_fwd", "geglu_bwd", "bias_residual_add", "add_layer_norm_fwd", "add_layer_norm_bwd"):
        rows, C = B, D
        code = dtype_code(torch.empty(0, dtype=dtype))
        t = lambda *shape: torch.randn(*shape, device=dev, dtype=dtype)  # noqa: E731
        if kind.startswith("geglu"):
            x, y, dy, dx = t(rows, 2 * C), t(rows, C), t(rows, C), t(rows, 2 * C)
            if kind == "geglu_fwd":
                def fn():
                    check(lib.ga_geglu_fwd(_ptr(x), _ptr(y), rows, C, code, stream_ptr()), "replay geglu fwd")
            else:
                def fn():
                    check(lib.ga_geglu_bwd(_ptr(x), _ptr(dy), _ptr(dx), rows, C, code, stream_ptr()), "replay geglu bwd")
        elif kind == "bias_residual_add":
            y, r, o, bias = t(rows, C), t(rows, C), t(rows, C), (t(C) if flag else None)

            def fn():
                check(lib.ga_bias_residual_add(_ptr(y), _ptr(bias), _ptr(r), _ptr(o), rows, C, code, stream_ptr()),
                      "replay bias residual")
        else:
            a, x, w, b_ = t(rows, C), t(rows, C), t(C), t(C)
            xn, y, dy, dx = t(rows, C), t(rows, C), t(rows, C), t(rows, C)
            stats = torch.empty(rows, 2, device=dev, dtype=torch.float32)
            check(lib.ga_add_layer_norm_fwd(_ptr(a), _ptr(x), _ptr(w), _ptr(b_), _ptr(xn), _ptr(y), _ptr(stats), rows, C,
                                            1e-5, code, stream_ptr()), "replay ln")
            if kind == "add_layer_norm_fwd":
                a_arg = a if flag else None

                def fn():
                    check(lib.ga_add_layer_norm_fwd(_ptr(a_arg), _ptr(x), _ptr(w), _ptr(b_), _ptr(xn), _ptr(y), _ptr(stats),
                                                    rows, C, 1e-5, code, stream_ptr()), "replay ln fwd")
            else:
                g_arg = a if flag else None

                def fn():
                    check(lib.ga_add_layer_norm_bwd(_ptr(xn), _ptr(stats), _ptr(w), _ptr(dy), _ptr(g_arg), _ptr(dx), rows,
                                                    C, code, stream_ptr()), "replay ln bwd")
    elif kind == "linear":        # key = (kind, M, K, 0, geglu | 2 ln | 4 residual, N, bias?, dtype)
        M, K, flags, N = B, H, Kt, D
        x = torch.randn(M, K, device=dev, dtype=dtype)
        n_copies = max(2, min(64, -(-320 * 2 ** 20 // (N * K * 2))))       # cold weights, as for the convolutions
        ws = [torch.randn(N, K, device=dev, dtype=dtype) * K ** -0.5 for _ in range(n_copies)]
        bias = torch.randn(N, device=dev, dtype=dtype) if flag else None
        geglu_, ln_, res_ = bool(flags & 1), bool(flags & 2), bool(flags & 4)
        n_out = N // 2 if geglu_ else N
        res = torch.randn(M, n_out, device=dev, dtype=dtype) if res_ else None
        ln = None
        if ln_:
            ln = (torch.rand(M, 5, 2, device=dev) * K, torch.randn(N, device=dev), torch.randn(N, device=dev), 1e-5)
        turn = [0]

        def fn():
            turn[0] += 1
            linear_fused(x, ws[turn[0] % n_copies], bias, residual=res, geglu=geglu_, ln=ln)
    elif kind in ("aggregate_maps", "aggregate_loss_fwd", "smooth_loss_fwd", "smooth_loss_bwd"):
        # B = guided tokens (aggregate_maps: tensors), H = head-maps in all, N = pixels, Kt = tokens of the context
        npix, res = N, int(round(N ** 0.5))
        nt = B if kind != "aggregate_maps" else 3
        entries = [{"index": 2 + i, "kind": "BOX", "geom": (.1 + .05 * i, .3, .4, .55), "subprompt": f"s{i}"}
                   for i in range(nt)]
        plan = LossPlan(entries, {"inside_loss_scale": .2, "outside_loss_scale": .2, "shrink_factor": .15})
        heads_all = max(H, 1)
        per = D if D and heads_all % D == 0 else (8 if heads_all % 8 == 0 else heads_all)   # heads per stored tensor, as counted
        maps = [torch.softmax(torch.randn(per, npix, Kt, device=dev), -1).to(dtype) for _ in range(heads_all // per)]
        A = torch.softmax(torch.randn(npix, Kt, device=dev), -1)
        if kind == "aggregate_maps":
            def fn():
                aggregate_maps(maps)
        elif kind == "aggregate_loss_fwd":
            def fn():
                aggregate_loss_fwd(maps, res, 1, Kt - 1, plan)
        elif kind == "smooth_loss_fwd":
            def fn():
                smooth_loss_fwd(A, res, 1, Kt - 1, plan)
        else:
            def fn():
                smooth_loss_bwd(A, res, 1, Kt - 1, plan, None, dtype if flag else None, 1.0 / 40)
    elif kind in ("latent_axpy", "latent_axpby", "cfg_ddim_step"):
        x, y, z = (torch.randn(N, device=dev, dtype=dtype) for _ in range(3))
        if kind == "latent_axpy":
            def fn():
                latent_axpy(x, y, 20.0, bool(flag))
        elif kind == "latent_axpby":
            def fn():
                latent_axpby(x, y, 0.9, 0.1)
        else:
            def fn():
                cfg_ddim_step(x, y, 7.5, z, 0.5, 0.6, bool(flag))
    elif kind.startswith("group_norm"):
        groups, HW, C = H, N, D
        side_len = int(round(HW ** 0.5))
        x = torch.randn(B, C, side_len, HW // side_len, device=dev, dtype=dtype).contiguous(memory_format=torch.channels_last)
        w, b_ = torch.ones(C, device=dev, dtype=dtype), torch.zeros(C, device=dev, dtype=dtype)
        y, dy = torch.empty_like(x), torch.randn_like(x)
        stats = torch.empty(B, groups, 2, device=dev, dtype=torch.float32)
        ws = torch.empty(B * 257 * groups * 2, device=dev, dtype=torch.float32)
        code = dtype_code(x)
        check(lib.ga_group_norm_fwd(_ptr(x), None, _ptr(w), _ptr(b_), _ptr(y), _ptr(stats), _ptr(ws), B, HW, C, groups,
                                    1e-5, int(flag), code, stream_ptr()), "replay gn")
        if kind == "group_norm_fwd":
            def fn():
                check(lib.ga_group_norm_fwd(_ptr(x), None, _ptr(w), _ptr(b_), _ptr(y), _ptr(stats), _ptr(ws), B, HW, C,
                                            groups, 1e-5, int(flag), code, stream_ptr()), "replay gn fwd")
        elif kind == "group_norm_apply":   # Kt = partial blocks per image (the producing convolution's epilogue left them)
            partials = torch.rand(B, Kt, groups, 2, device=dev) * (HW * C // groups) / Kt

            def fn():
                check(lib.ga_group_norm_apply(_ptr(x), None, _ptr(w), _ptr(b_), _ptr(y), _ptr(stats), _ptr(partials), Kt, B, HW,
                                              C, groups, 1e-5, int(flag), code, stream_ptr()), "replay gn apply")
        else:
            def fn():
                check(lib.ga_group_norm_bwd(_ptr(x), None, _ptr(dy), _ptr(w), _ptr(b_), _ptr(stats), None, _ptr(y), _ptr(ws), B,
                                            HW, C, groups, int(flag), code, stream_ptr()), "replay gn bwd")
    elif kind in ("attn_capture_fwd", "attn_capture_bwd", "self_attn_fwd", "self_attn_bwd"):
        q = torch.randn(B, N, H * D, device=dev, dtype=dtype)
        k = torch.randn(B, Kt, H * D, device=dev, dtype=dtype)
        v = torch.randn(B, Kt, H * D, device=dev, dtype=dtype)
        out = torch.empty_like(q)
        code, scale = dtype_code(q), D ** -0.5
    if kind == "attn_capture_fwd":
        probs = torch.empty(B * H, N, Kt, device=dev, dtype=dtype) if flag else None

        def fn():
            check(lib.ga_attn_capture_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(probs), B, H, N, Kt, D, scale,
                                          code, stream_ptr()), "replay fwd")
    elif kind == "attn_capture_bwd":
        d_o = torch.randn_like(q)
        dp = torch.randn(N, Kt, device=dev, dtype=dtype) * 1e-3 if flag else None

        def fn():
            check(lib.ga_attn_capture_bwd(_ptr(q), _ptr(k), _ptr(v), _ptr(d_o), _ptr(dp), 0, Kt, _ptr(out), None, None,
                                          B, H, N, Kt, D, scale, code, stream_ptr()), "replay bwd")
    elif kind in ("self_attn_fwd", "self_attn_bwd"):
        lse = torch.empty(B * H, N, device=dev, dtype=torch.float32)
        check(lib.ga_self_attn_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(lse), B, H, N, D, 0, scale, code,
                                   stream_ptr()), "replay sa")
        if kind == "self_attn_fwd":
            lse_arg = lse if flag else None

            def fn():
                check(lib.ga_self_attn_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(lse_arg), B, H, N, D, 0, scale,
                                           code, stream_ptr()), "replay sa fwd")
        else:
            d_o, delta = torch.randn_like(q), torch.empty_like(lse)
            dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)

            def fn():
                check(lib.ga_self_attn_bwd(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(d_o), _ptr(lse), _ptr(delta),
                                           _ptr(dq), _ptr(dk), _ptr(dv), B, H, N, D, 0, scale, code, stream_ptr()),
                      "replay sa bwd")
    if fn is None:
        raise GaError(f"no replay recipe for kernel kind {kind!r}")
    side = side_stream(dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with no_gc(), torch.cuda.graph(graph, stream=side):
            for _ in range(iters):
                fn()
        graph.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        side.synchronize()
        e0.record(side)
        graph.replay()
        e1.record(side)
        side.synchronize()
    torch.cuda.current_stream().wait_stream(side)
    return e0.elapsed_time(e1) * 1e3 / iters


# --------------------------------------------------------------------------------------- K1
def attn_capture_fwd(q, k, v, heads, scale, want_probs):
    """q (B,N,C), k/v (B,Kt,C) projections -> (o (B,N,C), probs (B*heads,N,Kt) or None)."""
    require_cuda(q, k, v)
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    B, N, C = q.shape
    Kt = k.shape[1]
    D = C // heads
    o = torch.empty_like(q)
    probs = torch.empty((B * heads, N, Kt), dtype=q.dtype, device=q.device) if want_probs else None
    _count(("attn_capture_fwd", B, heads, N, Kt, D, bool(want_probs), str(q.dtype)))
    check(load().ga_attn_capture_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(probs), B, heads, N, Kt, D,
                                     float(scale), dtype_code(q), stream_ptr()), "ga_attn_capture_fwd")
    return o, probs


def attn_capture_bwd(q, k, v, d_o, d_probs, heads, scale):
    """-> dq (B,N,C).  d_probs: None, a dense (B*heads,N,Kt) tensor, or an expanded (stride-0 over the
    head-map axis) view of one (N,Kt) map — passed to the kernel as strides, never materialised."""
    require_cuda(q, k, v, d_o, d_probs)
    B, N, C = q.shape
    Kt = k.shape[1]
    d_o = d_o.contiguous()
    sb = sn = 0
    if d_probs is not None:
        if d_probs.dtype != q.dtype:
            d_probs = d_probs.to(q.dtype)
        if d_probs.stride(2) != 1 or (d_probs.stride(0) != 0 and not d_probs.is_contiguous()):
            d_probs = d_probs.contiguous()
        sb, sn = d_probs.stride(0), d_probs.stride(1)
    dq = torch.empty_like(q)
    _count(("attn_capture_bwd", B, heads, N, Kt, C // heads, d_probs is not None, str(q.dtype)))
    check(load().ga_attn_capture_bwd(_ptr(q), _ptr(k), _ptr(v), _ptr(d_o), _ptr(d_probs), sb, sn, _ptr(dq), None, None,
                                     B, heads, N, Kt, C // heads, float(scale), dtype_code(q), stream_ptr()),
          "ga_attn_capture_bwd")
    return dq


class AttnCapture(torch.autograd.Function):
    """softmax(scale q k^T) v with the probabilities as a second, differentiable output."""

    @staticmethod
    def forward(ctx, q, k, v, heads, scale, want_probs):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        o, probs = attn_capture_fwd(q, k, v, heads, scale, want_probs)
        ctx.save_for_backward(q, k, v)
        ctx.heads, ctx.scale = heads, scale
        if probs is None:
            probs = q.new_empty(0)
            ctx.mark_non_differentiable(probs)
        return o, probs

    @staticmethod
    def backward(ctx, d_o, d_probs):
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            raise GaError("gradients w.r.t. the attention context (K/V) are not part of the guided-attention "
                          "path (only the latents are differentiated); freeze the UNet parameters")
        q, k, v = ctx.saved_tensors
        if d_probs is not None and d_probs.numel() == 0:
            d_probs = None
        if d_o is None:
            d_o = torch.zeros_like(q)
        dq = attn_capture_bwd(q, k, v, d_o, d_probs, ctx.heads, ctx.scale)
        return dq, None, None, None, None, None


# --------------------------------------------------------------------------------------- K1 + paint-with-words
def attn_scores_max(q, k, heads, scale):
    """-> (max over every scaled score of the call, f32 [1]; its flat index into [B*heads][N][Kt], int64 [1]), both
    on the device (no synchronisation): the `attention_scores.max()` of utils/ptp_utils.py:134."""
    require_cuda(q, k)
    q, k = q.contiguous(), k.contiguous()
    B, N, C = q.shape
    Kt = k.shape[1]
    packed = torch.zeros(1, dtype=torch.int64, device=q.device)
    check(load().ga_attn_scores_max(_ptr(q), _ptr(k), B, heads, N, Kt, C // heads, float(scale), dtype_code(q),
                                    _ptr(packed), stream_ptr()), "ga_attn_scores_max")
    hi = (packed >> 32) & 0xFFFFFFFF
    bits = torch.where((hi & 0x80000000) == 0, (~hi) & 0xFFFFFFFF, hi ^ 0x80000000)   # undo the order-preserving map
    value = (bits - ((bits >> 31) << 32)).to(torch.int32).view(torch.float32)
    return value, packed & 0xFFFFFFFF


def attn_capture_fwd_biased(q, k, v, heads, scale, want_probs, bias, coef):
    """ga_attn_capture_fwd with scores + bias[n][k] * coef[0] (bias (N, Kt) in q's dtype, coef f32 [1] on the device)."""
    require_cuda(q, k, v, bias, coef)
    q, k, v, bias = q.contiguous(), k.contiguous(), v.contiguous(), bias.to(q.dtype).contiguous()
    B, N, C = q.shape
    Kt = k.shape[1]
    if tuple(bias.shape) != (N, Kt) or coef.dtype != torch.float32:
        raise GaError(f"bias must be (N, Kt) = {(N, Kt)} and coef a float32 device scalar")
    o = torch.empty_like(q)
    probs = torch.empty((B * heads, N, Kt), dtype=q.dtype, device=q.device) if want_probs else None
    check(load().ga_attn_capture_fwd_biased(_ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(probs), _ptr(bias), _ptr(coef), B,
                                            heads, N, Kt, C // heads, float(scale), dtype_code(q), stream_ptr()),
          "ga_attn_capture_fwd_biased")
    return o, probs


def attn_capture_bwd_biased(q, k, v, d_o, d_probs, heads, scale, bias, coef):
    """-> (dq, d loss / d coef (f32 [1])) for the biased scores (see ga_hip.h)."""
    require_cuda(q, k, v, d_o, d_probs, bias, coef)
    B, N, C = q.shape
    Kt = k.shape[1]
    d_o, bias = d_o.contiguous(), bias.to(q.dtype).contiguous()
    sb = sn = 0
    if d_probs is not None:
        if d_probs.dtype != q.dtype:
            d_probs = d_probs.to(q.dtype)
        if d_probs.stride(2) != 1 or (d_probs.stride(0) != 0 and not d_probs.is_contiguous()):
            d_probs = d_probs.contiguous()
        sb, sn = d_probs.stride(0), d_probs.stride(1)
    dq = torch.empty_like(q)
    gsum = torch.zeros(1, dtype=torch.float32, device=q.device)
    check(load().ga_attn_capture_bwd_biased(_ptr(q), _ptr(k), _ptr(v), _ptr(d_o), _ptr(d_probs), sb, sn, _ptr(dq),
                                            _ptr(bias), _ptr(coef), _ptr(gsum), B, heads, N, Kt, C // heads, float(scale),
                                            dtype_code(q), stream_ptr()), "ga_attn_capture_bwd_biased")
    return dq, gsum


class AttnCapturePaintWithWords(torch.autograd.Function):
    """softmax(scale q k^T + mask * mult * max(scale q k^T)) v with the probabilities as a second output — the
    paint-with-words branch of the reference's get_attention_scores (utils/ptp_utils.py:113-138; mult =
    0.4 * log(1 + sigma_t)).  The maximum is taken over the whole call and carries a gradient, as in the reference."""

    @staticmethod
    def forward(ctx, q, k, v, heads, scale, want_probs, mask, mult):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        smax, arg = attn_scores_max(q, k, heads, scale)
        coef = smax * float(mult)
        o, probs = attn_capture_fwd_biased(q, k, v, heads, scale, want_probs, mask, coef)
        ctx.save_for_backward(q, k, v, mask, coef, arg)
        ctx.meta = (heads, scale, float(mult))
        if probs is None:
            probs = q.new_empty(0)
            ctx.mark_non_differentiable(probs)
        return o, probs

    @staticmethod
    def backward(ctx, d_o, d_probs):
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            raise GaError("gradients w.r.t. the attention context (K/V) are not part of the guided-attention path")
        q, k, v, mask, coef, arg = ctx.saved_tensors
        heads, scale, mult = ctx.meta
        if d_probs is not None and d_probs.numel() == 0:
            d_probs = None
        if d_o is None:
            d_o = torch.zeros_like(q)
        dq, gsum = attn_capture_bwd_biased(q, k, v, d_o, d_probs, heads, scale, mask, coef)
        # the gradient that reaches the scores through `.max()`: (mult * sum dS * mask) at the maximum's position
        B, N, C = q.shape
        Kt, D = k.shape[1], C // heads
        bh, rem = arg // (N * Kt), arg % (N * Kt)
        n, kk = rem // Kt, rem % Kt
        b, h = bh // heads, bh % heads
        krow = k.view(B * Kt * heads, D).index_select(0, (b * Kt + kk) * heads + h)           # (1, D)
        add = (krow.float() * (gsum * (mult * scale))).to(dq.dtype)
        dq.view(B * N * heads, D).index_add_(0, (b * N + n) * heads + h, add)
        return dq, None, None, None, None, None, None, None


# --------------------------------------------------------------------------------------- K2
def aggregate_maps(maps):
    """maps: list of (heads_i, npix, Kt) tensors of one dtype -> A (npix, Kt) f32."""
    require_cuda(*maps)
    maps = [m.contiguous() for m in maps]
    npix, Kt = maps[0].shape[1], maps[0].shape[2]
    n = len(maps)
    ptrs = (ctypes.c_void_p * n)(*[m.data_ptr() for m in maps])
    heads = (ctypes.c_int * n)(*[m.shape[0] for m in maps])
    A = torch.empty((npix, Kt), dtype=torch.float32, device=maps[0].device)
    _count(("aggregate_maps", n, sum(m.shape[0] for m in maps), npix, Kt, maps[0].shape[0], False, str(maps[0].dtype)))
    check(load().ga_aggregate_maps(ptrs, heads, n, npix, Kt, _ptr(A), dtype_code(maps[0]), stream_ptr()),
          "ga_aggregate_maps")
    return A


class AggregateMaps(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *maps):
        ctx.shapes = [m.shape for m in maps]
        ctx.dtype = maps[0].dtype
        return aggregate_maps(list(maps))

    @staticmethod
    def backward(ctx, dA):
        total = sum(s[0] for s in ctx.shapes)
        g = (dA * (1.0 / total)).to(ctx.dtype)
        # one (npix, Kt) map broadcast over every head-map: handed on as a stride-0 view
        return tuple(g.unsqueeze(0).expand(s) for s in ctx.shapes)


# --------------------------------------------------------------------------------------- K3+K4
class LossPlan:
    """Host-side descriptor of the guided tokens and hyper-parameters (what config.token_dict and
    shared_state.curHyperParams hold in the reference), marshalled once into the C structs."""

    # the hyper-parameters a plan reads (what its cache key must cover)
    HYPER_KEYS = ("strict", "inside_loss_scale", "outside_loss_scale", "bb_center_weight", "shrink_factor")

    def __init__(self, entries, hyper, smooth=True, sigma=0.5, kernel_size=3, sub_prompt_avg_within=False,
                 check_geometry=True):
        """entries: list of dict(index, kind 'BOX'|'COOR', geom, subprompt)."""
        self.entries = list(entries)
        self.check_geometry = check_geometry
        self._checked_res = set()
        self.shrink = float(hyper["shrink_factor"]) if "shrink_factor" in hyper else 0.0
        T = len(self.entries)
        self.T = T
        if T == 0:  # only custom (Python) losses are active: nothing for the fused kernel to do
            self.tokens, self.params, self.weights = None, None, []
            return
        counts = OrderedDict()
        for e in self.entries:
            counts[e["subprompt"]] = counts.get(e["subprompt"], 0) + 1
        self.weights = [1.0 / counts[e["subprompt"]] if sub_prompt_avg_within else 1.0 for e in self.entries]
        self.tokens = (_lib.ga_token_t * T)()
        for i, (e, w) in enumerate(zip(self.entries, self.weights)):
            tk = self.tokens[i]
            tk.token = int(e["index"])
            tk.kind = _lib.GA_TOK_BOX if e["kind"] == "BOX" else _lib.GA_TOK_COOR
            geom = list(e["geom"]) + [0.0] * (4 - len(e["geom"]))
            for j in range(4):
                tk.geom[j] = float(geom[j])
            tk.weight = w
        self.params = _lib.ga_loss_params_t()
        self.params.strict = 1 if hyper.get("strict", False) else 0
        self.params.inside_scale = hyper["inside_loss_scale"]
        self.params.outside_scale = hyper["outside_loss_scale"]
        self.params.center_weight = hyper.get("bb_center_weight", .05)
        self.params.sigma = sigma
        self.params.shrink = hyper["shrink_factor"]
        self.params.ksize = kernel_size
        self.params.smooth = 1 if smooth else 0
        self.T = T


def _check_boxes(plan, res):
    """The reference divides by the number of pixel centres inside the (shrunk) box (helpers.py:249,266:
    `at_most = 1.0 / num_inside`): a box that contains none raises ZeroDivisionError there, and so it does here
    (host-side, once per (plan, res))."""
    if not plan.check_geometry or res in plan._checked_res:
        return
    for e in plan.entries:
        if e["kind"] != "BOX":
            continue
        x, y, w, h = (float(v) * float(res) for v in e["geom"])
        ox, oy = plan.shrink * w, plan.shrink * h
        n_in = sum(1 for ii in range(res) for jj in range(res)
                   if x + ox <= jj + 0.5 <= x + w - ox and y + oy <= ii + 0.5 <= y + h - oy)
        if n_in == 0:
            raise ZeroDivisionError("float division by zero")
    plan._checked_res.add(res)


def smooth_loss_fwd(A, res, first, last, plan):
    require_cuda(A)
    if plan.T == 0:
        raise GaError("no guided tokens")
    _check_boxes(plan, res)
    if A.dtype != torch.float32:
        raise GaError("aggregated maps must be float32")
    A = A.contiguous()
    Kt = A.shape[-1]
    terms = torch.empty((plan.T, _lib.GA_TERMS), dtype=torch.float32, device=A.device)
    loss = torch.empty((1,), dtype=torch.float32, device=A.device)
    _count(("smooth_loss_fwd", plan.T, 0, res * res, Kt, 0, False, "torch.float32"))
    check(load().ga_smooth_loss_fwd(_ptr(A), res, Kt, first, last, plan.tokens, plan.T, ctypes.byref(plan.params),
                                    _ptr(terms), _ptr(loss), stream_ptr()), "ga_smooth_loss_fwd")
    return terms, loss


def smooth_loss_bwd(A, res, first, last, plan, dloss=None, bcast_dtype=None, bcast_scale=1.0):
    require_cuda(A, dloss)
    A = A.contiguous()
    Kt = A.shape[-1]
    dA = torch.empty_like(A)
    dPb = torch.empty(A.shape, dtype=bcast_dtype, device=A.device) if bcast_dtype is not None else None
    code = _lib.DTYPE_CODE[bcast_dtype] if bcast_dtype is not None else _lib.GA_F32
    if dloss is not None:
        dloss = dloss.to(torch.float32).contiguous()
    _count(("smooth_loss_bwd", plan.T, 0, res * res, Kt, 0, bcast_dtype is not None, str(bcast_dtype or torch.float32)))
    check(load().ga_smooth_loss_bwd(_ptr(A), res, Kt, first, last, plan.tokens, plan.T, ctypes.byref(plan.params),
                                    _ptr(dloss), _ptr(dA), _ptr(dPb), float(bcast_scale), code, stream_ptr()),
          "ga_smooth_loss_bwd")
    return dA, dPb


class SmoothLoss(torch.autograd.Function):
    """A (res*res, Kt) f32 -> (terms (T,8), loss (1,)).  Only `loss` is differentiable."""

    @staticmethod
    def forward(ctx, A, res, first, last, plan):
        terms, loss = smooth_loss_fwd(A, res, first, last, plan)
        ctx.save_for_backward(A)
        ctx.args = (res, first, last, plan)
        ctx.mark_non_differentiable(terms)
        return terms, loss

    @staticmethod
    def backward(ctx, _dterms, dloss):
        (A,) = ctx.saved_tensors
        res, first, last, plan = ctx.args
        dA, _ = smooth_loss_bwd(A, res, first, last, plan, dloss)
        return dA, None, None, None, None


_tickets = {}   # (device index, stream) -> one zeroed 32-bit word (the arrival counter of the fused aggregate + loss launch)


def _stream_key(device):
    """Scratch that kernels hand data through (split-K slabs, arrival tickets) is only safe under stream order: one set per
    (device, stream the launch goes to).  Launches captured into a hipGraph keep the set of their capture stream."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    return (idx, torch.cuda.current_stream(idx).cuda_stream)


def _ticket(device):
    key = _stream_key(device)
    t = _tickets.get(key)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            raise GaError("the loss launch's ticket word must exist before a hipGraph capture (call ops.prepare_device first)")
        t = _tickets[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return t


def aggregate_loss_fwd(maps, res, first, last, plan):
    """K2 + K3 + K4 in one launch: maps (list of (heads_i, res*res, Kt) tensors of one dtype) ->
    (A (res*res, Kt) f32, terms (T, 8), loss (1,))."""
    require_cuda(*maps)
    if plan.T == 0:
        raise GaError("no guided tokens")
    _check_boxes(plan, res)
    maps = [m.contiguous() for m in maps]
    npix, Kt = maps[0].shape[1], maps[0].shape[2]
    if npix != res * res:
        raise GaError(f"maps have {npix} pixels, expected {res * res}")
    n = len(maps)
    ptrs = (ctypes.c_void_p * n)(*[m.data_ptr() for m in maps])
    heads = (ctypes.c_int * n)(*[m.shape[0] for m in maps])
    dev = maps[0].device
    A = torch.empty((npix, Kt), dtype=torch.float32, device=dev)
    terms = torch.empty((plan.T, _lib.GA_TERMS), dtype=torch.float32, device=dev)
    loss = torch.empty((1,), dtype=torch.float32, device=dev)
    _count(("aggregate_loss_fwd", plan.T, sum(m.shape[0] for m in maps), npix, Kt, maps[0].shape[0], False, str(maps[0].dtype)))
    ticket = _ticket(dev)
    _check_ticketed(load().ga_aggregate_loss_fwd(ptrs, heads, n, res, Kt, first, last, plan.tokens, plan.T,
                                                 ctypes.byref(plan.params), _ptr(A), _ptr(terms), _ptr(loss), _ptr(ticket),
                                                 dtype_code(maps[0]), stream_ptr()), "ga_aggregate_loss_fwd", ticket)
    return A, terms, loss


class AggregateSmoothLoss(torch.autograd.Function):
    """(res, first, last, plan, *maps) -> (A (res*res, Kt) f32 [not differentiable here], terms (T, 8), loss (1,)): the
    aggregate and the smoothed box loss as ONE launch; the backward is one launch too (ga_smooth_loss_bwd emits the
    dtype-cast dLoss/dA / n_head_maps map that the capture kernels broadcast over the head-maps)."""

    @staticmethod
    def forward(ctx, res, first, last, plan, *maps):
        A, terms, loss = aggregate_loss_fwd(list(maps), res, first, last, plan)
        ctx.save_for_backward(A)
        ctx.args = (res, first, last, plan, [m.shape for m in maps], maps[0].dtype)
        ctx.mark_non_differentiable(A, terms)
        ctx.set_materialize_grads(False)
        return A, terms, loss

    @staticmethod
    def backward(ctx, _dA, _dterms, dloss):
        (A,) = ctx.saved_tensors
        res, first, last, plan, shapes, dtype = ctx.args
        if dloss is None:
            return (None,) * (4 + len(shapes))
        total = sum(s[0] for s in shapes)
        _, g = smooth_loss_bwd(A, res, first, last, plan, dloss, bcast_dtype=dtype, bcast_scale=1.0 / total)
        return (None, None, None, None) + tuple(g.unsqueeze(0).expand(s) for s in shapes)


def gaussian_weights(kernel_size, sigma):
    w = (ctypes.c_float * (kernel_size * kernel_size))()
    check(load().ga_gaussian_weights(kernel_size, float(sigma), w), "ga_gaussian_weights")
    return torch.tensor(list(w), dtype=torch.float32).reshape(kernel_size, kernel_size)


# --------------------------------------------------------------------------------------- K5 / K6 / DDIM
def latent_axpy(latents, grad, step, want_absmean=False):
    require_cuda(latents, grad)
    latents, grad = latents.contiguous(), grad.contiguous().to(latents.dtype)
    out = torch.empty_like(latents)
    absmean = torch.empty((1,), dtype=torch.float32, device=latents.device) if want_absmean else None
    _count(("latent_axpy", 1, 0, latents.numel(), 0, 0, bool(want_absmean), str(latents.dtype)))
    check(load().ga_latent_axpy(_ptr(latents), _ptr(grad), float(step), _ptr(out), _ptr(absmean), latents.numel(),
                                dtype_code(latents), stream_ptr()), "ga_latent_axpy")
    return out, absmean


def latent_axpby(x, y, a, b):
    require_cuda(x, y)
    x, y = x.contiguous(), y.contiguous().to(x.dtype)
    out = torch.empty_like(x)
    _count(("latent_axpby", 1, 0, x.numel(), 0, 0, False, str(x.dtype)))
    check(load().ga_latent_axpby(_ptr(x), _ptr(y), float(a), float(b), _ptr(out), x.numel(), dtype_code(x),
                                 stream_ptr()), "ga_latent_axpby")
    return out


def cfg_ddim_step(eps_uncond, eps_text, guidance, x, alpha_t, alpha_prev, want_x0=False):
    require_cuda(eps_uncond, eps_text, x)
    eps_uncond, eps_text, x = eps_uncond.contiguous(), eps_text.contiguous(), x.contiguous()
    prev = torch.empty_like(x)
    x0 = torch.empty_like(x) if want_x0 else None
    _count(("cfg_ddim_step", 1, 0, x.numel(), 0, 0, bool(want_x0), str(x.dtype)))
    check(load().ga_cfg_ddim_step(_ptr(eps_uncond), _ptr(eps_text), float(guidance), _ptr(x), float(alpha_t),
                                  float(alpha_prev), _ptr(prev), _ptr(x0), x.numel(), dtype_code(x), stream_ptr()),
          "ga_cfg_ddim_step")
    return prev, x0


# --------------------------------------------------------------------------------------- GroupNorm (+SiLU), NHWC
def _nhwc(x):
    return x if x.is_contiguous(memory_format=torch.channels_last) else x.contiguous(memory_format=torch.channels_last)


class GroupNormAct(torch.autograd.Function):
    """y = [silu](group_norm(x [+ chan_bias[:, :, None, None]])) on channels-last (B, C, H, W) tensors;
    differentiable w.r.t. x only (chan_bias is the time-embedding term: it carries no gradient on this path).
    with_alias: returns (y, x) — the caller hands that second output to x's OTHER consumer (the block's skip connection),
    so that both gradients arrive in this node and ga_group_norm_bwd adds them in its own pass (`g_res`); autograd's
    accumulation would be one more launch per block (23 per guidance backward of the SD-1.x UNet)."""

    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps, act, chan_bias, with_alias=False, produced=None):
        """produced = (partials (B, blocks, groups, 2) f32, blocks) from the epilogue of the convolution that made x
        (conv3x3_nhwc(..., gn=...)): the large-level forward is then ONE launch (ga_group_norm_apply) instead of two."""
        require_cuda(x, weight, bias, chan_bias)
        if x.dim() != 4:
            raise GaError("GroupNormAct expects a (B, C, H, W) tensor")
        x = _nhwc(x)
        B, C, H, W = x.shape
        if chan_bias is not None:
            if chan_bias.requires_grad:
                raise GaError("the channel bias of the fused GroupNorm carries no gradient on this path")
            chan_bias = chan_bias.to(x.dtype).contiguous()
            if tuple(chan_bias.shape) != (B, C):
                raise GaError(f"chan_bias must be (B, C) = {(B, C)}, got {tuple(chan_bias.shape)}")
        if produced is not None and produced[0] == "done":     # the producing concatenation's launch ran this norm too
            y, stats = produced[1], produced[2]
            ctx.save_for_backward(x, weight, bias, stats, chan_bias)
            ctx.meta = (groups, bool(act))
            if with_alias:
                ctx.set_materialize_grads(False)
                return y, x.view_as(x)
            return y
        y = torch.empty_like(x, memory_format=torch.channels_last)
        stats = torch.empty((B, groups, 2), dtype=torch.float32, device=x.device)
        if produced is not None:
            partials, blocks = produced
            _count(("group_norm_apply", B, groups, H * W, blocks, C, bool(act), str(x.dtype)))
            check(load().ga_group_norm_apply(_ptr(x), _ptr(chan_bias), _ptr(weight), _ptr(bias), _ptr(y), _ptr(stats),
                                             _ptr(partials), blocks, B, H * W, C, groups, float(eps), int(bool(act)),
                                             dtype_code(x), stream_ptr()), "ga_group_norm_apply")
        else:
            ws = torch.empty((B * 257 * groups * 2,), dtype=torch.float32, device=x.device)
            _count(("group_norm_fwd", B, groups, H * W, 0, C, bool(act), str(x.dtype)))
            check(load().ga_group_norm_fwd(_ptr(x), _ptr(chan_bias), _ptr(weight), _ptr(bias), _ptr(y), _ptr(stats), _ptr(ws),
                                           B, H * W, C, groups, float(eps), int(bool(act)), dtype_code(x), stream_ptr()),
                  "ga_group_norm_fwd")
        ctx.save_for_backward(x, weight, bias, stats, chan_bias)
        ctx.meta = (groups, bool(act))
        if with_alias:
            ctx.set_materialize_grads(False)
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, g_alias=None):
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            raise GaError("GroupNorm weight gradients are not part of the guided-attention path (frozen UNet)")
        if dy is None:           # only the alias was differentiated
            return g_alias, None, None, None, None, None, None, None, None
        x, weight, bias, stats, chan_bias = ctx.saved_tensors
        groups, act = ctx.meta
        B, C, H, W = x.shape
        dy = _nhwc(dy)
        if g_alias is not None:
            g_alias = _nhwc(g_alias)
        dx = torch.empty_like(x, memory_format=torch.channels_last)
        ws = torch.empty((B * 257 * groups * 2,), dtype=torch.float32, device=x.device)
        _count(("group_norm_bwd", B, groups, H * W, 0, C, bool(act), str(x.dtype)))
        check(load().ga_group_norm_bwd(_ptr(x), _ptr(chan_bias), _ptr(dy), _ptr(weight), _ptr(bias), _ptr(stats),
                                       _ptr(g_alias), _ptr(dx), _ptr(ws), B, H * W, C, groups, int(act), dtype_code(x),
                                       stream_ptr()),
              "ga_group_norm_bwd")
        return dx, None, None, None, None, None, None, None, None


def group_norm_act(x, weight, bias, groups, eps, act, chan_bias=None, with_alias=False):
    """-> y, or (y, x) with with_alias (use that x for the skip connection: see GroupNormAct).
    When x came out of conv3x3(..., gn_for=(groups, chan_bias)) — the convolution's epilogue took the statistics this norm
    needs (same group count, the SAME channel-bias tensor) — the statistics launch is skipped."""
    pre = getattr(x, "_ga_gn", None)
    produced = None
    if pre is not None and pre["groups"] == groups and pre["chan_bias"] is chan_bias and pre["shape"] == tuple(x.shape):
        if "done" not in pre:
            produced = (pre["partials"], pre["blocks"])
        elif pre["weight"] is weight and pre["bias"] is bias and pre["eps"] == float(eps) and pre["act"] == bool(act):
            produced = ("done",) + pre.pop("done")     # handed out once: a second norm of the same tensor launches its own
    return GroupNormAct.apply(x, weight, bias, groups, eps, act, chan_bias, with_alias, produced)


def gn_fused_with_cat(HW, C, groups, dtype):
    """True when ga_group_norm_fwd is ONE launch for the shape: a concatenation in front of it joins that launch."""
    code = _lib.DTYPE_CODE.get(dtype)
    return code is not None and (C // groups) % 2 == 0 and bool(load().ga_group_norm_one_launch(HW, C, groups, code))


def gn_two_launch(HW, C, groups, dtype):
    """True when ga_group_norm_fwd takes two launches for the shape (a producer's partial sums then save one)."""
    code = _lib.DTYPE_CODE.get(dtype)
    return code is not None and bool(load().ga_group_norm_two_launch(HW, C, groups, code))


# --------------------------------------------------------------------------------------- feed-forward / residual epilogues
class Geglu(torch.autograd.Function):
    """y = x[..., :F] * gelu(x[..., F:]) on the GEGLU projection's output (diffusers 0.12.1 GEGLU.forward)."""

    @staticmethod
    def forward(ctx, x):
        require_cuda(x)
        x = x.contiguous()
        F2 = x.shape[-1]
        rows = x.numel() // F2
        y = torch.empty(x.shape[:-1] + (F2 // 2,), dtype=x.dtype, device=x.device)
        _count(("geglu_fwd", rows, 0, 0, 0, F2 // 2, False, str(x.dtype)))
        check(load().ga_geglu_fwd(_ptr(x), _ptr(y), rows, F2 // 2, dtype_code(x), stream_ptr()), "ga_geglu_fwd")
        if ctx.needs_input_grad[0]:
            ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        F2 = x.shape[-1]
        dx = torch.empty_like(x)
        _count(("geglu_bwd", x.numel() // F2, 0, 0, 0, F2 // 2, False, str(x.dtype)))
        check(load().ga_geglu_bwd(_ptr(x), _ptr(dy), _ptr(dx), x.numel() // F2, F2 // 2, dtype_code(x), stream_ptr()),
              "ga_geglu_bwd")
        return dx


def geglu(x):
    return Geglu.apply(x)


def geglu_backward(x, dy):
    """d(h * gelu(gate)) w.r.t. the projection x = [h | gate] (..., 2F) given dy (..., F)."""
    require_cuda(x, dy)
    x, dy = x.contiguous(), dy.contiguous()
    F2 = x.shape[-1]
    dx = torch.empty_like(x)
    _count(("geglu_bwd", x.numel() // F2, 0, 0, 0, F2 // 2, False, str(x.dtype)))
    check(load().ga_geglu_bwd(_ptr(x), _ptr(dy), _ptr(dx), x.numel() // F2, F2 // 2, dtype_code(x), stream_ptr()),
          "ga_geglu_bwd")
    return dx


def cat_channels_supported(a, b):
    """Both channels-last-dense 4-D tensors of one 16- or 32-bit dtype on the GPU, same (B, H, W), channel counts that are
    whole 16-byte vectors, and a vector count the kernel's 32-bit row division serves."""
    if not (a.is_cuda and b.is_cuda and a.dim() == 4 and b.dim() == 4 and a.dtype == b.dtype):
        return False
    if a.dtype not in (torch.float16, torch.bfloat16, torch.float32):
        return False
    if a.shape[0] != b.shape[0] or a.shape[2:] != b.shape[2:]:
        return False
    per = 16 // a.element_size()
    if a.shape[1] % per or b.shape[1] % per:
        return False
    rows = a.shape[0] * a.shape[2] * a.shape[3]
    # Measured domain: maps of <= 64 x 64 pixels and results of <= 24 MB (every concatenation of the 512^2 SD-1.x passes:
    # -0.8 ... -0.9 % per forward pass, profiles/r3_ab_cat_channels.txt).  On the 768^2 configuration (96 x 96 maps, results up
    # to 53 MB) the library's copy was the faster one (0.917 against 0.887 images/s with every concatenation on this kernel).
    if a.shape[2] * a.shape[3] > 4096 or rows * (a.shape[1] + b.shape[1]) * a.element_size() > 24 * 1024 * 1024:
        return False
    return all(rows * (t.shape[1] // per) ** 2 < (1 << 32) for t in (a, b))   # the kernel's 32-bit multiply-high row division


class CatChannels(torch.autograd.Function):
    """torch.cat([a, b], dim=1) on channels-last activations as one ga_cat_channels launch (the UpBlock's concatenation of
    the running activation with the skip connection: diffusers 0.12.1 UpBlock2D / CrossAttnUpBlock2D, called from the
    reference's UNet forward pipeline_guided_attention.py:583-743).  Backward: the two channel slices of the gradient, as
    views — what torch.cat's own backward returns."""

    @staticmethod
    def forward(ctx, a, b):
        require_cuda(a, b)
        a, b = _nhwc(a), _nhwc(b)
        B, C1, H, W = a.shape
        C2 = b.shape[1]
        out = torch.empty((B, C1 + C2, H, W), dtype=a.dtype, device=a.device, memory_format=torch.channels_last)
        check(load().ga_cat_channels(_ptr(a), _ptr(b), _ptr(out), B * H * W, C1, C2, a.element_size(), stream_ptr()),
              "ga_cat_channels")
        ctx.c1 = C1
        return out

    @staticmethod
    def backward(ctx, g):
        return g[:, :ctx.c1], g[:, ctx.c1:]


class _CatChannelsGn(CatChannels):
    """CatChannels whose launch also takes the consuming GroupNorm's statistics (ga_cat_channels_gn); what it took leaves through
    `box` (forward-only side data without a gradient, as for _Conv3x3Gn)."""

    @staticmethod
    def forward(ctx, a, b, groups, box):
        require_cuda(a, b)
        a, b = _nhwc(a), _nhwc(b)
        B, C1, H, W = a.shape
        C2 = b.shape[1]
        blocks = int(load().ga_cat_channels_gn_blocks(H * W, C1 + C2, groups, dtype_code(a)))
        out = torch.empty((B, C1 + C2, H, W), dtype=a.dtype, device=a.device, memory_format=torch.channels_last)
        partials = torch.empty((B, blocks, groups, 2), dtype=torch.float32, device=a.device)
        check(load().ga_cat_channels_gn(_ptr(a), _ptr(b), _ptr(out), _ptr(partials), B, H * W, C1, C2, groups, dtype_code(a),
                                        stream_ptr()), "ga_cat_channels_gn")
        ctx.c1 = C1
        box.append((partials, blocks))
        return out

    @staticmethod
    def backward(ctx, g):
        return g[:, :ctx.c1], g[:, ctx.c1:], None, None


class _CatGroupNorm(CatChannels):
    """CatChannels and the GroupNorm(+SiLU) that consumes it as ONE launch (ga_cat_group_norm_fwd: the norms that are a single
    launch anyway).  The Function's output is the concatenation; the norm's output and statistics leave through `box` and
    GroupNormAct picks them up instead of launching (it stays the autograd node of the norm)."""

    @staticmethod
    def forward(ctx, a, b, weight, bias, groups, eps, act, box):
        require_cuda(a, b, weight, bias)
        a, b = _nhwc(a), _nhwc(b)
        B, C1, H, W = a.shape
        C2 = b.shape[1]
        out = torch.empty((B, C1 + C2, H, W), dtype=a.dtype, device=a.device, memory_format=torch.channels_last)
        y = torch.empty_like(out, memory_format=torch.channels_last)
        stats = torch.empty((B, groups, 2), dtype=torch.float32, device=a.device)
        _count(("group_norm_fwd", B, groups, H * W, 0, C1 + C2, bool(act), str(a.dtype)))
        check(load().ga_cat_group_norm_fwd(_ptr(a), _ptr(b), _ptr(out), _ptr(weight), _ptr(bias), _ptr(y), _ptr(stats), B, H * W,
                                           C1, C2, groups, float(eps), int(bool(act)), dtype_code(a), stream_ptr()),
              "ga_cat_group_norm_fwd")
        ctx.c1 = C1
        box.append((y, stats))
        return out

    @staticmethod
    def backward(ctx, g):
        return g[:, :ctx.c1], g[:, ctx.c1:], None, None, None, None, None, None


def cat_channels(a, b, gn_for=None, norm=None):
    """torch.cat([a, b], dim=1); the HIP launch where cat_channels_supported, the library otherwise (CPU oracle, odd widths).
    gn_for = the group count of the GroupNorm (no channel bias) that consumes the result: where that norm would take two launches
    this launch takes its statistics too, and the result carries them (`_ga_gn`, read by group_norm_act).
    norm = (weight, bias, eps, act) of that norm: where it is a single launch anyway, concatenation and norm are ONE launch, and
    the result carries the norm's output (`_ga_gn["done"]`) for group_norm_act to hand out."""
    if cat_channels_supported(a, b) and gn_for is not None and norm is not None and (a.shape[1] + b.shape[1]) % gn_for == 0 and \
            a.dtype in (torch.float16, torch.bfloat16) and norm[0].dtype == a.dtype and a.shape[1] % 2 == 0 and \
            gn_fused_with_cat(a.shape[2] * a.shape[3], a.shape[1] + b.shape[1], gn_for, a.dtype):
        weight, bias, eps, act = norm
        box = []
        y = _CatGroupNorm.apply(a, b, weight, bias, gn_for, eps, act, box)
        out, stats = box[0]
        y._ga_gn = {"done": (out, stats), "groups": gn_for, "chan_bias": None, "shape": tuple(y.shape), "weight": weight,
                    "bias": bias, "eps": float(eps), "act": bool(act)}
        return y
    if cat_channels_supported(a, b):
        if gn_for is not None and a.dtype in (torch.float16, torch.bfloat16) and (a.shape[1] + b.shape[1]) % gn_for == 0 and \
                gn_two_launch(a.shape[2] * a.shape[3], a.shape[1] + b.shape[1], gn_for, a.dtype) and \
                load().ga_cat_channels_gn_blocks(a.shape[2] * a.shape[3], a.shape[1] + b.shape[1], gn_for, dtype_code(a)):
            box = []
            y = _CatChannelsGn.apply(a, b, gn_for, box)
            partials, blocks = box[0]
            y._ga_gn = {"partials": partials, "blocks": blocks, "groups": gn_for, "chan_bias": None, "shape": tuple(y.shape)}
            return y
        return CatChannels.apply(a, b)
    return torch.cat([a, b], dim=1)


class BiasResidualAdd(torch.autograd.Function):
    """out = y + bias[c] + residual on channels-last activations (ResnetBlock2D: conv2's bias and the skip
    connection in one pass).  bias receives no gradient (frozen UNet)."""

    @staticmethod
    def forward(ctx, y, bias, residual):
        require_cuda(y, residual)
        if ctx.needs_input_grad[1]:
            raise GaError("bias gradients are not part of the guided-attention path (frozen UNet)")
        y, residual = _nhwc(y), _nhwc(residual)
        B, C, H, W = y.shape
        out = torch.empty_like(y, memory_format=torch.channels_last)
        _count(("bias_residual_add", B * H * W, 0, 0, 0, C, bias is not None, str(y.dtype)))
        check(load().ga_bias_residual_add(_ptr(y), _ptr(bias), _ptr(residual), _ptr(out), B * H * W, C, dtype_code(y),
                                          stream_ptr()), "ga_bias_residual_add")
        return out

    @staticmethod
    def backward(ctx, g):
        return g, None, g


def bias_residual_add(y, bias, residual):
    return BiasResidualAdd.apply(y, bias, residual)


def _ln_fwd(a, x, weight, bias, eps, need_stats):
    require_cuda(x, weight, bias)
    x = x.contiguous()
    C = x.shape[-1]
    rows = x.numel() // C
    y = torch.empty_like(x)
    stats = torch.empty((rows, 2), dtype=torch.float32, device=x.device) if need_stats else None
    xnew = None
    if a is not None:
        a = a.contiguous()
        xnew = torch.empty_like(x)
    _count(("add_layer_norm_fwd", rows, 0, 0, 0, C, a is not None, str(x.dtype)))
    check(load().ga_add_layer_norm_fwd(_ptr(a), _ptr(x), _ptr(weight), _ptr(bias), _ptr(xnew), _ptr(y), _ptr(stats), rows,
                                       C, float(eps), dtype_code(x), stream_ptr()), "ga_add_layer_norm_fwd")
    return (x if a is None else xnew), y, stats


def _ln_bwd(row, stats, weight, g_y, g_res):
    C = row.shape[-1]
    g_y = g_y.contiguous()
    g_res = g_res.contiguous() if g_res is not None else None
    d = torch.empty_like(row)
    _count(("add_layer_norm_bwd", row.numel() // C, 0, 0, 0, C, g_res is not None, str(row.dtype)))
    check(load().ga_add_layer_norm_bwd(_ptr(row), _ptr(stats), _ptr(weight), _ptr(g_y), _ptr(g_res), _ptr(d),
                                       row.numel() // C, C, dtype_code(row), stream_ptr()), "ga_add_layer_norm_bwd")
    return d


def _frozen(ctx, first):
    if ctx.needs_input_grad[first] or ctx.needs_input_grad[first + 1]:
        raise GaError("LayerNorm weight gradients are not part of the guided-attention path (frozen UNet)")


class LayerNorm(torch.autograd.Function):
    """y = LayerNorm(x) over the last dimension (one wave per row)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        _frozen(ctx, 1)
        row, y, stats = _ln_fwd(None, x, weight, bias, eps, ctx.needs_input_grad[0])
        if stats is not None:
            ctx.save_for_backward(row, stats, weight)
        return y

    @staticmethod
    def backward(ctx, g_y):
        row, stats, weight = ctx.saved_tensors
        return _ln_bwd(row, stats, weight, g_y, None), None, None, None


class AddLayerNorm(torch.autograd.Function):
    """(x_new, y) = (a + x, LayerNorm(a + x)).  The backward folds the gradient arriving at x_new from its other
    consumer (the next residual add) into the same launch."""

    @staticmethod
    def forward(ctx, a, x, weight, bias, eps):
        _frozen(ctx, 2)
        xnew, y, stats = _ln_fwd(a, x, weight, bias, eps, ctx.needs_input_grad[0] or ctx.needs_input_grad[1])
        if stats is not None:
            ctx.save_for_backward(xnew, stats, weight)
        return xnew, y

    @staticmethod
    def backward(ctx, g_xnew, g_y):
        xnew, stats, weight = ctx.saved_tensors
        d = g_xnew if g_y is None else _ln_bwd(xnew, stats, weight, g_y, g_xnew)
        return d, d, None, None, None


def layer_norm(x, weight, bias, eps):
    return LayerNorm.apply(x, weight, bias, eps)


def add_layer_norm(a, x, weight, bias, eps):
    """-> (a + x, LayerNorm(a + x))"""
    return AddLayerNorm.apply(a, x, weight, bias, eps)


# --------------------------------------------------------------------------------------- 3x3 convolution (implicit GEMM)
_conv_plan_cache = {}
_conv_pack_cache = {}   # (weight data_ptr, version, dtype, transpose) -> packed tensor (weights are frozen on this path)


_conv_plan_table = None


def _measured_conv_plans():
    """{(M, Cin, Cout, stride): (bm, bn, splits)} measured on an MI355X for the UNet's own shapes (tools/conv_tune.py);
    shapes not listed take the rule of ga_conv3x3_plan."""
    global _conv_plan_table
    if _conv_plan_table is None:
        import json
        from pathlib import Path
        path = Path(__file__).resolve().parent / "conv_plans.json"
        _conv_plan_table = {}
        if path.exists():
            for k, v in json.loads(path.read_text()).items():
                _conv_plan_table[tuple(int(x) for x in k.split(","))] = tuple(v)
    return _conv_plan_table


def conv3x3_plan(B, H, W, Cin, Cout, stride):
    key = (B, H, W, Cin, Cout, stride)
    plan = _conv_plan_cache.get(key)
    if plan is None:
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        hit = _measured_conv_plans().get((B * Ho * Wo, Cin, Cout, stride))
        if hit is not None:
            bm, bn, sp = hit
        else:
            c_bm, c_bn, c_sp, c_ws = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_longlong()
            check(load().ga_conv3x3_plan(B, H, W, Cin, Cout, stride, ctypes.byref(c_bm), ctypes.byref(c_bn), ctypes.byref(c_sp),
                                         ctypes.byref(c_ws)), "ga_conv3x3_plan")
            bm, bn, sp = c_bm.value, c_bn.value, c_sp.value
        # the slices' f32 slabs live in the persistent per-device scratch: a plan never asks for more than it holds
        floats = lambda k: int(load().ga_splitk_workspace_floats(B * Ho * Wo, Cout, bm, bn, k))  # noqa: E731
        while sp > 1 and floats(sp) > LIN_SLAB_FLOATS:
            sp -= 1
        plan = _conv_plan_cache[key] = (bm, bn, sp, floats(sp))
    return plan


def conv3x3_packed_weights(weight, transpose_flip):
    """Blocked [9][N / 64][C / 64][64][64] pack (csrc/conv3x3.hip) of a (Cout, Cin, 3, 3) weight (any strides), cached per
    (storage, version)."""
    key = (weight.data_ptr(), weight._version, weight.dtype, bool(transpose_flip), tuple(weight.stride()), tuple(weight.shape))
    entry = _conv_pack_cache.get(key)
    # the entry is only good for the tensor object it was made from: a freed weight's address can be handed to another
    # tensor of the same shape with other contents (tests do that; the UNet's frozen parameters never move)
    hit = entry[1] if entry is not None and entry[0]() is weight else None
    if hit is None:
        require_cuda(weight)
        Cout, Cin = weight.shape[0], weight.shape[1]
        N, C = (Cin, Cout) if transpose_flip else (Cout, Cin)
        hit = torch.empty(int(load().ga_conv3x3_packed_elems(N, C)), dtype=weight.dtype, device=weight.device)
        so, si, sy, sx = weight.stride()
        check(load().ga_conv3x3_pack_weights(_ptr(weight), _ptr(hit), Cout, Cin, so, si, sy, sx, int(bool(transpose_flip)),
                                             dtype_code(weight), stream_ptr()), "ga_conv3x3_pack_weights")
        if len(_conv_pack_cache) > 512:
            # captured hipGraphs read the packs by raw pointer: only entries whose weight tensor is gone may be dropped
            # (a live weight's pack must stay where it is for as long as a graph may replay against it)
            for dead in [k for k, (ref, _) in _conv_pack_cache.items() if ref() is None]:
                del _conv_pack_cache[dead]
        _conv_pack_cache[key] = (weakref.ref(weight), hit)
    return keep_alive(hit)


CONV_KC = 64   # channels per k-step of ga_conv3x3_nhwc (kKC in csrc/conv3x3.hip): Cin must be a multiple


def conv3x3_supported(x, weight, stride=1):
    return (x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and weight.shape[2:] == (3, 3) and
            weight.shape[1] % CONV_KC == 0 and weight.shape[0] % CONV_KC == 0 and stride in (1, 2))   # both ways round: backward


def conv3x3_nhwc(x, wp, cout, stride=1, bias=None, residual=None, plan=None, gn=None):
    """x (B, Cin, H, W) channels-last, wp the [9][Cout][Cin] pack -> y (B, Cout, Ho, Wo) channels-last.
    gn = (groups, chan_bias (B, Cout) or None): the epilogue also leaves the GroupNorm statistics of y (+ chan_bias) for the
    norm layer that consumes it -> (y, (partials, blocks)) — or (y, None) where the shape is not one that saves a launch."""
    require_cuda(x, wp, bias, residual)
    x = _nhwc(x)
    B, Cin, H, W = x.shape
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    bm, bn, splits = (plan or conv3x3_plan(B, H, W, Cin, cout, stride))[:3]
    y = torch.empty((B, cout, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    ws, tickets = splitk_workspace(x.device, B * Ho * Wo, cout, bm, bn, splits)
    if residual is not None:
        residual = _nhwc(residual)
    _count(("conv3x3", B, Cin, H * W, stride, cout, bias is not None or residual is not None, str(x.dtype)))
    if gn is not None:
        groups, cb = gn
        blocks = int(load().ga_conv3x3_gn_blocks(H, W, cout, groups, bm, bn)) if stride == 1 else 0
        if blocks and gn_two_launch(H * W, cout, groups, x.dtype):
            if cb is not None:
                require_cuda(cb)
                cb = cb.to(x.dtype).contiguous()
                if tuple(cb.shape) != (B, cout):
                    raise GaError(f"chan_bias must be (B, C) = {(B, cout)}, got {tuple(cb.shape)}")
            partials = torch.empty((B, blocks, groups, 2), dtype=torch.float32, device=x.device)
            _check_ticketed(load().ga_conv3x3_nhwc_gn(_ptr(x), _ptr(wp), _ptr(y), _ptr(ws), _ptr(tickets), _ptr(bias),
                                                      _ptr(residual), B, H, W, Cin, cout, bm, bn, splits, dtype_code(x),
                                                      stream_ptr(), _ptr(partials), _ptr(cb), groups),
                            "ga_conv3x3_nhwc_gn", tickets)
            return y, (partials, blocks)
    _check_ticketed(load().ga_conv3x3_nhwc(_ptr(x), _ptr(wp), _ptr(y), _ptr(ws), _ptr(tickets), _ptr(bias), _ptr(residual), B,
                                           H, W, Cin, cout, stride, bm, bn, splits, dtype_code(x), stream_ptr()),
                    "ga_conv3x3_nhwc", tickets)
    return (y, None) if gn is not None else y


GA_ERR_SHAPE = -2   # include/ga_hip.h
_no_fused_upsample = set()   # (B, H, W, Cin, Cout, dtype) the patch kernel does not serve: the caller up-samples itself


def conv3x3_up2x_nhwc(x, wp, cout, bias=None, plan=None):
    """conv3x3(nearest-neighbour 2x up-sampling of x) without the up-sampled tensor (ga_conv3x3_up2x_nhwc), or None when the
    shape is not served by the patch kernel."""
    require_cuda(x, wp, bias)
    x = _nhwc(x)
    B, Cin, H, W = x.shape
    key = (B, H, W, Cin, cout, x.dtype)
    if key in _no_fused_upsample:
        return None
    bm, bn, splits = (plan or conv3x3_plan(B, 2 * H, 2 * W, Cin, cout, 1))[:3]
    y = torch.empty((B, cout, 2 * H, 2 * W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    ws, tickets = splitk_workspace(x.device, B * 4 * H * W, cout, bm, bn, splits)
    rc = load().ga_conv3x3_up2x_nhwc(_ptr(x), _ptr(wp), _ptr(y), _ptr(ws), _ptr(tickets), _ptr(bias), None, B, H, W, Cin,
                                     cout, bm, bn, splits, dtype_code(x), stream_ptr())
    if rc == GA_ERR_SHAPE:
        _no_fused_upsample.add(key)
        return None
    _check_ticketed(rc, "ga_conv3x3_up2x_nhwc", tickets)
    _count(("conv3x3", B, Cin, 4 * H * W, 1, cout, bias is not None, str(x.dtype)))
    return y


class UpsampleConv3x3(torch.autograd.Function):
    """y = conv2d(interpolate(x, scale_factor=2, mode="nearest"), weight, padding=1) + bias (diffusers 0.12.1 Upsample2D)
    in one launch; backward: the stride-1 backward-to-input kernel, then the 2x2 sums of the up-sampling's adjoint."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        if ctx.needs_input_grad[1] or (bias is not None and ctx.needs_input_grad[2]):
            raise GaError("convolution weight gradients are not part of the guided-attention path (frozen UNet)")
        y = conv3x3_up2x_nhwc(x, conv3x3_packed_weights(weight, False), weight.shape[0], bias)
        if y is None:
            up = torch.nn.functional.interpolate(x, scale_factor=2.0, mode="nearest")
            y = conv3x3_nhwc(up, conv3x3_packed_weights(weight, False), weight.shape[0], 1, bias)
        ctx.weight, ctx.in_shape = weight, tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, gy):
        if not ctx.needs_input_grad[0]:
            return None, None, None
        g_up = conv3x3_nhwc(_nhwc(gy), conv3x3_packed_weights(ctx.weight, True), ctx.weight.shape[1], 1)
        B, C, H, W = ctx.in_shape
        gx = torch.ops.aten.upsample_nearest2d_backward(g_up, [2 * H, 2 * W], [B, C, H, W], 2.0, 2.0)
        return gx, None, None


def upsample_conv3x3(x, weight, bias=None):
    return UpsampleConv3x3.apply(x, weight, bias)


class Conv3x3(torch.autograd.Function):
    """y = conv2d(x, weight, bias=None, padding=1, stride) (+ bias + residual) on channels-last 16-bit activations.
    Differentiable w.r.t. x (stride 1: the same kernel on the flipped / transposed pack) and the residual; the weights
    are frozen on this path."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, stride, gn=None):
        """gn (see conv3x3_nhwc): the second output then holds (partials, blocks) for the consuming norm, as a Python attribute
        of the first (`_ga_gn_out`): forward-only side data, no gradient."""
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            raise GaError("convolution weight gradients are not part of the guided-attention path (frozen UNet)")
        wp = conv3x3_packed_weights(weight, False)
        ctx.weight, ctx.stride, ctx.in_shape = weight, stride, tuple(x.shape)
        ctx.has_res = residual is not None
        if gn is None:
            return conv3x3_nhwc(x, wp, weight.shape[0], stride, bias, residual)
        y, made = conv3x3_nhwc(x, wp, weight.shape[0], stride, bias, residual, gn=gn)
        ctx.gn_made = made
        return y

    @staticmethod
    def backward(ctx, gy):
        weight, stride = ctx.weight, ctx.stride
        gx = None
        # ONE dense copy when the gradient is a strided view (a channel slice of a concatenation's gradient): the kernel below
        # and the skip connection's consumer (ga_group_norm_bwd's g_res) both read it
        gy = _nhwc(gy)
        if ctx.needs_input_grad[0]:
            if stride == 1:
                gx = conv3x3_nhwc(gy, conv3x3_packed_weights(weight, True), weight.shape[1], 1)
            else:   # the three down-sampling convolutions: library transposed convolution
                gx = torch.nn.grad.conv2d_input(ctx.in_shape, weight, gy, stride=stride, padding=1)
        return gx, None, None, (gy if ctx.has_res else None), None


def conv3x3(x, weight, bias=None, residual=None, stride=1, gn_for=None):
    """gn_for = (groups, chan_bias or None) of the GroupNorm that consumes the result: where that norm would take two launches
    the convolution's epilogue takes its statistics, and the result carries them (`_ga_gn`, read by group_norm_act)."""
    if gn_for is None:
        return Conv3x3.apply(x, weight, bias, residual, stride)
    groups, cb = gn_for
    B, _, H, W = x.shape
    cout = weight.shape[0]
    if stride != 1 or not gn_two_launch(H * W, cout, groups, x.dtype):
        return Conv3x3.apply(x, weight, bias, residual, stride)
    box = []
    y = _Conv3x3Gn.apply(x, weight, bias, residual, (groups, cb), box)
    if box and box[0] is not None:
        partials, blocks = box[0]
        y._ga_gn = {"partials": partials, "blocks": blocks, "groups": groups, "chan_bias": cb, "shape": tuple(y.shape)}
    return y


class _Conv3x3Gn(Conv3x3):
    """Conv3x3 (stride 1) whose epilogue also takes the consuming GroupNorm's statistics; what it took leaves through `box` (a
    Function's outputs are tensors; this is forward-only side data without a gradient)."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, gn, box):
        y = Conv3x3.forward(ctx, x, weight, bias, residual, 1, gn)
        box.append(ctx.gn_made)
        return y

    @staticmethod
    def backward(ctx, gy):
        return Conv3x3.backward(ctx, gy) + (None,)


conv3x3.supported = conv3x3_supported
conv3x3.upsample = upsample_conv3x3


# --------------------------------------------------------------------------------------- conv_in / conv_out (one side 4 channels)
_thin_pack_cache = {}


def conv3x3_thin_supported(x, weight, stride=1):
    """The UNet's edge convolutions (csrc/thin_conv.hip): 3x3, stride 1, 4 channels in or 4 channels out, 16-bit, W % 16 == 0."""
    if not (x.is_cuda and x.dim() == 4 and x.dtype in (torch.float16, torch.bfloat16) and weight.dtype == x.dtype
            and tuple(weight.shape[2:]) == (3, 3) and stride == 1 and x.shape[1] == weight.shape[1]):
        return False
    cout, cin = weight.shape[0], weight.shape[1]
    if min(cin, cout) != 4 or max(cin, cout) % 64 != 0:     # both ways round: the backward is the other kernel
        return False
    lib = load()
    return bool(lib.ga_conv3x3_thin_supported(x.shape[2], x.shape[3], cin, cout) and
                lib.ga_conv3x3_thin_supported(x.shape[2], x.shape[3], cout, cin))


def conv3x3_thin_packed_weights(weight, transpose_flip):
    """ga_conv3x3_thin_pack of a (Cout, Cin, 3, 3) weight, cached per (storage, version) like conv3x3_packed_weights."""
    key = (weight.data_ptr(), weight._version, weight.dtype, bool(transpose_flip), tuple(weight.stride()), tuple(weight.shape))
    entry = _thin_pack_cache.get(key)
    hit = entry[1] if entry is not None and entry[0]() is weight else None
    if hit is None:
        require_cuda(weight)
        Cout, Cin = weight.shape[0], weight.shape[1]
        hit = torch.empty(int(load().ga_conv3x3_thin_packed_elems(Cout, Cin)), dtype=weight.dtype, device=weight.device)
        so, si, sy, sx = weight.stride()
        check(load().ga_conv3x3_thin_pack(_ptr(weight), _ptr(hit), Cout, Cin, so, si, sy, sx, int(bool(transpose_flip)),
                                          dtype_code(weight), stream_ptr()), "ga_conv3x3_thin_pack")
        if len(_thin_pack_cache) > 64:   # graphs read the packs by raw pointer: only a dead weight's entry may go
            for dead in [k for k, (ref, _) in _thin_pack_cache.items() if ref() is None]:
                del _thin_pack_cache[dead]
        _thin_pack_cache[key] = (weakref.ref(weight), hit)
    return keep_alive(hit)


def conv3x3_thin(x, wp, cout, bias=None):
    """x (B, 4, H, W) dense NCHW -> (B, cout, H, W) channels-last, or x (B, C, H, W) channels-last -> (B, 4, H, W) dense NCHW
    (cout == 4); wp from conv3x3_thin_packed_weights."""
    require_cuda(x, wp, bias)
    B, Cin, H, W = x.shape
    if bias is not None:
        bias = bias.to(x.dtype).contiguous()
    if Cin == 4:
        x = x.contiguous()
        y = torch.empty((B, cout, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        _count(("conv3x3_thin_in", B, 4, H * W, 1, cout, bias is not None, str(x.dtype)))
        check(load().ga_conv3x3_thin_in(_ptr(x), _ptr(wp), _ptr(bias), _ptr(y), B, H, W, cout, dtype_code(x), stream_ptr()),
              "ga_conv3x3_thin_in")
        return y
    if cout != 4:
        raise GaError(f"conv3x3_thin serves 4 -> C and C -> 4 channels, got {Cin} -> {cout}")
    x = _nhwc(x)
    y = torch.empty((B, 4, H, W), dtype=x.dtype, device=x.device)
    _count(("conv3x3_thin_out", B, Cin, H * W, 1, 4, bias is not None, str(x.dtype)))
    check(load().ga_conv3x3_thin_out(_ptr(x), _ptr(wp), _ptr(bias), _ptr(y), B, H, W, Cin, dtype_code(x), stream_ptr()),
          "ga_conv3x3_thin_out")
    return y


class Conv3x3Thin(torch.autograd.Function):
    """y = conv2d(x, weight, bias, padding=1) for the UNet's conv_in / conv_out.  Differentiable w.r.t. x: the adjoint of the
    4 -> C kernel is the C -> 4 kernel on the transposed, mirrored pack and the other way round; the weights are frozen."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        if ctx.needs_input_grad[1] or (bias is not None and ctx.needs_input_grad[2]):
            raise GaError("convolution weight gradients are not part of the guided-attention path (frozen UNet)")
        ctx.weight = weight
        return conv3x3_thin(x, conv3x3_thin_packed_weights(weight, False), weight.shape[0], bias)

    @staticmethod
    def backward(ctx, gy):
        if not ctx.needs_input_grad[0]:
            return None, None, None
        weight = ctx.weight
        return conv3x3_thin(gy, conv3x3_thin_packed_weights(weight, True), weight.shape[1]), None, None


def conv3x3_thin_apply(x, weight, bias=None):
    return Conv3x3Thin.apply(x, weight, bias)


conv3x3_thin_apply.supported = conv3x3_thin_supported


# --------------------------------------------------------------------------------------- tiled self-attention
def _sub_ptr(t, elem_offset):
    return ctypes.c_void_p(t.data_ptr() + elem_offset * t.element_size())


def self_attn_fwd(q, k, v, heads, scale, want_lse=True):
    """q,k,v (B,N,C) projections -> (o (B,N,C), lse (B*heads,N) f32 log2-domain or None)."""
    require_cuda(q, k, v)
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    B, N, C = q.shape
    o = torch.empty_like(q)
    lse = torch.empty((B * heads, N), dtype=torch.float32, device=q.device) if want_lse else None
    _count(("self_attn_fwd", B, heads, N, N, C // heads, bool(want_lse), str(q.dtype)))
    check(load().ga_self_attn_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(lse), B, heads, N, C // heads, 0,
                                  float(scale), dtype_code(q), stream_ptr()), "ga_self_attn_fwd")
    return o, lse


def self_attn_bwd(q, k, v, o, d_o, lse, heads, scale):
    require_cuda(q, k, v, o, d_o, lse)
    B, N, C = q.shape
    d_o = d_o.contiguous()
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty_like(lse)
    _count(("self_attn_bwd", B, heads, N, N, C // heads, True, str(q.dtype)))
    check(load().ga_self_attn_bwd(_ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(d_o), _ptr(lse), _ptr(delta), _ptr(dq),
                                  _ptr(dk), _ptr(dv), B, heads, N, C // heads, 0, float(scale), dtype_code(q),
                                  stream_ptr()), "ga_self_attn_bwd")
    return dq, dk, dv


class SelfAttention(torch.autograd.Function):
    """softmax(scale q k^T) v for long key sequences; the probabilities are never materialised."""

    @staticmethod
    def forward(ctx, q, k, v, heads, scale):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        need = any(ctx.needs_input_grad[:3])
        o, lse = self_attn_fwd(q, k, v, heads, scale, want_lse=need)
        if need:
            ctx.save_for_backward(q, k, v, o, lse)
        ctx.meta = (heads, scale)
        return o

    @staticmethod
    def backward(ctx, d_o):
        q, k, v, o, lse = ctx.saved_tensors
        heads, scale = ctx.meta
        dq, dk, dv = self_attn_bwd(q, k, v, o, d_o, lse, heads, scale)
        return dq, dk, dv, None, None


class SelfAttentionFusedQKV(torch.autograd.Function):
    """Same kernels on ONE (B, N, 3C) tensor [q | k | v] (the output of a fused QKV projection): the kernels read
    the three column slices in place (row stride 3C) and the backward writes dq | dk | dv into one (B, N, 3C)
    tensor, so the projection's backward is a single GEMM too."""

    @staticmethod
    def forward(ctx, qkv, heads, scale):
        require_cuda(qkv)
        qkv = qkv.contiguous()
        B, N, C3 = qkv.shape
        C = C3 // 3
        need = ctx.needs_input_grad[0]
        o = torch.empty((B, N, C), dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty((B * heads, N), dtype=torch.float32, device=qkv.device) if need else None
        _count(("self_attn_fwd", B, heads, N, N, C // heads, bool(need), str(qkv.dtype)))
        check(load().ga_self_attn_fwd(_sub_ptr(qkv, 0), _sub_ptr(qkv, C), _sub_ptr(qkv, 2 * C), _ptr(o), _ptr(lse), B,
                                      heads, N, C // heads, C3, float(scale), dtype_code(qkv), stream_ptr()),
              "ga_self_attn_fwd")
        if need:
            ctx.save_for_backward(qkv, o, lse)
        ctx.meta = (heads, scale)
        return o

    @staticmethod
    def backward(ctx, d_o):
        qkv, o, lse = ctx.saved_tensors
        heads, scale = ctx.meta
        B, N, C3 = qkv.shape
        C = C3 // 3
        d_o = d_o.contiguous()
        d_qkv = torch.empty_like(qkv)
        delta = torch.empty_like(lse)
        _count(("self_attn_bwd", B, heads, N, N, C // heads, True, str(qkv.dtype)))
        check(load().ga_self_attn_bwd(_sub_ptr(qkv, 0), _sub_ptr(qkv, C), _sub_ptr(qkv, 2 * C), _ptr(o), _ptr(d_o),
                                      _ptr(lse), _ptr(delta), _sub_ptr(d_qkv, 0), _sub_ptr(d_qkv, C),
                                      _sub_ptr(d_qkv, 2 * C), B, heads, N, C // heads, C3, float(scale),
                                      dtype_code(qkv), stream_ptr()), "ga_self_attn_bwd")
        return d_qkv, None, None


def self_attention_supported(q, heads, channels=None):
    d = (channels or q.shape[-1]) // heads
    return q.is_cuda and d % 8 == 0 and d <= (80 if q.dtype == torch.float32 else 160) and q.dtype in _lib.DTYPE_CODE


# --------------------------------------------------------------------------------------- Linear layers with folded neighbours
_lin_ws = {}        # (device index, stream) -> {"slabs": f32 tensor, "tickets": int32 tensor}; never freed (captured graphs hold the pointers)
_lin_plan_cache = {}
_lin_plan_table = None
LIN_SLAB_FLOATS = 64 * 2 ** 20     # 256 MB of split-K slabs per device (of 288 GB), allocated once (outside any capture)
LIN_TICKETS = 1 << 16


def linear_workspace(device):
    """The split-K scratch of the CURRENT stream on `device`: slices of one launch hand their partial tiles to the last arriver
    through it, and two launches may only share it under stream order — so every stream that launches gets its own (the
    header's "stream-ordered use" made structural: the default stream, the one side stream of warm-ups / captures / micro-replays,
    a GUI thread's stream never meet in one slab or ticket array)."""
    key = _stream_key(device)
    ws = _lin_ws.get(key)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            raise GaError("the Linear workspace must exist before a hipGraph capture (call ops.prepare_device first)")
        ws = _lin_ws[key] = {"slabs": torch.empty(LIN_SLAB_FLOATS, dtype=torch.float32, device=device),
                             "tickets": torch.zeros(LIN_TICKETS, dtype=torch.int32, device=device)}
    return ws


def _check_ticketed(rc, what, tickets):
    """check() for a launch that takes arrival tickets: a launch that failed after some slices had arrived would leave its
    ticket words non-zero, and every later launch on those tiles would never see a last arriver (Y silently unwritten) —
    return them to zero before raising."""
    if rc != 0 and tickets is not None and not torch.cuda.is_current_stream_capturing():
        tickets.zero_()
    check(rc, what)


def tickets_are_zero(device=None):
    """True when every arrival-ticket word of every stream's scratch on `device` is zero (what every completed launch leaves;
    synchronises).  GraphRunner.release asserts it."""
    torch.cuda.synchronize(device)
    idx = torch.device(device).index if device is not None else torch.cuda.current_device()
    words = [ws["tickets"] for (d, _), ws in _lin_ws.items() if d == idx] + [t for (d, _), t in _tickets.items() if d == idx]
    return all(int(w.abs().sum().item()) == 0 for w in words)


def splitk_workspace(device, M, N, bm, bn, splits):
    """(f32 slabs, tickets) of an in-launch split-K reduction from the persistent per-device scratch, or (None, None)."""
    if splits <= 1:
        return None, None
    ws = linear_workspace(device)
    need = int(load().ga_splitk_workspace_floats(M, N, bm, bn, splits))
    if need > ws["slabs"].numel() or -(-M // bm) * -(-N // bn) > ws["tickets"].numel():
        raise GaError(f"split-K workspace too small: M={M} N={N} plan {(bm, bn, splits)} needs {need} floats")
    return ws["slabs"], ws["tickets"]


def prepare_device(device, stream=None):
    """Allocate the persistent scratch of the kernels (split-K slabs, arrival tickets) for the current stream (or `stream`) on
    `device` — before any hipGraph capture on that stream: captured launches keep these never-moving buffers."""
    device = torch.device(device)
    if device.type == "cuda":
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if stream is not None:
            with torch.cuda.stream(stream):
                linear_workspace(device)
                _ticket(device)
        else:
            linear_workspace(device)
            _ticket(device)


def _measured_linear_plans():
    global _lin_plan_table
    if _lin_plan_table is None:
        import json
        from pathlib import Path
        path = Path(__file__).resolve().parent / "linear_plans.json"
        _lin_plan_table = {}
        if path.exists():
            for k, v in json.loads(path.read_text()).items():
                _lin_plan_table[tuple(int(x) for x in k.split(","))] = tuple(v)
    return _lin_plan_table


def library_block(M, C):
    """True when tools/linear_tune.py measured the LIBRARY form of a whole transformer block (hipBLASLt GEMMs + the separate
    LayerNorm / GEGLU / residual kernels) faster than the folded ga_linear_fused form at M tokens x C channels — the long, wide
    GEMMs of SDXL's 1280-channel level at batch 3 (K = 1280 - 5120, N up to 10240); key "M,C,-1,-1" of linear_plans.json."""
    return bool(_measured_linear_plans().get((int(M), int(C), -1, -1), (0,))[0])


LINEAR_STREAM = _lib.GA_LINEAR_STREAM     # `stages` value of the persistent one-workgroup-per-CU form (csrc/linear.hip)
LINEAR_STREAM_PLAN = (128, 128, 1, LINEAR_STREAM)


def linear_stream_serves(K, parts, ln, bias, residual, want_preact, want_ln_stats, want_row_partials):
    """What linear_stream_kernel takes: the LayerNorm-folded no-grad forms (optionally GEGLU) at K >= 320."""
    return (ln is not None and bias is None and residual is None and not want_preact and not want_ln_stats
            and not want_row_partials and K // 64 >= 5 and 2 <= parts <= 20)


def linear_plan(M, K, N, geglu=False, stream_ok=False):
    """(bm, bn, splits[, stages]) for Y[M][N or N/2] = X[M][K] W[N][K]^T: the measured table (tools/linear_tune.py) or a rule:
    about one workgroup per CU and more; the depth is split when the tiles alone leave most of the chip idle.  stream_ok: the
    call is one linear_stream_kernel serves — the table's "M,K,N,geglu,8" entry says whether that form measured faster."""
    key = (int(M), int(K), int(N), int(bool(geglu)))
    if stream_ok and _measured_linear_plans().get(key + (LINEAR_STREAM,), (0,))[0]:
        return LINEAR_STREAM_PLAN
    plan = _lin_plan_cache.get(key)
    if plan is None:
        plan = _measured_linear_plans().get(key)
        if plan is None:
            n_out = N // 2 if geglu else N
            steps = K // 64
            best, best_cost = None, None
            for bm, bn in ((128, 128), (128, 64), (64, 128), (64, 64)):
                outc = bn // 2 if geglu else bn
                tiles = -(-M // bm) * -(-n_out // outc)
                waste = (-(-M // bm) * bm) * (-(-n_out // outc) * outc) / (M * n_out)
                for sp in (1, 2, 3, 4, 6, 8, 12, 16):
                    if sp > 1 and (steps // sp < 4 or sp * tiles * bm * bn > LIN_SLAB_FLOATS or tiles > LIN_TICKETS):
                        break     # too shallow a slice, or more slabs / tickets than the persistent scratch holds
                    wgs = tiles * sp
                    rounds = -(-wgs // 256)
                    per_wg = -(-steps // sp) * (0.09 if bm * bn == 128 * 128 else 0.055 if bm * bn == 128 * 64 else 0.04)
                    cost = rounds * (1.2 + per_wg) * waste ** 0.5 + (1.5 + sp * bm * bn * 4 / 100e3 if sp > 1 else 0.0)
                    if best_cost is None or cost < best_cost:
                        best, best_cost = (bm, bn, sp), cost
            plan = best
        _lin_plan_cache[key] = plan
    return plan


def linear_fused(x, weight, bias=None, residual=None, geglu=False, want_preact=False, ln=None, want_ln_stats=False,
                 want_row_partials=False, plan=None, out=None, gn=None):
    """Y = epilogue(X W^T) through ga_linear_fused (see include/ga_hip.h).
    x (..., K): rows `x.stride(-2)` elements apart, last dimension contiguous; weight (N, K) contiguous.
    ln = (partials (M, parts, 2) f32, colsum (N,) f32, shift (N,) f32, eps): LayerNorm folded in front (then `weight` is
    gamma o W and `bias` is ignored).  gn = (groups, hw): the rows are hw pixels per image and the epilogue also leaves the
    GroupNorm statistics of the stored result for the norm that consumes it ("gn": (partials, blocks), or None where the plan
    or shape does not serve).  -> dict(y, preact, ln_stats, row_partials, parts, gn)."""
    require_cuda(x, weight, bias, residual)
    if x.stride(-1) != 1 or not weight.is_contiguous():
        raise GaError("linear_fused needs unit-stride feature axes")
    K = x.shape[-1]
    lead = x.shape[:-1]
    M = 1
    for d in lead:
        M *= d
    ldx = x.stride(-2) if x.dim() > 1 else K
    if x.dim() > 2 and any(x.stride(i) != x.stride(i + 1) * x.shape[i + 1] for i in range(x.dim() - 2)):
        x = x.contiguous()
        ldx = K
    N = weight.shape[0]
    n_out = N // 2 if geglu else N
    if plan is None:
        stream_ok = ln is not None and linear_stream_serves(K, ln[0].shape[1], ln, bias, residual, want_preact, want_ln_stats,
                                                            want_row_partials)
        plan = linear_plan(M, K, N, geglu, stream_ok)
    plan = tuple(plan)
    bm, bn, splits = plan[:3]
    stages = plan[3] if len(plan) > 3 else 0
    y = out if out is not None else torch.empty(lead + (n_out,), dtype=x.dtype, device=x.device)
    ep = _lib.ga_linear_epilogue_t()
    ep.bias = bias.data_ptr() if bias is not None else None
    if residual is not None:
        if residual.stride(-1) != 1 or tuple(residual.shape) != tuple(y.shape):
            raise GaError("residual must have the result's shape and a unit-stride feature axis")
        if residual.dim() > 2 and any(residual.stride(i) != residual.stride(i + 1) * residual.shape[i + 1]
                                      for i in range(residual.dim() - 2)):
            residual = residual.contiguous()
        ep.residual, ep.ld_res = residual.data_ptr(), residual.stride(-2) if residual.dim() > 1 else n_out
    ep.geglu = int(bool(geglu))
    preact = None
    if geglu and want_preact:
        preact = torch.empty(lead + (N,), dtype=x.dtype, device=x.device)
        ep.preact, ep.ld_pre = preact.data_ptr(), N
    ln_stats = None
    if ln is not None:
        partials, colsum, shift, eps = ln
        if partials.dtype != torch.float32 or tuple(partials.shape) != (M, partials.shape[1], 2) or not partials.is_contiguous():
            raise GaError("LayerNorm partial sums must be a contiguous (M, parts, 2) float32 tensor")
        ep.ln_partials, ep.ln_parts, ep.ln_eps = partials.data_ptr(), partials.shape[1], float(eps)
        ep.ln_colsum, ep.ln_shift = colsum.data_ptr(), shift.data_ptr()
        if want_ln_stats:
            ln_stats = torch.empty((M, 2), dtype=torch.float32, device=x.device)
            ep.ln_stats_out = ln_stats.data_ptr()
    outc = bn // 2 if geglu else bn
    parts = -(-n_out // outc)
    row_partials = None
    if want_row_partials:
        row_partials = torch.empty((M, parts, 2), dtype=torch.float32, device=x.device)
        ep.row_partials_out = row_partials.data_ptr()
    gn_made = None
    if gn is not None and not geglu and stages != LINEAR_STREAM:
        groups, hw = gn
        blocks = int(load().ga_linear_gn_blocks(hw, N, groups, bm, bn)) if M % hw == 0 else 0
        if blocks:
            gn_partials = torch.empty((M // hw, blocks, groups, 2), dtype=torch.float32, device=x.device)
            ep.gn_partials, ep.gn_groups, ep.gn_hw = gn_partials.data_ptr(), groups, hw
            gn_made = (gn_partials, blocks)
    slabs = tickets = None
    if splits > 1:
        ws = linear_workspace(x.device)
        tiles = -(-M // bm) * parts
        if splits * tiles * bm * bn > ws["slabs"].numel() or tiles > ws["tickets"].numel():
            raise GaError(f"split-K workspace too small for M={M} N={N} plan {(bm, bn, splits)}")
        slabs, tickets = ws["slabs"], ws["tickets"]
    _count(("linear", M, K, 0, int(bool(geglu)) + 2 * int(ln is not None) + 4 * int(residual is not None), N, bias is not None,
            str(x.dtype)))
    _check_ticketed(load().ga_linear_fused(_ptr(x), ldx, _ptr(weight), _ptr(y), y.stride(-2) if y.dim() > 1 else n_out,
                                           ctypes.byref(ep), _ptr(slabs), _ptr(tickets), M, K, N, bm, bn, splits, stages,
                                           dtype_code(x), stream_ptr()), "ga_linear_fused", tickets)
    return {"y": y, "preact": preact, "ln_stats": ln_stats, "row_partials": row_partials, "parts": parts, "gn": gn_made}
