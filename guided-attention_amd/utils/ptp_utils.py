"""Attention hook layer with the reference's interface (utils/ptp_utils.py:59-289): the attention
processor that computes softmax(scale Q K^T) and hands it to the controller, the controller /
AttentionStore classes, `register_attention_control` and `aggregate_attention`.

What differs underneath: the probabilities are produced by the HIP capture kernel
(`ga_attn_capture_fwd`: QK^T on MFMA, row softmax in registers, P written once, PV fused) directly
from / into the projection layout, and the store can be told to materialise only the maps the loss
consumes.  Capture policies (`AttentionStore(capture=...)`):
  "loss-only"  (default)  keep P only for cross-attention layers with attention_res^2 pixels — the
               five 16x16 maps `aggregate_attention(res=16, is_cross=True)` reads;
  "reference"  keep every P with at most 32^2 pixels, cross and self, as the reference does (:228).
Layers whose P is not kept still tick the controller (the layer counter drives between_steps), with a
`ProbsNotCaptured` placeholder carrying only the shape.
"""
import abc
from typing import List

import torch
import torch.nn.functional as F

from .. import ops
from .._lib import GaError
from . import shared_state as state

MAX_CAPTURE_KEYS = 128  # ga_attn_capture_* keep the whole key axis on chip up to this length


# ---- notebook helpers of the reference (utils/ptp_utils.py:14-56), host-side, off the hot path.  The reference draws
# with OpenCV and shows through IPython; neither is required here: PIL draws, and `display` is used when importable.
def text_under_image(image, text, text_color=(0, 0, 0)):
    """(h, w, 3) uint8 -> (h + 20 %, w, 3) with `text` centred in the white strip below (reference :14-23)."""
    import numpy as np
    from PIL import Image, ImageDraw
    from . import helpers
    h, w, c = image.shape
    offset = int(h * .2)
    img = np.ones((h + offset, w, c), dtype=np.uint8) * 255
    img[:h] = image
    pil = Image.fromarray(img)
    draw = ImageDraw.Draw(pil)
    font = helpers._font(max(10, offset // 2))
    box = draw.textbbox((0, 0), text, font=font)
    tw, th = box[2] - box[0], box[3] - box[1]
    draw.text(((w - tw) // 2, h + (offset - th) // 2 - box[1]), text, fill=tuple(text_color), font=font)
    return np.asarray(pil).copy()


def view_images(images, num_rows=1, offset_ratio=0.02, display_image=True):
    """Tile equally sized (h, w, 3) panels into `num_rows` rows separated by white gutters of `offset_ratio` of the
    panel height and return the PIL image — the notebook helper of the reference (:26-56), including its habit of
    appending `len(images) % num_rows` blank panels before the column count is taken."""
    import numpy as np
    from PIL import Image
    single = not isinstance(images, list) and np.ndim(images) != 4
    panels = [np.asarray(p, dtype=np.uint8) for p in ([images] if single else images)]
    h, w = panels[0].shape[:2]
    panels += [np.full_like(panels[0], 255)] * (0 if single else len(panels) % num_rows)
    cols, gap = len(panels) // num_rows, int(h * offset_ratio)
    v_gutter = np.full((h, gap, 3), 255, np.uint8)
    strips = []
    for r in range(num_rows):
        row = panels[r * cols:(r + 1) * cols]
        parts = [x for panel in row for x in (panel, v_gutter)][:-1]          # panel | gutter | panel | ... | panel
        strips.append(np.concatenate(parts, axis=1))
        strips.append(np.full((gap, strips[-1].shape[1], 3), 255, np.uint8))
    sheet = Image.fromarray(np.concatenate(strips[:-1], axis=0))
    if display_image:
        try:
            from IPython.display import display
        except ImportError:
            display = None
        if display is not None:
            display(sheet)
    return sheet


class ProbsNotCaptured:
    """Shape-only stand-in for a probability tensor that was deliberately not materialised."""

    def __init__(self, shape, dtype, device):
        self.shape, self.dtype, self.device = torch.Size(shape), dtype, device


def _tiled_attention(q, k, v, heads, scale):
    """softmax(scale q k^T) v for long key sequences (self-attention), probabilities never materialised:
    the HIP flash kernels (ga_self_attn_fwd/bwd).  q,k,v: (B, N, heads*d) projections -> (B, N, heads*d)."""
    if not q.is_cuda:
        raise GaError("attention runs on the GPU only; there is no CPU fallback")
    if ops.self_attention_supported(q, heads):
        return ops.SelfAttention.apply(q, k, v, heads, scale)
    # outside the kernels' envelope (f32 with head_dim > 80, head_dim > 160 or not a multiple of 8): the
    # materialising GPU path, explicitly
    return _materialised_attention(q, k, v, heads, scale)[0]


def _materialised_attention(q, k, v, heads, scale):
    """Reference-capture mode for key sequences too long for the capture kernel (self-attention with
    N <= 32^2): P is materialised in the reference's (B*heads, N, Kt) layout."""
    B, N, C = q.shape
    d = C // heads

    def split(t):
        return t.view(B, -1, heads, d).transpose(1, 2).reshape(B * heads, -1, d)

    probs = torch.softmax(torch.bmm(split(q), split(k).transpose(1, 2)) * scale, dim=-1)
    o = torch.bmm(probs, split(v)).view(B, heads, N, d).transpose(1, 2).reshape(B, N, C)
    return o, probs


def _kv_key(attn, context):
    return (context.data_ptr(), context._version, tuple(context.shape), context.dtype, attn.to_k.weight._version,
            attn.to_v.weight._version, attn.to_k.weight.data_ptr())


def cached_context_projections(attn, context):
    """to_k / to_v of the text context.  They depend only on the prompt embedding and the frozen weights, i.e. they
    are constant across all ~170 UNet passes of an image, so they are computed once per (context tensor, version)
    and kept resident (the reference recomputes the 32 small GEMMs in every forward).  An in-place change of the
    embedding or of the weights bumps a version counter and invalidates the entry."""
    cache = attn.__dict__.setdefault("_kv_cache", {})
    k = _kv_key(attn, context)
    hit = cache.get(k)
    if hit is None:
        with torch.no_grad():
            hit = [context, attn.to_k(context), attn.to_v(context), 0]   # [ctx, K, V, pin count]
        unpinned = [key for key, e in cache.items() if e[3] == 0]
        while len(unpinned) >= KV_CACHE_ENTRIES:   # FIFO over the entries no captured hipGraph reads
            cache.pop(unpinned.pop(0))
        cache[k] = hit
    return hit[1], hit[2]


KV_CACHE_ENTRIES = 4


def pin_context_projections(unet, storages, delta):
    """Pin (+1) / unpin (-1) every cached (K, V) pair whose context lives in one of `storages` (data pointers of
    untyped storages).  Captured hipGraphs read these K/V tensors by raw pointer: a pinned entry is never evicted,
    so the memory stays alive and `refresh_context_projections` keeps it current.  -> number of entries touched."""
    n = 0
    for mod in unet.modules():
        cache = mod.__dict__.get("_kv_cache")
        if not cache:
            continue
        for e in cache.values():
            if e[0].untyped_storage().data_ptr() in storages:
                e[3] = max(0, e[3] + delta)
                n += 1
    return n


def fused_qkv_weight(attn):
    """[to_q | to_k | to_v] weights row-concatenated (built once per module, rebuilt if a weight changes).  Parameter
    names / state_dict are untouched."""
    ws = (attn.to_q.weight, attn.to_k.weight, attn.to_v.weight)
    key = tuple((w.data_ptr(), w._version) for w in ws)
    cache = attn.__dict__.get("_qkv_cache")
    if cache is None or cache[0] != key:
        with torch.no_grad():
            cache = (key, torch.cat(ws, dim=0))
        attn.__dict__["_qkv_cache"] = cache
    return cache[1]


def fused_qkv_projection(attn, hidden_states):
    """[to_q | to_k | to_v](hidden_states) as ONE GEMM."""
    return F.linear(hidden_states, fused_qkv_weight(attn))


def refresh_context_projections(unet):
    """Recompute every cached (K, V) pair IN PLACE from its context tensor and re-key it: used after the static
    prompt-embedding buffer of the hipGraph runner was overwritten (the captured kernels read these tensors)."""
    for mod in unet.modules():
        cache = mod.__dict__.get("_kv_cache")
        if not cache:
            continue
        fresh = {}
        with torch.no_grad():
            for ctx, k, v, pins in cache.values():
                k.copy_(mod.to_k(ctx))
                v.copy_(mod.to_v(ctx))
                fresh[_kv_key(mod, ctx)] = [ctx, k, v, pins]
        cache.clear()
        cache.update(fresh)


_pww_cache = {}


def paint_with_words_bias(n_pixels, n_keys, dtype, device):
    """The additive mask of the reference's paint-with-words branch (utils/ptp_utils.py:113-131) for the current
    step, or None when it is off: -> (mask (N, 77) with `paint_with_words_weight` inside each BOX token's (shrunk)
    rectangle at that layer's resolution, multiplier 0.4 * log(1 + sigma_t)).  Active while
    cur_time_step_iter < curHyperParams["paint_with_words_stop"] (0 = off, the default), for 77-key layers only."""
    import math
    hp = state.curHyperParams or {}
    stop = hp.get("paint_with_words_stop", 0)
    if not stop or n_keys != 77 or not state.cur_time_step_iter < stop:
        return None
    from . import helpers
    w = hp.get("paint_with_words_weight", 1.0)
    hw = int(n_pixels ** .5)
    boxes = tuple((idx, info["loss"].as_tuple()) for idx, info in state.config.token_dict.items()
                  if info["loss_type"] == helpers.AnnotationType.BOX)
    key = (hw, n_pixels, boxes, float(hp["shrink_factor"]), float(w), dtype, str(device))
    mask = _pww_cache.get(key)
    if mask is None:
        m = torch.zeros((hw, hw, 77))
        for idx, geom in boxes:
            m[:, :, idx][helpers.inside_mask(helpers.Rect(*geom, 1).of_size(hw), hw)] = w
        mask = m.reshape(hw * hw, 77).to(device=device, dtype=dtype)
        if len(_pww_cache) > 64:
            _pww_cache.clear()
        _pww_cache[key] = mask
    return mask, .4 * math.log(1 + float(state.get_sigma()))


class AttendExciteCrossAttnProcessor:
    """proc(attn, hidden_states, encoder_hidden_states=None, attention_mask=None) -> hidden_states"""

    def __init__(self, attnstore, place_in_unet):
        self.attnstore = attnstore
        self.place_in_unet = place_in_unet

    supports_folded_layer_norm = True   # accepts `folded=` (see __call__); a foreign processor gets the plain protocol

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, folded=None):
        """The reference's protocol, plus `folded` (this build's transformer blocks only): dict(partials, norm, residual,
        want_partials).  Then `hidden_states` is the RAW residual stream: the block's LayerNorm is applied inside the
        q / qkv projection (fused_linear.ln_linear), the output projection adds its bias and the residual in its epilogue
        and the call returns (new residual stream, its row partial sums or None) — no LayerNorm launch, no add launch."""
        if attention_mask is not None:
            raise GaError("attention masks are not part of the guided-attention path")
        from .. import fused_linear as fl
        is_cross = encoder_hidden_states is not None
        store = self.attnstore

        residual = [folded["residual"]] if folded is not None else None   # replaced by ln_linear's pass-through of the stream

        def finish(out):
            if folded is not None:
                return fl.linear(out, attn.to_out[0].weight, attn.to_out[0].bias, residual=residual[0],
                                 want_partials=folded["want_partials"])
            return attn.to_out[1](attn.to_out[0](out))

        if not is_cross:
            n_pix = hidden_states.shape[1]
            fused_ok = (n_pix > MAX_CAPTURE_KEYS or torch.is_grad_enabled()) and hidden_states.is_cuda and \
                ops.self_attention_supported(hidden_states, attn.heads, attn.to_q.out_features)
            if fused_ok and not (store is not None and store.wants_probs(False, n_pix)):
                # self-attention without capture: one fused QKV GEMM, flash kernels on its column slices
                if folded is not None:
                    qkv, residual[0] = fl.ln_linear(hidden_states, folded["partials"], folded["norm"], fused_qkv_weight(attn))
                else:
                    qkv = fused_qkv_projection(attn, hidden_states)
                out = ops.SelfAttentionFusedQKV.apply(qkv, attn.heads, attn.scale)
                if store is not None:
                    store(ProbsNotCaptured((hidden_states.shape[0] * attn.heads, n_pix, n_pix), out.dtype, out.device),
                          False, self.place_in_unet)
                return finish(out)
            if folded is not None:   # the materialising self-attention paths read the normalised rows themselves
                norm = folded["norm"]
                hidden_states = ops.layer_norm(hidden_states, norm.weight, norm.bias, norm.eps)
        context = encoder_hidden_states if is_cross else hidden_states
        if folded is not None and is_cross:
            query, residual[0] = fl.ln_linear(hidden_states, folded["partials"], folded["norm"], attn.to_q.weight)
        else:
            query = attn.to_q(hidden_states)
        if is_cross and not context.requires_grad:
            key, value = cached_context_projections(attn, context)
        else:
            key = attn.to_k(context)
            value = attn.to_v(context)
        n_pix, n_keys = query.shape[1], key.shape[1]
        want = store is not None and store.wants_probs(is_cross, n_pix)
        probs = None
        ctx_needs_grad = torch.is_grad_enabled() and (key.requires_grad or value.requires_grad)
        pww = paint_with_words_bias(n_pix, n_keys, query.dtype, query.device) if is_cross else None
        if pww is not None and not ctx_needs_grad:
            if int(n_pix ** .5) ** 2 != n_pix:
                raise GaError("paint-with-words needs a square attention map (reference: int(N ** .5))")
            out, probs = ops.AttnCapturePaintWithWords.apply(query, key, value, attn.heads, attn.scale, want, *pww)
            if not want:
                probs = None
        elif n_keys <= MAX_CAPTURE_KEYS and not ctx_needs_grad:
            # the capture kernels: whole key axis on chip; context (text) carries no gradient
            out, probs = ops.AttnCapture.apply(query, key, value, attn.heads, attn.scale, want)
            if not want:
                probs = None
        elif want:
            out, probs = _materialised_attention(query, key, value, attn.heads, attn.scale)
        else:
            out = _tiled_attention(query, key, value, attn.heads, attn.scale)
        if store is not None:
            if probs is None:
                probs = ProbsNotCaptured((query.shape[0] * attn.heads, n_pix, n_keys), query.dtype, query.device)
            store(probs, is_cross, self.place_in_unet)
        return finish(out)


def default_processor():
    """What an attention layer uses when no controller is registered: the HIP kernels, no capture."""
    return AttendExciteCrossAttnProcessor(attnstore=None, place_in_unet=None)


def register_attention_control(model, controller):
    attn_procs = {}
    count = 0
    for name in model.unet.attn_processors.keys():
        if name.startswith("mid_block"):
            place = "mid"
        elif name.startswith("up_blocks"):
            place = "up"
        elif name.startswith("down_blocks"):
            place = "down"
        else:
            continue
        count += 1
        attn_procs[name] = AttendExciteCrossAttnProcessor(attnstore=controller, place_in_unet=place)
    model.unet.set_attn_processor(attn_procs)
    controller.num_att_layers = count


class AttentionControl(abc.ABC):
    def __init__(self):
        self.cur_step = 0
        self.num_att_layers = -1
        self.cur_att_layer = 0

    def step_callback(self, x_t):
        return x_t

    def between_steps(self):
        return

    @property
    def num_uncond_att_layers(self):
        return 0

    def wants_probs(self, is_cross: bool, n_pixels: int) -> bool:
        return True

    @abc.abstractmethod
    def forward(self, attn, is_cross: bool, place_in_unet: str):
        raise NotImplementedError

    def __call__(self, attn, is_cross: bool, place_in_unet: str):
        if self.cur_att_layer >= self.num_uncond_att_layers:
            self.forward(attn, is_cross, place_in_unet)
        self.cur_att_layer += 1
        if self.cur_att_layer == self.num_att_layers + self.num_uncond_att_layers:
            self.cur_att_layer = 0
            self.cur_step += 1
            self.between_steps()

    def reset(self):
        self.cur_step = 0
        self.cur_att_layer = 0

    def flush(self):
        """Close a forward that stopped before its last attention layer (truncated guidance forward): publish
        what was captured exactly as the layer counter would have after the final layer."""
        if self.cur_att_layer != 0:
            self.cur_att_layer = 0
            self.cur_step += 1
            self.between_steps()


class EmptyControl(AttentionControl):
    def wants_probs(self, is_cross, n_pixels):
        return False

    def forward(self, attn, is_cross: bool, place_in_unet: str):
        return attn


class AttentionStore(AttentionControl):
    @staticmethod
    def get_empty_store():
        return {"down_cross": [], "mid_cross": [], "up_cross": [],
                "down_self": [], "mid_self": [], "up_self": []}

    def __init__(self, save_global_store=False, capture="loss-only", attention_res=16):
        super().__init__()
        if capture not in ("loss-only", "reference"):
            raise ValueError(f"capture must be 'loss-only' or 'reference', got {capture!r}")
        self.save_global_store = save_global_store
        self.capture = capture
        self.attention_res = attention_res
        self.step_store = self.get_empty_store()
        self.attention_store = {}
        self.global_store = {}
        self.curr_step_index = 0

    def _save_all(self):
        return bool(getattr(state.config, "save_individual_CA_maps", False))

    def wants_probs(self, is_cross, n_pixels):
        if self.capture == "reference" or self.save_global_store:
            return self._save_all() or n_pixels <= 32 ** 2
        return is_cross and n_pixels == self.attention_res ** 2

    def forward(self, attn, is_cross: bool, place_in_unet: str):
        if isinstance(attn, ProbsNotCaptured):
            return attn
        key = f"{place_in_unet}_{'cross' if is_cross else 'self'}"
        if self._save_all() or attn.shape[1] <= 32 ** 2:  # avoid memory overhead (reference :228)
            self.step_store[key].append(attn)
        return attn

    def between_steps(self):
        self.attention_store = self.step_store
        if self.save_global_store:
            with torch.no_grad():
                if len(self.global_store) == 0:
                    self.global_store = self.step_store
                else:
                    for key in self.global_store:
                        for i in range(len(self.global_store[key])):
                            self.global_store[key][i] += self.step_store[key][i].detach()
        self.step_store = self.get_empty_store()

    def get_average_attention(self):
        return self.attention_store

    def get_average_global_attention(self):
        return {key: [item / self.cur_step for item in self.global_store[key]] for key in self.attention_store}

    def reset(self):
        super().reset()
        self.step_store = self.get_empty_store()
        self.attention_store = {}
        self.global_store = {}


def stored_maps(attention_store: AttentionStore, res: int, from_where: List[str], is_cross: bool, select: int = 0):
    """The tensors `aggregate_attention` averages, in its order (reference :279-286): every stored map with res^2
    pixels of the listed locations."""
    maps = []
    attention_maps = attention_store.get_average_attention()
    num_pixels = res ** 2
    for location in from_where:
        for item in attention_maps[f"{location}_{'cross' if is_cross else 'self'}"]:
            if item.shape[1] == num_pixels:
                maps.append(item)
    if not maps:
        raise RuntimeError(f"no stored attention map has {res}x{res} pixels: nothing to aggregate "
                           "(reference: torch.cat of an empty list)")
    if select != 0:
        raise IndexError(f"index {select} is out of bounds for dimension 0 with size 1")
    return maps


def aggregate_attention(attention_store: AttentionStore, res: int, from_where: List[str], is_cross: bool,
                        select: int) -> torch.Tensor:
    """Mean over the heads of every stored map with res^2 pixels, in `from_where` order
    (reference :273-289).  Runs `ga_aggregate_maps`; returns float32 (res, res, n_keys) and stays
    differentiable w.r.t. the stored probabilities."""
    out = ops.AggregateMaps.apply(*stored_maps(attention_store, res, from_where, is_cross, select))
    return out.view(res, res, out.shape[-1])
