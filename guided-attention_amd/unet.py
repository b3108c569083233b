"""Host-side UNet2DCondition for SD-1.x / SD-2.x shapes in plain PyTorch-ROCm (conv / GEMM / norm
on MIOpen, rocBLAS, hipBLASLt).  The reference delegates this to diffusers 0.12.1
(`UNet2DConditionModel`, called from pipeline_guided_attention.py:583-743), which is not vendored in
the reference checkout and not installed here; the topology and parameter names follow the published
diffusers layout so that a local `diffusion_pytorch_model.safetensors` loads by name.

Every attention layer dispatches to an *attention processor* with the reference's protocol
(utils/ptp_utils.py:66): proc(attn, hidden_states, encoder_hidden_states=None, attention_mask=None).
The default processor is the HIP one (no CPU fallback).
"""
import math
from dataclasses import dataclass, field
from types import SimpleNamespace
from typing import Optional, Tuple, Union

import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class UNetConfig:
    sample_size: int = 64
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    down_block_types: Tuple[str, ...] = ("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D",
                                         "DownBlock2D")
    up_block_types: Tuple[str, ...] = ("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D")
    layers_per_block: int = 2
    attention_head_dim: Union[int, Tuple[int, ...]] = 8  # diffusers naming: this is the number of heads
    cross_attention_dim: int = 768
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    use_linear_projection: bool = False
    center_input_sample: bool = False
    transformer_layers_per_block: Union[int, Tuple[int, ...]] = 1   # BasicTransformerBlocks per Transformer2DModel, per level
    addition_embed_type: Optional[str] = None                       # "text_time": SDXL's pooled-text + size/crop conditioning
    addition_time_embed_dim: int = 256
    projection_class_embeddings_input_dim: int = 2816

    @classmethod
    def sd15(cls):
        return cls()

    @classmethod
    def sd21(cls, sample_size=96):
        return cls(sample_size=sample_size, attention_head_dim=(5, 10, 20, 20), cross_attention_dim=1024,
                   use_linear_projection=True)

    @classmethod
    def sdxl(cls, sample_size=128):
        """SDXL-base UNet shapes (BASELINE config 5; no reference behaviour exists — diffusers 0.12.1 predates SDXL):
        3 levels 320/640/1280, no attention on the top level, 2 / 10 transformer blocks per attention on the lower two,
        head_dim 64 (5/10/20 heads), 2048-wide text context, linear projections, text_time added conditioning."""
        return cls(sample_size=sample_size, block_out_channels=(320, 640, 1280),
                   down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
                   up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
                   attention_head_dim=(5, 10, 20), cross_attention_dim=2048, use_linear_projection=True,
                   transformer_layers_per_block=(1, 2, 10), addition_embed_type="text_time")

    @classmethod
    def tiny_sdxl(cls, sample_size=32, cross_attention_dim=64):
        """The SDXL layout at reduced width / depth, for tests."""
        return cls(sample_size=sample_size, block_out_channels=(32, 64, 128),
                   down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
                   up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
                   attention_head_dim=(1, 2, 4), cross_attention_dim=cross_attention_dim, use_linear_projection=True,
                   transformer_layers_per_block=(1, 2, 3), addition_embed_type="text_time", addition_time_embed_dim=8,
                   projection_class_embeddings_input_dim=6 * 8 + 40)

    @classmethod
    def tiny(cls, sample_size=64, cross_attention_dim=64):
        """Full SD-1.x topology (4 resolutions, 16 cross + 16 self attention layers) at 1/10 width, for tests."""
        return cls(sample_size=sample_size, block_out_channels=(32, 64, 128, 128), attention_head_dim=2,
                   cross_attention_dim=cross_attention_dim)

    def get(self, key, default=None):
        return getattr(self, key, default)


class Attention(nn.Module):
    """The module an attention processor is handed (diffusers 0.12.1 `CrossAttention` surface)."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.upcast_attention = False
        self.upcast_softmax = False
        self.is_cross = cross_attention_dim is not None
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(cross_attention_dim or query_dim, inner, bias=False)
        self.to_v = nn.Linear(cross_attention_dim or query_dim, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(0.0)])
        self.processor = None  # None -> the HIP default, resolved lazily

    def set_processor(self, processor):
        self.processor = processor

    def prepare_attention_mask(self, attention_mask, target_length):
        return attention_mask

    def head_to_batch_dim(self, t):
        b, n, c = t.shape
        return t.reshape(b, n, self.heads, c // self.heads).permute(0, 2, 1, 3).reshape(b * self.heads, n, c // self.heads)

    def batch_to_head_dim(self, t):
        bh, n, d = t.shape
        return t.reshape(bh // self.heads, self.heads, n, d).permute(0, 2, 1, 3).reshape(bh // self.heads, n, d * self.heads)

    def _processor(self):
        proc = self.processor
        if proc is None:
            from .utils.ptp_utils import default_processor
            proc = self.processor = default_processor()
        return proc

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, folded=None):
        """folded (this build's transformer blocks, processors that declare `supports_folded_layer_norm`): see
        utils/ptp_utils.AttendExciteCrossAttnProcessor.__call__."""
        proc = self._processor()
        if folded is not None:
            return proc(self, hidden_states, encoder_hidden_states=encoder_hidden_states, attention_mask=attention_mask,
                        folded=folded)
        return proc(self, hidden_states, encoder_hidden_states=encoder_hidden_states, attention_mask=attention_mask)


# Measured in round 3 (profiles/r3_*), on since: Upsample2D as one launch (the patch gather reads the half-size map) and the
# GroupNorm backward adding the skip connection's gradient itself.  Module attributes, not environment switches: an A/B run
# sets them from its own script (tools/unet_bench.py).
_FUSE_UPSAMPLE = True
_GN_ALIAS = True


class GroupNormAct(nn.GroupNorm):
    """GroupNorm with an optional fused SiLU.  Same parameters / state_dict keys as nn.GroupNorm.  `impl` is a
    callable (x, weight, bias, groups, eps, act) -> y installed by the GPU pipeline (the channels-last HIP
    kernels, ops.group_norm_act); None = the portable PyTorch ops (what the CPU oracle runs)."""

    def __init__(self, num_groups, num_channels, eps=1e-5, act=False):
        super().__init__(num_groups, num_channels, eps=eps)
        self.act = act
        self.impl = None

    def forward(self, x, chan_bias=None, with_alias=False):
        """chan_bias (B, C), optional: normalise x + chan_bias[:, :, None, None] (folded into the kernels).
        with_alias: -> (y, x) where the returned x is what the block's skip connection must consume: with the HIP
        kernels and a differentiated input it is an alias produced by the norm's autograd node, so the skip connection's
        gradient is added inside the norm's backward kernel (ops.GroupNormAct); otherwise x itself."""
        if self.impl is not None:
            if with_alias and _GN_ALIAS and x.requires_grad and torch.is_grad_enabled():
                return self.impl(x, self.weight, self.bias, self.num_groups, self.eps, self.act, chan_bias, True)
            y = self.impl(x, self.weight, self.bias, self.num_groups, self.eps, self.act, chan_bias)
            return (y, x) if with_alias else y
        x_in = x
        if chan_bias is not None:
            x = x + chan_bias[:, :, None, None]
        y = super().forward(x)
        y = F.silu(y) if self.act else y
        return (y, x_in) if with_alias else y


def conv3x3(x, conv, impl, residual=None, with_bias=True, gn_for=None):
    """A 3x3 Conv2d (padding 1) of the UNet, optionally with its bias and a residual added in the same pass.  `impl` is
    the implicit-GEMM HIP kernel (ops.conv3x3, installed by the GPU pipeline for the 16-bit dtypes) or None = the library
    convolution (what the CPU oracle and the fp32 parity runs use).  gn_for = (groups, chan_bias) of the GroupNorm that
    consumes the result: the HIP kernel's epilogue then takes that norm's statistics (ops.conv3x3)."""
    bias = conv.bias if with_bias else None
    if impl is not None and impl.supported(x, conv.weight, conv.stride[0]):
        if gn_for is not None:
            return impl(x, conv.weight, bias, residual, conv.stride[0], gn_for=gn_for)
        return impl(x, conv.weight, bias, residual, conv.stride[0])
    y = F.conv2d(x, conv.weight, bias, stride=conv.stride, padding=1)
    return y if residual is None else y + residual


def pointwise_conv_tokens(x_tokens, conv):
    """A 1x1 Conv2d applied to a (B, HW, C) token view as ONE GEMM with the bias in its epilogue: on channels-last
    activations the token view is free, and the library GEMM beats the implicit-GEMM conv path at these sizes."""
    return F.linear(x_tokens, conv.weight.view(conv.out_channels, conv.in_channels), conv.bias)


def nchw_to_tokens(x):
    b, c, h, w = x.shape
    return x.permute(0, 2, 3, 1).reshape(b, h * w, c)  # a view when x is channels-last


def tokens_to_nchw(t, h, w):
    b, _, c = t.shape
    return t.reshape(b, h, w, c).permute(0, 3, 1, 2)   # channels-last strides, no copy


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    impl = None  # fused h * gelu(gate) (ops.geglu), installed by UNet2DConditionModel.set_fused_impl

    def forward(self, x):
        x = self.proj(x)
        if self.impl is not None and x.shape[-1] % 16 == 0:
            return self.impl(x)
        h, gate = x.chunk(2, dim=-1)
        return h * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * 4), nn.Dropout(0.0), nn.Linear(dim * 4, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, cross_attention_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, None, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, cross_attention_dim, heads, dim_head)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    ln_impl = None  # (layer_norm, add_layer_norm) from ops: residual add + next LayerNorm in one launch
    lin_impl = None  # fused_linear module: LayerNorm folded into the projections, residual adds into their epilogues

    def folds(self, x):
        """Whether this block takes the folded form for x: the module is installed, the dtype / widths are served and both
        attention processors understand the `folded` keyword (a foreign processor keeps the reference's protocol)."""
        lin = self.lin_impl
        return (lin is not None and lin.block_folds(x) and
                all(getattr(a._processor(), "supports_folded_layer_norm", False) for a in (self.attn1, self.attn2)))

    def forward_folded(self, x, xp, context, want_partials):
        """x: raw residual stream (B, N, C); xp: its row partial sums (from the producing GEMM).  Every LayerNorm is applied
        inside the projection that consumes it, every residual add inside the projection that produces the new stream:
        8 launches (6 GEMMs + 2 attention kernels) where the unfused block takes 14.  -> (x, partial sums or None)."""
        lin = self.lin_impl
        x, xp = self.attn1(x, folded={"partials": xp, "norm": self.norm1, "residual": x, "want_partials": True})
        x, xp = self.attn2(x, encoder_hidden_states=context,
                           folded={"partials": xp, "norm": self.norm2, "residual": x, "want_partials": True})
        proj, out = self.ff.net[0].proj, self.ff.net[2]
        h, x = lin.ln_linear(x, xp, self.norm3, proj.weight, proj.bias, geglu=True)
        return lin.linear(h, out.weight, out.bias, residual=x, want_partials=want_partials)

    def forward(self, x, context):
        if self.ln_impl is not None and x.shape[-1] % 8 == 0:
            ln, add_ln = self.ln_impl
            h = self.attn1(ln(x, self.norm1.weight, self.norm1.bias, self.norm1.eps))
            x, n = add_ln(h, x, self.norm2.weight, self.norm2.bias, self.norm2.eps)
            h = self.attn2(n, encoder_hidden_states=context)
            x, n = add_ln(h, x, self.norm3.weight, self.norm3.bias, self.norm3.eps)
            return self.ff(n) + x
        x = self.attn1(self.norm1(x)) + x
        x = self.attn2(self.norm2(x), encoder_hidden_states=context) + x
        return self.ff(self.norm3(x)) + x


class Transformer2DModel(nn.Module):
    def __init__(self, heads, dim_head, in_channels, cross_attention_dim, groups, use_linear_projection, depth=1):
        super().__init__()
        inner = heads * dim_head
        self.use_linear_projection = use_linear_projection
        self.norm = GroupNormAct(groups, in_channels, eps=1e-6, act=False)
        if use_linear_projection:
            self.proj_in = nn.Linear(in_channels, inner)
            self.proj_out = nn.Linear(inner, in_channels)
        else:
            self.proj_in = nn.Conv2d(in_channels, inner, 1)
            self.proj_out = nn.Conv2d(inner, in_channels, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(inner, heads, dim_head, cross_attention_dim)
                                                 for _ in range(depth)])

    feeds_norm = False   # set by the block that owns this layer when its result goes straight into a GroupNorm (DownBlock, UNet)

    def _proj_weights(self, proj):
        if self.use_linear_projection:
            return proj.weight, proj.bias
        return proj.weight.reshape(proj.out_channels, proj.in_channels), proj.bias

    def forward(self, x, context):
        b, c, h, w = x.shape
        x, res = self.norm(x, with_alias=True)   # res: the skip connection's view of the input (see GroupNormAct.forward)
        x = nchw_to_tokens(x)
        blocks = self.transformer_blocks
        lin = blocks[0].lin_impl
        if lin is not None and x.is_contiguous() and lin.supported(x, c, blocks[0].norm1.normalized_shape[0]) and \
                res.is_contiguous(memory_format=torch.channels_last) and all(blk.folds(x) for blk in blocks):
            # proj_in leaves the row sums the first LayerNorm needs; proj_out lands on the block's input (its residual)
            x, xp = lin.linear(x, *self._proj_weights(self.proj_in), want_partials=True)
            for i, blk in enumerate(blocks):
                x, xp = blk.forward_folded(x, xp, context, want_partials=i + 1 < len(blocks))
            # proj_out's epilogue also takes the GroupNorm statistics of the block's result for the norm that consumes it (the
            # next ResnetBlock's norm1, conv_norm_out: same group count as this block's own norm) where that norm takes two launches
            x, _ = lin.linear(x, *self._proj_weights(self.proj_out), residual=nchw_to_tokens(res),
                              gn_for=(self.norm.num_groups, h * w) if self.feeds_norm and self.norm.impl is not None else None)
            made = getattr(x, "_ga_gn_tokens", None)
            x = tokens_to_nchw(x, h, w)
            if made is not None:
                x._ga_gn = {"partials": made[0], "blocks": made[1], "groups": made[2], "chan_bias": None, "shape": tuple(x.shape)}
            return x
        x = self.proj_in(x) if self.use_linear_projection else pointwise_conv_tokens(x, self.proj_in)
        for blk in self.transformer_blocks:
            x = blk(x, context)
        x = self.proj_out(x) if self.use_linear_projection else pointwise_conv_tokens(x, self.proj_out)
        return tokens_to_nchw(x, h, w) + res


class ResnetBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, temb_channels, groups, eps):
        super().__init__()
        self.norm1 = GroupNormAct(groups, in_channels, eps=eps, act=True)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = GroupNormAct(groups, out_channels, eps=eps, act=True)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None

    add_impl = None   # fused conv2-bias + residual add (ops.bias_residual_add)
    conv_impl = None  # implicit-GEMM 3x3 convolution with bias / residual epilogue (ops.conv3x3)
    lin_impl = None   # fused_linear module: the 1x1 shortcut as ga_linear_fused

    def forward(self, x, temb_act):
        """temb_act = SiLU(time embedding), computed once per UNet forward; or the dict the UNet prepared with this
        block's time_emb_proj(temb_act) already evaluated (all blocks in one GEMM)."""
        # norm1 / norm2 carry the SiLU.  conv1's bias rides on the time projection (one add instead of two): the
        # UNet's batched projection already contains it; the stand-alone path adds it here
        h, x = self.norm1(x, with_alias=True)   # x: the skip connection's view of the input (see GroupNormAct.forward)
        if isinstance(temb_act, dict):
            tproj = temb_act[id(self)]
        else:
            tproj = self.time_emb_proj(temb_act) + self.conv1.bias
        # conv1's epilogue takes norm2's statistics (of its result + the time term) where norm2 would need a launch for them;
        # conv2's does the same for whatever norm reads this block's output next (the attention's GroupNorm, conv_norm_out)
        gn = self.norm2.impl is not None
        h = conv3x3(h, self.conv1, self.conv_impl, with_bias=False, gn_for=(self.norm2.num_groups, tproj) if gn else None)
        h = self.norm2(h, chan_bias=tproj)  # the time term is added inside the norm's loads
        if self.conv_shortcut is not None:
            _, _, hh, ww = x.shape
            xt = nchw_to_tokens(x)
            sc = self.conv_shortcut
            if self.lin_impl is not None and xt.is_contiguous() and self.lin_impl.supported(xt, sc.in_channels, sc.out_channels):
                xt, _ = self.lin_impl.linear(xt, sc.weight.reshape(sc.out_channels, sc.in_channels), sc.bias)
            else:
                xt = pointwise_conv_tokens(xt, sc)
            x = tokens_to_nchw(xt, hh, ww)
        if self.conv_impl is not None and self.conv_impl.supported(h, self.conv2.weight, 1):
            # bias + skip connection in the epilogue
            if gn:
                return self.conv_impl(h, self.conv2.weight, self.conv2.bias, x, 1, gn_for=(self.norm2.num_groups, None))
            return self.conv_impl(h, self.conv2.weight, self.conv2.bias, x, 1)
        if self.add_impl is not None and self.conv2.out_channels % 8 == 0:
            # conv2's bias and the skip connection in one pass (the library conv adds its bias as a separate kernel)
            return self.add_impl(F.conv2d(h, self.conv2.weight, None, padding=1), self.conv2.bias, x)
        return x + self.conv2(h)


class Downsample2D(nn.Module):
    conv_impl = None

    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, stride=2, padding=1)

    def forward(self, x):
        return conv3x3(x, self.conv, self.conv_impl)


class Upsample2D(nn.Module):
    conv_impl = None

    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)

    def forward(self, x, output_size=None):
        impl = self.conv_impl
        if output_size is None and impl is not None and getattr(impl, "upsample", None) is not None and _FUSE_UPSAMPLE and \
                impl.supported(x, self.conv.weight, 1):
            return impl.upsample(x, self.conv.weight, self.conv.bias)   # the up-sampling happens in the patch gather
        if output_size is None:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        else:
            x = F.interpolate(x, size=output_size, mode="nearest")
        return conv3x3(x, self.conv, self.conv_impl)


class DownBlock(nn.Module):
    def __init__(self, cfg, in_c, out_c, temb_c, heads, add_downsample, has_attn, depth=1):
        super().__init__()
        self.has_cross_attention = has_attn
        self.resnets = nn.ModuleList([ResnetBlock2D(in_c if i == 0 else out_c, out_c, temb_c, cfg.norm_num_groups,
                                                    cfg.norm_eps) for i in range(cfg.layers_per_block)])
        if has_attn:
            self.attentions = nn.ModuleList([Transformer2DModel(heads, out_c // heads, out_c, cfg.cross_attention_dim,
                                                                cfg.norm_num_groups, cfg.use_linear_projection, depth)
                                             for _ in range(cfg.layers_per_block)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_c)]) if add_downsample else None
        if has_attn:
            for attn in self.attentions[:-1]:   # the next ResnetBlock's norm1 reads this layer's result directly
                attn.feeds_norm = True

    def forward(self, x, temb_act, context):
        outs = []
        for i, resnet in enumerate(self.resnets):
            x = resnet(x, temb_act)
            if self.has_cross_attention:
                x = self.attentions[i](x, context)
            outs.append(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs.append(x)
        return x, outs


class MidBlock(nn.Module):
    def __init__(self, cfg, channels, temb_c, heads, depth=1):
        super().__init__()
        self.has_cross_attention = True
        self.resnets = nn.ModuleList([ResnetBlock2D(channels, channels, temb_c, cfg.norm_num_groups, cfg.norm_eps)
                                      for _ in range(2)])
        self.attentions = nn.ModuleList([Transformer2DModel(heads, channels // heads, channels, cfg.cross_attention_dim,
                                                            cfg.norm_num_groups, cfg.use_linear_projection, depth)])

        self.attentions[0].feeds_norm = True   # resnets[1].norm1 reads it directly

    def forward(self, x, temb_act, context):
        x = self.resnets[0](x, temb_act)
        x = self.attentions[0](x, context)
        return self.resnets[1](x, temb_act)


class UpBlock(nn.Module):
    def __init__(self, cfg, in_c, out_c, prev_c, temb_c, heads, add_upsample, has_attn, depth=1):
        super().__init__()
        self.has_cross_attention = has_attn
        n = cfg.layers_per_block + 1
        resnets = []
        for i in range(n):
            skip_c = in_c if i == n - 1 else out_c
            res_in = prev_c if i == 0 else out_c
            resnets.append(ResnetBlock2D(res_in + skip_c, out_c, temb_c, cfg.norm_num_groups, cfg.norm_eps))
        self.resnets = nn.ModuleList(resnets)
        if has_attn:
            self.attentions = nn.ModuleList([Transformer2DModel(heads, out_c // heads, out_c, cfg.cross_attention_dim,
                                                                cfg.norm_num_groups, cfg.use_linear_projection, depth)
                                             for _ in range(n)])
        self.upsamplers = nn.ModuleList([Upsample2D(out_c)]) if add_upsample else None

    cat_impl = None   # ops.cat_channels: the concatenation with the skip connection as one HIP launch (set_fused_impl)

    def forward(self, x, skips, temb_act, context, upsample_size=None, n_layers=None):
        """`skips` is consumed from its end.  n_layers < len(resnets) stops early (truncated guidance
        forward) and returns before the upsampler."""
        for i, resnet in enumerate(self.resnets):
            if n_layers is not None and i >= n_layers:
                return x
            skip = skips.pop()
            if self.cat_impl is not None:   # the launch also takes norm1's statistics where that saves norm1 a launch
                n1 = resnet.norm1
                if n1.impl is not None:
                    x = self.cat_impl(x, skip, gn_for=n1.num_groups, norm=(n1.weight, n1.bias, n1.eps, n1.act))
                else:
                    x = self.cat_impl(x, skip)
                x = resnet(x, temb_act)
            else:
                x = resnet(torch.cat([x, skip], dim=1), temb_act)
            if self.has_cross_attention:
                x = self.attentions[i](x, context)
        if n_layers is None and self.upsamplers is not None:
            x = self.upsamplers[0](x, upsample_size)
        return x


def timestep_embedding(timesteps, dim):
    """diffusers `Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)`: [cos | sin], f32."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=timesteps.device) / half
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_c, dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_c, dim)
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


@dataclass
class UNetOutput:
    sample: Optional[torch.Tensor]


class UNet2DConditionModel(nn.Module):
    def __init__(self, config: Optional[UNetConfig] = None):
        super().__init__()
        cfg = self.config = config or UNetConfig()
        ch = cfg.block_out_channels
        heads = cfg.attention_head_dim if isinstance(cfg.attention_head_dim, (tuple, list)) else (cfg.attention_head_dim,) * len(ch)
        tl = cfg.transformer_layers_per_block
        depth = tuple(tl) if isinstance(tl, (tuple, list)) else (tl,) * len(ch)
        temb_c = ch[0] * 4
        self.in_channels = cfg.in_channels
        self.conv_in = nn.Conv2d(cfg.in_channels, ch[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(ch[0], temb_c)
        self.down_blocks = nn.ModuleList()
        out_c = ch[0]
        for i, kind in enumerate(cfg.down_block_types):
            in_c, out_c = out_c, ch[i]
            self.down_blocks.append(DownBlock(cfg, in_c, out_c, temb_c, heads[i], i != len(ch) - 1,
                                              kind == "CrossAttnDownBlock2D", depth[i]))
        self.mid_block = MidBlock(cfg, ch[-1], temb_c, heads[-1], depth[-1])
        self.up_blocks = nn.ModuleList()
        rev, rev_heads, rev_depth = list(reversed(ch)), list(reversed(heads)), list(reversed(depth))
        out_c = rev[0]
        for i, kind in enumerate(cfg.up_block_types):
            prev_c, out_c = out_c, rev[i]
            in_c = rev[min(i + 1, len(ch) - 1)]
            self.up_blocks.append(UpBlock(cfg, in_c, out_c, prev_c, temb_c, rev_heads[i], i != len(ch) - 1,
                                          kind == "CrossAttnUpBlock2D", rev_depth[i]))
        self.num_upsamplers = len(ch) - 1
        self.add_embedding = None
        if cfg.addition_embed_type == "text_time":   # SDXL: emb += add_embedding([pooled text | sinusoid(time_ids)])
            self.add_embedding = TimestepEmbedding(cfg.projection_class_embeddings_input_dim, temb_c)
        elif cfg.addition_embed_type is not None:
            raise ValueError(f"addition_embed_type {cfg.addition_embed_type!r} is not supported")
        self._added = None      # (2, temb_c) rows [uncond, cond] of add_embedding(...), set by set_added_cond
        self._added_version = 0
        self.conv_norm_out = GroupNormAct(cfg.norm_num_groups, ch[0], eps=cfg.norm_eps, act=True)
        self.conv_out = nn.Conv2d(ch[0], cfg.out_channels, 3, padding=1)
        last = self.up_blocks[-1]
        if getattr(last, "has_cross_attention", False):
            last.attentions[-1].feeds_norm = True      # conv_norm_out reads it directly

    # ---- attention-processor registry (diffusers protocol used by utils/ptp_utils.py:149-175)
    def _attention_modules(self):
        """(name, module) of every attention layer, in registration order (the module tree is fixed after construction;
        register_attention_control asks twice per image)."""
        mods = self.__dict__.get("_attn_mods")
        if mods is None:
            mods = self.__dict__["_attn_mods"] = [(n, m) for n, m in self.named_modules() if isinstance(m, Attention)]
        return mods

    @property
    def attn_processors(self):
        return {f"{name}.processor": mod.processor for name, mod in self._attention_modules()}

    def set_attn_processor(self, processor):
        mods = dict(self._attention_modules())
        if isinstance(processor, dict):
            if len(processor) != len(mods):
                raise ValueError(f"A dict of processors was passed, but the number of processors {len(processor)} does "
                                 f"not match the number of attention layers: {len(mods)}.")
            for name, mod in mods.items():
                mod.set_processor(processor[f"{name}.processor"])
        else:
            for mod in mods.values():
                mod.set_processor(processor)

    def _time_projections(self, temb_act):
        """Every ResnetBlock2D adds time_emb_proj(SiLU(temb)): 22 GEMMs with one row of input each.  They are
        evaluated as ONE GEMM against the row-concatenated weights (built once; rebuilt if a weight changes) and
        handed to the blocks as views."""
        blocks = self._resnet_blocks()
        key = tuple((b.time_emb_proj.weight.data_ptr(), b.time_emb_proj.weight._version, b.time_emb_proj.bias._version,
                     b.conv1.bias._version) for b in blocks)
        cache = self.__dict__.get("_tproj_cache")
        if cache is None or cache[0] != key:
            with torch.no_grad():
                w = torch.cat([b.time_emb_proj.weight for b in blocks], dim=0)
                bias = torch.cat([b.time_emb_proj.bias + b.conv1.bias for b in blocks], dim=0)  # conv1's bias rides along
            cache = (key, w, bias, [b.time_emb_proj.out_features for b in blocks])
            self.__dict__["_tproj_cache"] = cache
        _, w, bias, sizes = cache
        parts = F.linear(temb_act, w, bias).split(sizes, dim=1)
        return {id(b): p for b, p in zip(blocks, parts)}

    # rows of the [uncond, cond] added conditioning each batch layout of the pipeline uses: guidance pass (cond),
    # CFG pass (uncond, cond), joint pass (cond | uncond, cond)
    ADDED_ROWS = {1: (1,), 2: (0, 1), 3: (1, 0, 1)}

    def set_added_cond(self, text_embeds, time_ids):
        """SDXL added conditioning (`added_cond_kwargs` of the published UNet): text_embeds (2, P) pooled text
        embeddings and time_ids (2, 6) = (orig h, w, crop top, left, target h, w) for [uncond, cond].  They are
        constant over an image, so add_embedding(...) is evaluated once here and folded into the time projection."""
        if self.add_embedding is None:
            raise ValueError("this UNet has no added conditioning (addition_embed_type is None)")
        with torch.no_grad():
            ids = timestep_embedding(time_ids.reshape(-1).to(self.device), self.config.addition_time_embed_dim)
            aug = torch.cat([text_embeds.to(self.device, torch.float32), ids.reshape(time_ids.shape[0], -1)], dim=-1)
            if aug.shape != (2, self.config.projection_class_embeddings_input_dim):
                raise ValueError(f"added conditioning has shape {tuple(aug.shape)}, expected "
                                 f"(2, {self.config.projection_class_embeddings_input_dim})")
            self._added = self.add_embedding(aug.to(self.dtype))
        self._added_version += 1

    def _added_rows(self, batch):
        if self.add_embedding is None:
            return None
        if self._added is None:
            raise ValueError("set_added_cond(text_embeds, time_ids) must be called before running this UNet")
        if batch not in self.ADDED_ROWS:
            raise ValueError(f"no added-conditioning layout for batch {batch}")
        return self._added[list(self.ADDED_ROWS[batch])]

    def time_projection(self, timestep, batch):
        """Everything the UNet derives from the timestep alone — sinusoid, the 2-layer time MLP, SiLU and the 22
        per-block projections (with conv1's bias) — as ONE flat buffer laid out block-major, [block][batch][C_block],
        so that every ResnetBlock2D gets a contiguous (batch, C) view.  It depends on the timestep and on frozen
        weights only, so it is computed once per (timestep, batch) and kept (the reference recomputes ~15 small
        kernels per UNet call; with batch > 1 the per-block column slices of a row-major projection also cost one
        copy each).  Inference-time only: the cache is bypassed while any time-embedding weight requires grad."""
        # this runs on the host before EVERY hipGraph replay: the module walk (662 modules) and the 22-block key used to
        # cost ~0.5 ms per call — a device-idle gap of that length at the start of each of the ~120 passes of an image
        blocks = self._resnet_blocks()
        wkey = tuple((p.data_ptr(), p._version) for p in self._time_params())
        cache = self.__dict__.setdefault("_tp_cache", {"key": None, "items": {}})
        if cache["key"] != wkey:
            cache["key"], cache["items"] = wkey, {}
        k = (float(timestep), int(batch), self.dtype, self._added_version)
        flat = cache["items"].get(k)
        if flat is None:
            with torch.no_grad():
                t = torch.tensor([timestep], dtype=torch.int64 if float(timestep).is_integer() else torch.float64,
                                 device=self.device)
                t_emb = timestep_embedding(t, self.config.block_out_channels[0]).to(self.dtype)
                emb = self.time_embedding(t_emb)
                added = self._added_rows(batch)
                if added is not None:
                    emb = emb + added          # (batch, temb_c): per-sample added conditioning
                parts = self._time_projections(F.silu(emb))
                flat = torch.cat([parts[id(b)].expand(batch, -1).reshape(-1) for b in blocks])
            cache["items"][k] = flat
        return flat

    def _resnet_blocks(self):
        """The ResnetBlock2D modules in registration order (the module tree is fixed after construction)."""
        blocks = self.__dict__.get("_resnets")
        if blocks is None:
            blocks = self.__dict__["_resnets"] = [m for m in self.modules() if isinstance(m, ResnetBlock2D)]
        return blocks

    def _time_params(self):
        """Every parameter the timestep-only part of the UNet reads (their storage and version key its cache)."""
        ps = self.__dict__.get("_time_ps")
        if ps is None:
            ps = [self.time_embedding.linear_1.weight, self.time_embedding.linear_1.bias, self.time_embedding.linear_2.weight,
                  self.time_embedding.linear_2.bias]
            for b in self._resnet_blocks():
                ps += [b.time_emb_proj.weight, b.time_emb_proj.bias, b.conv1.bias]
            self.__dict__["_time_ps"] = ps
        return ps

    def _split_time_projection(self, flat, batch):
        blocks = self._resnet_blocks()
        want = batch * sum(b.time_emb_proj.out_features for b in blocks)
        if flat.numel() != want:
            raise ValueError(f"time projection buffer has {flat.numel()} elements, expected {want} for batch {batch}")
        out, off = {}, 0
        for b in blocks:
            c = b.time_emb_proj.out_features
            out[id(b)] = flat[off:off + batch * c].view(batch, c)
            off += batch * c
        return out

    def set_norm_impl(self, impl):
        """Install (or with None remove) the GroupNorm(+SiLU) implementation of every norm layer."""
        for m in self.modules():
            if isinstance(m, GroupNormAct):
                m.impl = impl

    def set_fused_impl(self, geglu=None, bias_residual_add=None, layer_norms=None, conv=None, linear=None, cat=None):
        """Install (or with None remove) the fused element-wise epilogues: GEGLU, conv-bias + residual, and
        (layer_norm, add_layer_norm) for the transformer blocks; `conv` = the implicit-GEMM 3x3 convolution; `linear` = the
        fused_linear module (LayerNorm / GEGLU / residual folded into the transformer blocks' GEMMs, 1x1 shortcuts); `cat` = the
        UpBlocks' channel concatenation (ops.cat_channels)."""
        for m in self.modules():
            if isinstance(m, (Downsample2D, Upsample2D)):
                m.conv_impl = conv
            if isinstance(m, GEGLU):
                m.impl = geglu
            elif isinstance(m, ResnetBlock2D):
                m.add_impl = bias_residual_add
                m.conv_impl = conv
                m.lin_impl = linear
            elif isinstance(m, BasicTransformerBlock):
                m.ln_impl = layer_norms
                m.lin_impl = linear
            elif isinstance(m, UpBlock):
                m.cat_impl = cat

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    @property
    def device(self):
        return self.conv_in.weight.device

    def forward(self, sample, timestep, encoder_hidden_states, class_labels=None, attention_mask=None,
                cross_attention_kwargs=None, return_dict=True, stop_after_up_block=None, time_projection=None):
        """stop_after_up_block = (i, n): run up_blocks[:i] fully and only the first n layers of up_blocks[i],
        then return sample=None — everything after the last attention map the loss consumes is skipped.
        time_projection: the flat buffer of `time_projection(timestep, batch)` (then `timestep` is not read); a Python
        number as `timestep` uses the cached buffer by itself when the time-embedding weights are frozen."""
        cfg = self.config
        up_factor = 2 ** self.num_upsamplers
        forward_upsample_size = any(s % up_factor != 0 for s in sample.shape[-2:])
        if cfg.center_input_sample:
            sample = 2 * sample - 1.0
        if time_projection is None and not torch.is_tensor(timestep) and sample.is_cuda and \
                not self.time_embedding.linear_1.weight.requires_grad:
            time_projection = self.time_projection(timestep, sample.shape[0])
        if time_projection is not None:
            temb_act = self._split_time_projection(time_projection, sample.shape[0])
        else:
            if not torch.is_tensor(timestep):
                timestep = torch.tensor([timestep], dtype=torch.int64 if isinstance(timestep, int) else torch.float64,
                                        device=sample.device)
            elif timestep.dim() == 0:
                timestep = timestep[None].to(sample.device)
            timestep = timestep.expand(sample.shape[0])
            t_emb = timestep_embedding(timestep, cfg.block_out_channels[0]).to(self.dtype)
            emb = self.time_embedding(t_emb)
            added = self._added_rows(sample.shape[0])
            if added is not None:
                emb = emb + added
            temb_act = self._time_projections(F.silu(emb))

        x = self._edge_conv(self.conv_in, sample)
        skips = [x]
        for blk in self.down_blocks:
            x, outs = blk(x, temb_act, encoder_hidden_states)
            skips.extend(outs)
        x = self.mid_block(x, temb_act, encoder_hidden_states)
        for i, blk in enumerate(self.up_blocks):
            n_res = len(blk.resnets)
            upsample_size = None
            if i != len(self.up_blocks) - 1 and forward_upsample_size:
                upsample_size = skips[-n_res - 1].shape[2:]
            if stop_after_up_block is not None and i == stop_after_up_block[0]:
                blk(x, skips, temb_act, encoder_hidden_states, upsample_size, n_layers=stop_after_up_block[1])
                return UNetOutput(sample=None) if return_dict else (None,)
            x = blk(x, skips, temb_act, encoder_hidden_states, upsample_size)
        x = self._edge_conv(self.conv_out, self.conv_norm_out(x))
        return UNetOutput(sample=x) if return_dict else (x,)

    # conv_in / conv_out (4 channels on one side): ops.conv3x3_thin_apply (csrc/thin_conv.hip), installed by the GPU pipeline
    # for the 16-bit dtypes; None, or a shape it does not serve (map width not a multiple of 16): the library convolution.
    edge_conv_impl = None

    def _edge_conv(self, conv, x):
        impl = self.edge_conv_impl
        if impl is not None and impl.supported(x, conv.weight, conv.stride[0]):
            return impl(x, conv.weight, conv.bias)
        return conv(x)

    # ---- weights
    def init_weights_(self, seed=0):
        """Seeded random init for runs without a checkpoint (there is no network here): PyTorch's default
        layer init drawn from one CPU generator, so every rank/device builds identical weights."""
        g = torch.Generator(device="cpu").manual_seed(seed)
        with torch.no_grad():
            for name, p in self.named_parameters():
                if name.endswith("bias"):
                    p.zero_()
                elif p.dim() == 1:
                    p.fill_(1.0)
                else:
                    fan_in = p[0].numel()
                    bound = math.sqrt(3.0 / fan_in)  # unit-gain uniform: activations keep O(1) scale in fp16
                    p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * bound)
        return self

    def load_diffusers_state(self, state):
        missing, unexpected = self.load_state_dict(state, strict=False)
        if missing or unexpected:
            raise RuntimeError(f"UNet checkpoint mismatch: missing {missing[:5]}..., unexpected {unexpected[:5]}...")
        return self
