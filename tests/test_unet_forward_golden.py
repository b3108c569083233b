"""§8 a14 — the top-level UNet forward, pinned to the reference's OWN forward.

tests/golden/g11_unet_forward.* holds what `GuidedAttention.forward` of the reference
(pipeline_guided_attention.py:583-743) returned when it was driven over this build's blocks
(tests/golden/make_golden.py:_DiffusersFacade): time embedding, conv_in, the down / mid / up sequence with the
skip-tuple slicing, `forward_upsample_size`, `center_input_sample`, norm / act / conv_out.  Block arithmetic itself is
diffusers' (absent from the reference checkout: unpinned).

  * CPU (`-m "not gpu"`): `guided_attention_amd.unet.UNet2DConditionModel.forward` with the oracle's attention
    processors must reproduce the fixture; deliberately mis-wired forwards (skip order, upsample_size) must NOT.
  * GPU (`-m gpu`): the same UNet with the HIP kernels installed (fp32), through the product's default processors.
"""
import math

import numpy as np
import pytest
import torch

import hashrand
from conftest import load_json, load_npz

G11 = load_json("g11_unet_forward.json")


def build_unet(meta):
    from guided_attention_amd.unet import UNet2DConditionModel, UNetConfig
    c = meta["config"]
    hd = c["attention_head_dim"]
    cfg = UNetConfig(sample_size=c["sample_size"], block_out_channels=tuple(c["block_out_channels"]),
                     attention_head_dim=hd if isinstance(hd, int) else tuple(hd),
                     cross_attention_dim=c["cross_attention_dim"], use_linear_projection=c["use_linear_projection"],
                     center_input_sample=c["center_input_sample"])
    unet = UNet2DConditionModel(cfg).float()
    seed = meta["seed"]
    with torch.no_grad():   # make_golden.hash_init_ + the non-zero biases of g11
        for pi, (name, p) in enumerate(unet.named_parameters()):
            if name.endswith("bias"):
                p.copy_(torch.from_numpy(hashrand.normalish(tuple(p.shape), seed + 5000 + pi) * np.float32(0.05)))
            elif p.dim() == 1:
                p.fill_(1.0)
            else:
                u = hashrand.uniform(tuple(p.shape), seed + pi) * np.float32(2.0) - np.float32(1.0)
                p.copy_(torch.from_numpy(u * np.float32(math.sqrt(3.0 / p[0].numel()))))
    for p in unet.parameters():
        p.requires_grad_(False)
    B, _, H, W = meta["shape"]
    x = torch.from_numpy(hashrand.normalish((B, 4, H, W), seed + 1))
    ctx = torch.from_numpy(hashrand.normalish((B, 77, c["cross_attention_dim"]), seed + 2))
    return unet, x, ctx


def rel_err(got, ref):
    return float(np.abs(got - ref).max() / np.abs(ref).max())


def cpu_forward(unet, x, t, ctx):
    from oracle.attention import OracleStore
    from oracle.pipeline import install_processors
    install_processors(unet, OracleStore())
    with torch.no_grad():
        return unet(x, t, encoder_hidden_states=ctx).sample.numpy()


@pytest.mark.parametrize("meta", G11, ids=lambda m: m["name"])
def test_unet_forward_matches_reference_forward_cpu(meta):
    unet, x, ctx = build_unet(meta)
    ref = load_npz("g11_unet_forward.npz")[f"{meta['name']}.out"]
    got = cpu_forward(unet, x, meta["timestep"], ctx)
    assert got.shape == ref.shape
    assert rel_err(got, ref) < 2e-5, rel_err(got, ref)
    got_t = cpu_forward(unet, x, torch.tensor(meta["timestep"]), ctx)   # 0-d tensor timestep (reference :641-642)
    assert rel_err(got_t, ref) < 2e-5


def test_miswired_forwards_are_caught():
    """The fixture must be able to fail: three deliberate wiring bugs, each has to move the output far outside the
    tolerance of the test above."""
    import guided_attention_amd.unet as U
    meta = next(m for m in G11 if m["name"] == "tiny_36x28_b2")
    ref = load_npz("g11_unet_forward.npz")[f"{meta['name']}.out"]
    unet, x, ctx = build_unet(meta)

    # (1) skips consumed in the wrong order inside an up block (front instead of end; same shapes at equal widths)
    orig_up = U.UpBlock.forward

    def wrong_order(self, x_, skips, temb_act, context, upsample_size=None, n_layers=None):
        n = len(self.resnets)
        mine = skips[-n:]
        del skips[-n:]
        fixed = list(reversed(mine))            # what pop() will now hand out: first-stored first
        same_width = len({t.shape[1] for t in fixed}) == 1
        return orig_up(self, x_, fixed if same_width else list(mine), temb_act, context, upsample_size, n_layers)

    U.UpBlock.forward = wrong_order
    try:
        assert rel_err(cpu_forward(unet, x, meta["timestep"], ctx), ref) > 1e-2
    finally:
        U.UpBlock.forward = orig_up

    # (2) `upsample_size` not forwarded: 2x nearest up-sampling of a 5x4 map gives 10x8, the skip is 9x7 -> cat fails
    orig_upsample = U.Upsample2D.forward
    U.Upsample2D.forward = lambda self, x_, output_size=None: orig_upsample(self, x_, None)
    try:
        with pytest.raises(RuntimeError):
            cpu_forward(unet, x, meta["timestep"], ctx)
    finally:
        U.Upsample2D.forward = orig_upsample

    # (3) the time embedding of another timestep
    assert rel_err(cpu_forward(unet, x, meta["timestep"] + 20, ctx), ref) > 1e-3

    # and the unmodified forward still matches
    assert rel_err(cpu_forward(unet, x, meta["timestep"], ctx), ref) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("meta", G11, ids=lambda m: m["name"])
def test_hip_unet_forward_matches_reference_forward(meta):
    """The HIP-installed UNet (channels-last, fused GroupNorm / GEGLU / LayerNorm kernels, capture + flash attention
    kernels through the product's default processors, cached time projections) against the reference's forward."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from guided_attention_amd.pipeline_guided_attention import GuidedAttention
    unet, x, ctx = build_unet(meta)
    ref = load_npz("g11_unet_forward.npz")[f"{meta['name']}.out"]
    pipe = GuidedAttention(unet).to("cuda", torch.float32)
    with torch.no_grad():
        got = pipe.unet(x.cuda(), meta["timestep"], encoder_hidden_states=ctx.cuda()).sample
        got_t = pipe.unet(x.cuda(), torch.tensor(meta["timestep"]).cuda(), encoder_hidden_states=ctx.cuda()).sample
    assert rel_err(got.float().cpu().numpy(), ref) < 1e-4, rel_err(got.float().cpu().numpy(), ref)
    assert rel_err(got_t.float().cpu().numpy(), ref) < 1e-4
    # with autograd on (guidance pass): same values, and the flash / capture backward kernels run to the latents
    xg = x.cuda().requires_grad_(True)
    out = pipe.unet(xg, meta["timestep"], encoder_hidden_states=ctx.cuda()).sample
    assert rel_err(out.detach().float().cpu().numpy(), ref) < 1e-4
    (g,) = torch.autograd.grad(out.square().sum(), [xg])
    assert torch.isfinite(g).all()
