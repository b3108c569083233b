"""The oracle's denoising loop against the reference's own `__call__` (tests/golden/g9_loop.*:
reduced-width UNet of the real topology, the build's DDIM).  CPU only."""
import math

import numpy as np
import pytest
import torch

import hashrand
from conftest import load_json, load_npz
from oracle import loss as oloss
from oracle.pipeline import GuidedSampler

G9 = load_json("g9_loop.json")
BASE_ENTRIES = [{"index": 2, "kind": "BOX", "geom": (.6, .3, .4, .55), "subprompt": "robot"},
                {"index": 5, "kind": "BOX", "geom": (.2, .3, .4, .55), "subprompt": "blue vase"},
                {"index": 6, "kind": "BOX", "geom": (.2, .3, .4, .55), "subprompt": "blue vase"}]


def hash_init_(module, seed):
    with torch.no_grad():
        for pi, (name, p) in enumerate(module.named_parameters()):
            if name.endswith("bias"):
                p.zero_()
            elif p.dim() == 1:
                p.fill_(1.0)
            else:
                u = hashrand.uniform(tuple(p.shape), seed + pi) * np.float32(2.0) - np.float32(1.0)
                p.copy_(torch.from_numpy(u * np.float32(math.sqrt(3.0 / p[0].numel()))))
    return module


def g9_setup(meta):
    from guided_attention_amd.unet import UNet2DConditionModel, UNetConfig
    unet = hash_init_(UNet2DConditionModel(UNetConfig.tiny(sample_size=32, cross_attention_dim=48)), meta["unet_seed"]).float()
    for p in unet.parameters():
        p.requires_grad_(False)
    embeds = torch.from_numpy(hashrand.normalish((2, 77, 48), meta["embed_seed"]))
    lat0 = torch.from_numpy(hashrand.normalish((1, 4, 32, 32), meta["latent_seed"]))
    gen = torch.Generator("cpu").manual_seed(meta["renoise_seed"])
    noise = [torch.randn(lat0.shape, generator=gen) for _ in range(64)]
    thr = {int(k): v for k, v in meta["thresholds"].items()}
    return unet, embeds, lat0, noise, thr


@pytest.mark.parametrize("meta", G9, ids=lambda m: m["name"])
def test_oracle_loop_matches_reference(meta):
    g = load_npz("g9_loop.npz")
    unet, embeds, lat0, noise, thr = g9_setup(meta)
    plan = oloss.TokenPlan(BASE_ENTRIES, meta["hyper"])
    s = GuidedSampler(unet, plan, thresholds=thr, only_update_on_threshold_steps=meta["only_update_on_threshold_steps"],
                      max_iter_to_alter=meta["max_iter_to_alter"], steps=meta["steps"], scale_factor=meta["scale_factor"])
    torch.set_num_threads(1)
    out = s.sample(lat0, embeds, noise)
    n = meta["name"]
    assert (s.calls["fwd_b1_grad"], s.calls["bwd"], s.calls["fwd_b2"]) == (meta["fwd_b1"], meta["bwd"], meta["fwd_b2"])
    assert sum(1 for e in s.trace if e[0] == "refine") == meta["subiterations"]
    fin = [e[3] for e in s.trace if e[0] == "refine_final"]
    np.testing.assert_allclose(fin, g[f"{n}.refine_final_losses"], rtol=2e-3)
    ref = g[f"{n}.final_latents"]
    err = np.abs(out.numpy() - ref).max() / np.abs(ref).max()
    assert err < 2e-3, err
