#!/usr/bin/env python3
"""Validates the extrapolation bench.py uses for its `cpu_baseline` leg: ONE full 50-step guided image of the CPU
oracle (oracle/pipeline.py) is run end to end on the reduced-width UNet (`bench.py --model tiny` shapes: the SD-1.x
topology at 1/10 width, latent 64^2) and its wall time is compared with  sum_k calls_k * median seconds_k  from the
per-kind timings (1 warm-up + 3 repetitions each), the way the full-width number is obtained.  CPU only.
Prints one JSON line."""
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from guided_attention_amd.unet import UNet2DConditionModel, UNetConfig  # noqa: E402
from oracle import loss as oloss  # noqa: E402
from oracle.pipeline import GuidedSampler  # noqa: E402


def main():
    cores = bench.usable_cores()
    torch.set_num_threads(cores)
    cfg = UNetConfig.tiny(64, 768)
    unet = UNet2DConditionModel(cfg).init_weights_(seed=0).float()
    for p in unet.parameters():
        p.requires_grad_(False)
    entries = [{"index": 2, "kind": "BOX", "geom": (.6, .3, .4, .55), "subprompt": "robot"},
               {"index": 5, "kind": "BOX", "geom": (.2, .3, .4, .55), "subprompt": "blue vase"},
               {"index": 6, "kind": "BOX", "geom": (.2, .3, .4, .55), "subprompt": "blue vase"}]
    plan = oloss.TokenPlan(entries)
    g = torch.Generator("cpu").manual_seed(1234)
    embeds = torch.randn(2, 77, cfg.cross_attention_dim, generator=g)
    gs = torch.Generator("cpu").manual_seed(0)
    lat = torch.randn(1, 4, 64, 64, generator=gs)
    noise = [torch.randn(1, 4, 64, 64, generator=gs) for _ in range(100)]
    s = GuidedSampler(unet, plan, thresholds={0: 1.0}, steps=50)
    t0 = time.perf_counter()
    out = s.sample(lat, embeds, noise)
    full = time.perf_counter() - t0
    calls = dict(s.calls)
    # per-kind timings exactly as bench.cpu_baseline takes them
    s2 = GuidedSampler(unet, plan, thresholds={0: 1.0}, steps=50)
    box = {}

    def fwd():
        with torch.enable_grad():
            box["lat"], box["r"], _ = s2._evaluate(lat, 981, embeds[1:2])

    def bwd():
        with torch.enable_grad():
            s2._update(box["lat"], box["r"]["loss"], 20.0)

    def cfg_fwd():
        with torch.no_grad():
            unet(torch.cat([lat] * 2), 981, encoder_hidden_states=embeds)

    per = {"fwd_b1_grad": bench._timed(fwd, 3)[0], "bwd": bench._timed(bwd, 3)[0], "fwd_b2": bench._timed(cfg_fwd, 3)[0]}
    extrap = sum(per[k] * calls[k] for k in per)
    print(json.dumps({"model": "tiny (SD-1.x topology, 1/10 width, latent 64^2)", "cores": cores, "cpu_model": bench.cpu_model_name(),
                      "full_run_seconds": round(full, 2), "calls": calls,
                      "seconds_per_kind": {k: round(v, 4) for k, v in per.items()},
                      "extrapolated_seconds": round(extrap, 2), "extrapolation_over_measured": round(extrap / full, 3),
                      "finite": bool(torch.isfinite(out).all())}))


if __name__ == "__main__":
    main()
