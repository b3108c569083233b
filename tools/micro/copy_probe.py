#!/usr/bin/env python3
"""Which framework (non-ga) kernels does one UNet pass still launch, and from where?  One EAGER batch-B forward (and, with `grad`,
the backward to the latents) of the SD-1.x UNet under torch.profiler with Python stacks: every aten op that launched a device
kernel, grouped by (op, input shapes, innermost frames inside this package).   usage: copy_probe.py [B] [grad]"""
import sys
from collections import Counter
from pathlib import Path

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from guided_attention_amd.pipeline_guided_attention import GuidedAttention  # noqa: E402
from guided_attention_amd.text import SyntheticTextEncoder, WordTokenizer  # noqa: E402
from guided_attention_amd.unet import UNet2DConditionModel, UNetConfig  # noqa: E402


def main():
    B = next((int(a) for a in sys.argv[1:] if a.isdigit()), 3)
    grad = "grad" in sys.argv
    with torch.device("cuda"):
        unet = UNet2DConditionModel(UNetConfig.sd15()).half()
    pipe = GuidedAttention(unet, None, None, SyntheticTextEncoder(768), WordTokenizer()).to("cuda", torch.float16)
    emb = torch.randn(B, 77, 768, device="cuda", dtype=torch.half)
    lat = torch.randn(B, 4, 64, 64, device="cuda", dtype=torch.half, requires_grad=grad)

    def one():
        with torch.set_grad_enabled(grad):
            out = pipe.unet(lat, 500, encoder_hidden_states=emb).sample
            if grad:
                out.float().square().mean().backward()

    one()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        one()
        torch.cuda.synchronize()
    seen = Counter()
    dur = Counter()
    for e in prof.events():
        if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
            continue
        if not e.name.startswith("aten::"):
            continue
        frames = [f for f in (e.stack or []) if "guided-attention_amd" in f or "guided_attention_amd" in f][:3]
        key = (e.name, str(e.input_shapes)[:90], " <- ".join(f.split("guided")[-1][-60:] for f in frames))
        seen[key] += 1
        dur[key] += sum(k.duration for k in e.kernels)
    print(f"B={B} grad={grad}: aten ops that launched kernels in one pass")
    for key, n in sorted(seen.items(), key=lambda kv: -dur[kv[0]]):
        print(f"{n:4d} x {dur[key]:8.1f} us  {key[0]:28s} {key[1]:92s} {key[2]}")


if __name__ == "__main__":
    main()
