#!/usr/bin/env python3
"""GPU time of the three captured passes of the guided-attention loop for the SD-1.x UNet (hipGraph replay,
so host launch cost is excluded): guidance forward + loss (B=1, autograd), its backward to the latents, and
the CFG forward (B=2), the batch-3 joint pass.  usage: unet_bench.py [truncated] [nhwc-off] [benchmark] [only=eval|grad|cfg|joint]
A/B arms (same box, one call): [no-gn-producer] the convolutions' epilogues leave no GroupNorm statistics (every large-level norm
takes its own statistics launch, as before round 4); [no-cat-norm] the small-map UpBlock concatenations as launches of their own; [no-stream] the Linear layers never take the persistent stream form."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from guided_attention_amd import run  # noqa: E402
from guided_attention_amd.config import RunConfig  # noqa: E402
from guided_attention_amd.graphs import GraphRunner  # noqa: E402
from guided_attention_amd.pipeline_guided_attention import GuidedAttention  # noqa: E402
from guided_attention_amd.text import SyntheticTextEncoder, WordTokenizer  # noqa: E402
from guided_attention_amd.unet import UNet2DConditionModel, UNetConfig  # noqa: E402
from guided_attention_amd.utils import ptp_utils, shared_state as state  # noqa: E402


def replay_ms(graph, n=10):
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    from guided_attention_amd import ops
    if "no-gn-producer" in sys.argv:
        ops.gn_two_launch = lambda *a, **k: False
    if "no-cat-norm" in sys.argv:    # the 16 x 16 / 8 x 8 UpBlock concatenations as their own launch in front of resnet.norm1
        ops.gn_fused_with_cat = lambda *a, **k: False
    if "no-stream" in sys.argv:
        ops.linear_stream_serves = lambda *a, **k: False
    if "benchmark" in sys.argv:
        torch.backends.cudnn.benchmark = True  # MIOpen: exhaustive find instead of the default heuristic pick
    with torch.device("cuda"):
        unet = UNet2DConditionModel(UNetConfig.sd15()).half()
    pipe = GuidedAttention(unet, None, None, SyntheticTextEncoder(768), WordTokenizer()).to("cuda", torch.float16)
    if "nhwc-off" in sys.argv:
        pipe.unet.set_norm_impl(None)
        pipe.unet.to(memory_format=torch.contiguous_format)
    pipe.guidance_forward = "truncated" if "truncated" in sys.argv else "full"
    rc = RunConfig(meta_prompt="a [robot:.6,.3,.4,.55] and a [blue vase:.2,.3,.4,.55]", output_path="/tmp/ga_ub")
    rc.stable = pipe
    state.curHyperParams = state.get_hyperparam_states()[0]
    run.overrideConfig(rc)
    run.parseMetaPrompt(rc)
    store = ptp_utils.AttentionStore()
    ptp_utils.register_attention_control(pipe, store)
    pipe._attention_store = store
    pipe._truncate_at = pipe._truncation_point(16, 512, 512)
    emb = torch.randn(2, 77, 768, device="cuda", dtype=torch.half)
    lat = torch.randn(1, 4, 64, 64, device="cuda", dtype=torch.half)
    r = GraphRunner(pipe, store, emb, lat, 16, True, 0.5, 3, False)
    only = next((a.split("=", 1)[1] for a in sys.argv if a.startswith("only=")), None)
    if only:  # for rocprofv3: one pass replayed 60 times dominates the kernel statistics
        g = {"eval": r.g_eval, "grad": r.g_grad, "cfg": r.g_cfg, "joint": getattr(r, "g_joint", None)}[only]
        replay_ms(g, 5)
        from guided_attention_amd import ops
        ops.latent_axpby(lat, lat, 1.0, 0.0)   # marker launch for tools/trace_window_stats.py (no pass uses this kernel)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(60):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        print(f"{only}: {e0.elapsed_time(e1) / 60:7.3f} ms")
        return
    print(f"guidance forward + loss (B=1): {replay_ms(r.g_eval):7.3f} ms")
    print(f"backward to the latents      : {replay_ms(r.g_grad):7.3f} ms")
    print(f"CFG forward (B=2)            : {replay_ms(r.g_cfg):7.3f} ms")
    if getattr(r, "g_joint", None) is not None:
        print(f"joint pass (B=3) + loss      : {replay_ms(r.g_joint):7.3f} ms")


if __name__ == "__main__":
    main()
