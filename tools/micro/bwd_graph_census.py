#!/usr/bin/env python3
"""Census of the autograd graph of one guidance evaluation (SD-1.x UNet, fp16, eager): node types, and the places where a
gradient is ACCUMULATED (a node whose output feeds more than one consumer costs fan-in - 1 element-wise add launches in the
backward) or copied between layouts — what tools/unet_bench.py only=grad shows as torch add / copy kernels."""
import sys
from collections import Counter
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from guided_attention_amd import run  # noqa: E402
from guided_attention_amd.config import RunConfig  # noqa: E402
from guided_attention_amd.graphs import GraphRunner  # noqa: E402
from guided_attention_amd.pipeline_guided_attention import GuidedAttention  # noqa: E402
from guided_attention_amd.text import SyntheticTextEncoder, WordTokenizer  # noqa: E402
from guided_attention_amd.unet import UNet2DConditionModel, UNetConfig  # noqa: E402
from guided_attention_amd.utils import ptp_utils, shared_state as state  # noqa: E402

with torch.device("cuda"):
    unet = UNet2DConditionModel(UNetConfig.sd15()).half()
pipe = GuidedAttention(unet, None, None, SyntheticTextEncoder(768), WordTokenizer()).to("cuda", torch.float16)
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
parts, _ = r._eval_body(store)
loss = parts[1] if isinstance(parts, (tuple, list)) else parts
loss = r.loss if hasattr(r, "loss") and r.loss.grad_fn is not None else loss
root = loss.grad_fn
seen, fan_in, order = {}, Counter(), []
stack = [root]
while stack:
    fn = stack.pop()
    if id(fn) in seen:
        continue
    seen[id(fn)] = fn
    order.append(fn)
    for nxt, _ in fn.next_functions:
        if nxt is not None:
            fan_in[id(nxt)] += 1
            stack.append(nxt)
names = Counter(type(f).__name__ for f in order)
print(len(order), "nodes")
for n, c in names.most_common(40):
    print(f"  {c:5d} {n}")
multi = Counter()
for i, c in fan_in.items():
    if c > 1:
        f = seen[i]
        consumers = sorted(type(p).__name__ for p in order if any(nx is f for nx, _ in p.next_functions))
        multi[(type(f).__name__, tuple(consumers))] += c - 1
print("accumulations (node, consumers) -> adds:")
for (n, cons), c in multi.most_common(40):
    print(f"  {c:4d} {n} <- {', '.join(cons)}")

# run the backward eagerly under the profiler: which aten ops launch copies / adds
from torch.profiler import profile, ProfilerActivity  # noqa: E402
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=False) as prof:
    torch.autograd.grad(loss, [r.lat_g], retain_graph=True)
ops_ = Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::add", "aten::add_", "aten::contiguous", "aten::clone", "aten::cat", "aten::zeros",
                  "aten::fill_", "aten::zero_", "aten::sum", "aten::mul", "aten::slice_backward", "aten::narrow"):
        ops_[(e.name, str(e.input_shapes)[:120])] += 1
print("aten ops of interest during the backward:")
for (n, shp), c in sorted(ops_.items(), key=lambda kv: -kv[1])[:60]:
    print(f"  {c:4d} {n} {shp}")

INTEREST = ("aten::copy_", "aten::add", "aten::add_", "aten::contiguous", "aten::clone", "aten::cat", "aten::zeros", "aten::mul",
            "aten::fill_", "aten::zero_", "aten::sum", "aten::to", "aten::_to_copy", "aten::silu", "aten::empty_like",
            "aten::upsample_nearest2d", "aten::linear", "aten::addmm", "aten::convolution", "aten::mm", "aten::bmm", "aten::div",
            "aten::sub", "aten::exp", "aten::cos", "aten::sin", "aten::index", "aten::stack")
for name, body in (("guidance forward (B=1, autograd)", lambda: r._eval_body(store)), ("CFG forward (B=2)", lambda: r._cfg_body(store))):
    with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
        body()
    c = Counter()
    for e in prof.events():
        if e.name in INTEREST:
            c[(e.name, str(e.input_shapes)[:110])] += 1
    print(f"aten ops of interest during the {name}:")
    for (n, shp), k in sorted(c.items(), key=lambda kv: -kv[1])[:45]:
        print(f"  {k:4d} {n} {shp}")
