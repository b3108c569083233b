#!/usr/bin/env python3
"""Launch ONE hand-written kernel a few times on a given shape (for `rocprofv3 --pmc ... -- python3 tools/kernel_once.py`).
  kernel_once.py sa_fwd|sa_bwd B H N D [dtype]        self-attention forward / backward (needs a forward first)
  kernel_once.py cap_fwd|cap_bwd B H N D [dtype]      cross-attention capture (Kt = 77; fwd stores P)
  kernel_once.py gn_fwd|gn_bwd B C HW [dtype]         GroupNorm(+SiLU), 32 groups, channels-last
  kernel_once.py conv B Cin HW stride Cout [dtype]    3x3 implicit-GEMM convolution with bias + residual, the planner's tile
  kernel_once.py lin M K N flags [dtype]              ga_linear_fused: flags = geglu | 2 LayerNorm fold | 4 residual
Inputs are resident in HBM before the launches; 5 launches each."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from guided_attention_amd import ops  # noqa: E402

kind = sys.argv[1]
nums = [int(x) for x in sys.argv[2:] if x.lstrip("-").isdigit()]
dt = {"f16": torch.half, "bf16": torch.bfloat16, "f32": torch.float32}[next((x for x in sys.argv[2:] if x in ("f16", "bf16", "f32")), "f16")]
dev = "cuda"
reps = 5
if kind.startswith("sa_"):
    B, H, N, D = nums
    q, k, v, do = (torch.randn(B, N, H * D, device=dev, dtype=dt) for _ in range(4))
    o, lse = ops.self_attn_fwd(q, k, v, H, D ** -0.5)
    torch.cuda.synchronize()
    for _ in range(reps):
        if kind == "sa_fwd":
            ops.self_attn_fwd(q, k, v, H, D ** -0.5)
        else:
            ops.self_attn_bwd(q, k, v, o, do, lse, H, D ** -0.5)
elif kind.startswith("cap_"):
    B, H, N, D = nums
    q = torch.randn(B, N, H * D, device=dev, dtype=dt)
    k, v = (torch.randn(B, 77, H * D, device=dev, dtype=dt) for _ in range(2))
    do = torch.randn_like(q)
    dp = (torch.randn(N, 77, device=dev, dtype=dt) * 1e-3).unsqueeze(0).expand(B * H, N, 77)
    for _ in range(reps):
        if kind == "cap_fwd":
            ops.attn_capture_fwd(q, k, v, H, D ** -0.5, True)
        else:
            ops.attn_capture_bwd(q, k, v, do, dp, H, D ** -0.5)
elif kind.startswith("gn_"):
    B, C, HW = nums
    side = int(round(HW ** 0.5))
    x = torch.randn(B, C, side, HW // side, device=dev, dtype=dt).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w, b_ = torch.ones(C, device=dev, dtype=dt), torch.zeros(C, device=dev, dtype=dt)
    g = torch.randn_like(x)
    for _ in range(reps):
        y = ops.group_norm_act(x, w, b_, 32, 1e-5, True)
        if kind == "gn_bwd":
            y.backward(g)
elif kind == "conv":
    B, cin, HW, stride, cout = nums
    side = int(round(HW ** 0.5))
    x = torch.randn(B, cin, side, HW // side, device=dev, dtype=dt).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 3, 3, device=dev, dtype=dt) * (9 * cin) ** -0.5).contiguous(memory_format=torch.channels_last)
    bias = torch.randn(cout, device=dev, dtype=dt)
    ho = (side - 1) // stride + 1
    res = torch.randn(B, cout, ho, (HW // side - 1) // stride + 1, device=dev, dtype=dt).contiguous(memory_format=torch.channels_last)
    wp = ops.conv3x3_packed_weights(w, False)
    torch.cuda.synchronize()
    for _ in range(reps):
        ops.conv3x3_nhwc(x, wp, cout, stride, bias, res)
elif kind == "lin":
    M, K, N, flags = nums
    x = torch.randn(M, K, device=dev, dtype=dt)
    w = torch.randn(N, K, device=dev, dtype=dt) * K ** -0.5
    bias = torch.randn(N, device=dev, dtype=dt)
    n_out = N // 2 if flags & 1 else N
    res = torch.randn(M, n_out, device=dev, dtype=dt) if flags & 4 else None
    ln = (torch.rand(M, 5, 2, device=dev) * K, torch.randn(N, device=dev), torch.randn(N, device=dev), 1e-5) if flags & 2 else None
    ops.prepare_device(torch.device(dev, torch.cuda.current_device()))
    torch.cuda.synchronize()
    for _ in range(reps):
        ops.linear_fused(x, w, None if ln is not None else bias, residual=res, geglu=bool(flags & 1), ln=ln)   # (the fold's shift carries the bias)
else:
    raise SystemExit(f"unknown kernel kind {kind}")
torch.cuda.synchronize()
