#!/usr/bin/env python3
"""Per-shape time of the SD-1.x 3x3 convolutions through the library (MIOpen, benchmark mode, channels-last fp16),
forward and backward-data, plus a plain GEMM of the same M x N x K for scale.  hipGraph replay timing."""
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
torch.backends.cudnn.benchmark = True
SHAPES = [  # (B, Cin, Cout, H)
    (1, 320, 320, 64), (2, 320, 320, 64), (1, 640, 320, 64), (1, 960, 320, 64),
    (1, 320, 640, 32), (1, 640, 640, 32), (2, 640, 640, 32), (1, 1280, 640, 32), (1, 1920, 640, 32),
    (1, 640, 1280, 16), (1, 1280, 1280, 16), (2, 1280, 1280, 16), (1, 2560, 1280, 16), (1, 1920, 1280, 16),
    (1, 1280, 1280, 8), (1, 2560, 1280, 8),
]


def replay_us(fn, iters=20):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(iters):
                fn()
        g.replay()
        s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(5):
            g.replay()
        e1.record(s)
        e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * iters)


print(f"{'B':>2} {'Cin':>5} {'Cout':>5} {'HW':>4} {'fwd us':>8} {'TF/s':>7} {'dgrad us':>9} {'TF/s':>7} {'gemm us':>8} {'TF/s':>7}")
for B, ci, co, h in SHAPES:
    x = torch.randn(B, ci, h, h, device="cuda", dtype=torch.half).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(co, ci, 3, 3, device="cuda", dtype=torch.half) * 0.02).contiguous(memory_format=torch.channels_last)
    b = torch.zeros(co, device="cuda", dtype=torch.half)
    gy = torch.randn(B, co, h, h, device="cuda", dtype=torch.half).contiguous(memory_format=torch.channels_last)
    flop = 2.0 * B * h * h * co * ci * 9
    t_f = replay_us(lambda: F.conv2d(x, w, None, padding=1))
    t_b = replay_us(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1,
                                                                (True, False, False)))
    a2 = torch.randn(B * h * h, ci * 9, device="cuda", dtype=torch.half)
    w2 = torch.randn(co, ci * 9, device="cuda", dtype=torch.half)
    t_g = replay_us(lambda: F.linear(a2, w2))
    print(f"{B:>2} {ci:>5} {co:>5} {h:>4} {t_f:8.1f} {flop / t_f / 1e6:7.1f} {t_b:9.1f} {flop / t_b / 1e6:7.1f} {t_g:8.1f} {flop / t_g / 1e6:7.1f}")
