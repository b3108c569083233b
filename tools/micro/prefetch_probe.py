#!/usr/bin/env python3
"""Does touching a layer's weights one kernel EARLIER make the consuming convolution faster (Infinity Cache hit instead of HBM)?
Per shape, hipGraph-replayed chains over weight copies that exceed the 256 MB Infinity Cache:
   cold       conv(w[i])                                   every launch streams its weights from HBM
   warm       conv(w[0])                                   the same 15 - 59 MB every launch (L2 / Infinity Cache resident)
   touch      touch(w[i])                                  the touch kernel alone (a full read: torch.max over the int16 view)
   prefetched touch(w[i+1]); conv(w[i])                    the convolution finds its weights touched one launch earlier
What the in-kernel prefetch could gain at best per launch = cold - (prefetched - touch)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from guided_attention_amd import ops  # noqa: E402
from conv_tune import replay_us  # noqa: E402

ops.prepare_device(torch.device("cuda"))
print(f"{'B':>2} {'Cin':>5} {'Cout':>5} {'HW':>3} | {'cold':>7} {'warm':>7} {'touch':>7} {'pref':>7} | conv after touch, gain vs cold")
for B, ci, co, h in ((1, 1280, 1280, 16), (3, 1280, 1280, 16), (1, 1280, 1280, 8), (1, 2560, 1280, 16), (1, 640, 640, 32), (3, 640, 640, 32),
                     (1, 320, 320, 64)):
    x = torch.randn(B, ci, h, h, device="cuda", dtype=torch.half).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(co, ci, 3, 3, device="cuda", dtype=torch.half) * 0.02).contiguous(memory_format=torch.channels_last)
    wp = ops.conv3x3_packed_weights(w, False)
    n = max(2, min(48, -(-400 * 2 ** 20 // (wp.numel() * 2))))
    wps = [wp] + [wp.clone() for _ in range(n - 1)]
    views = [p.view(torch.int16) for p in wps]
    turn = [0]

    def cold():
        turn[0] += 1
        ops.conv3x3_nhwc(x, wps[turn[0] % n], co, 1)

    def warm():
        ops.conv3x3_nhwc(x, wps[0], co, 1)

    def touch():
        turn[0] += 1
        torch.max(views[turn[0] % n])

    def pref():
        turn[0] += 1
        torch.max(views[(turn[0] + 1) % n])
        ops.conv3x3_nhwc(x, wps[turn[0] % n], co, 1)

    it = max(10, min(n, 40))
    tc, tw, tt, tp = (replay_us(f, iters=it) for f in (cold, warm, touch, pref))
    print(f"{B:>2} {ci:>5} {co:>5} {h:>3} | {tc:7.1f} {tw:7.1f} {tt:7.1f} {tp:7.1f} | {tp - tt:7.1f}  {100 * (1 - (tp - tt) / tc):5.1f} %", flush=True)
