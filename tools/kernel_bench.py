#!/usr/bin/env python3
"""Per-shape launch time of the capture kernels on the SD-1.x layer shapes (GPU box only):
back-to-back replay between two HIP events, algorithmic bytes / time vs the 8 TB/s HBM peak."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from guided_attention_amd import ops  # noqa: E402


def capture_bytes(key):
    kind, B, H, N, Kt, D, flag, dt = key
    esz, C = 2, H * D
    kv = 2 * B * Kt * C
    if kind == "attn_capture_fwd":
        return esz * (2 * B * N * C + kv + (B * H * N * Kt if flag else 0))
    return esz * (3 * B * N * C + kv + (N * Kt if flag else 0))


def main():
    ops.load()
    print(f"{'kernel':18s} {'B':>2s} {'N':>5s} {'D':>4s} {'flag':>5s} {'us':>8s} {'GB/s':>8s} {'frac':>6s}")
    for B in (1, 2):
        for N, D in ((4096, 40), (1024, 80), (256, 160), (64, 160)):
            for kind, flag in (("attn_capture_fwd", False), ("attn_capture_fwd", True), ("attn_capture_bwd", False),
                               ("attn_capture_bwd", True)):
                if B == 2 and kind == "attn_capture_bwd":
                    continue
                key = (kind, B, 8, N, 77, D, flag, "torch.float16")
                us = ops.replay_launch_us(key, 300)
                gbs = capture_bytes(key) / us / 1e3
                print(f"{kind:18s} {B:2d} {N:5d} {D:4d} {str(flag):>5s} {us:8.2f} {gbs:8.1f} {gbs / 8000:6.3f}", flush=True)


if __name__ == "__main__":
    main()
