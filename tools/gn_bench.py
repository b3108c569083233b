#!/usr/bin/env python3
"""GroupNorm(+SiLU) forward / backward launch times for the UNet's norm shapes (hipGraph replay, HIP events)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from guided_attention_amd import ops  # noqa: E402

ops.load()
shapes = [(1, 320, 4096), (3, 320, 4096), (1, 640, 4096), (1, 960, 4096), (1, 640, 1024), (3, 640, 1024), (1, 1280, 1024), (1, 1920, 1024),
          (1, 1280, 256), (1, 2560, 256), (1, 1280, 64)]
for B, C, HW in shapes:
    row = []
    for kind in ("group_norm_fwd", "group_norm_bwd"):
        us = ops.replay_launch_us((kind, B, 32, HW, 0, C, True, "torch.float16"))
        nbytes = 2 * B * C * HW * (2 if kind.endswith("fwd") else 3)
        row.append(f"{kind[-3:]} {us:6.2f} us {nbytes / us / 1e3:7.1f} GB/s")
    print(f"B={B} C={C:5d} HW={HW:5d}  " + "   ".join(row))
