#!/usr/bin/env python3
"""Launch time of the loss kernels by token count (slope = per-token cost, intercept = statistics + hand-off):
aggregate_maps, smooth_loss_fwd, the fused aggregate + loss launch, smooth_loss_bwd.  hipGraph replay, 100 launches."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from guided_attention_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda")
    ops.prepare_device(dev)
    for res in (16, 32):
        npix = res * res
        for T in (1, 3, 8):
            key = lambda kind, flag=False, dt="torch.float16": (kind, T, 40, npix, 77, 0, flag, dt)  # noqa: E731
            row = {k: ops.replay_launch_us(key(k, f, d)) for k, f, d in (
                ("aggregate_maps", False, "torch.float16"), ("smooth_loss_fwd", False, "torch.float32"),
                ("aggregate_loss_fwd", False, "torch.float16"), ("smooth_loss_bwd", True, "torch.float16"))}
            print(f"res {res:2d} T {T}: " + "  ".join(f"{k} {v:6.1f} us" for k, v in row.items()), flush=True)


if __name__ == "__main__":
    main()
