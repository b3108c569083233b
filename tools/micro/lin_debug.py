#!/usr/bin/env python3
"""Which tiles of a ga_linear_fused result are wrong, per plan (debugging aid)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from guided_attention_amd import ops  # noqa: E402

dev = torch.device("cuda")
ops.prepare_device(dev)
torch.manual_seed(0)
for M, K, N in ((4096, 320, 320), (12288, 320, 320), (4096, 640, 640)):
    x = torch.randn(M, K, device=dev, dtype=torch.half)
    w = torch.randn(N, K, device=dev, dtype=torch.half) * K ** -0.5
    ref = (x.float() @ w.float().T)
    for plan in ((128, 64, 1, 3), (128, 64, 2, 3), (128, 64, 5, 3), (64, 128, 2, 3), (128, 64, 2, 4), (64, 64, 1, 5), (64, 64, 2, 5),
                 (64, 64, 5, 5), (128, 128, 2, 3), (128, 128, 2, 2)):
        for rep in range(20):
            y = ops.linear_fused(x, w, None, plan=plan)["y"].float()
            err = (y - ref).abs()
            bm, bn = plan[0], plan[1]
            if M % bm or N % bn:
                continue
            te = err.reshape(M // bm, bm, N // bn, bn).amax((1, 3))
            bad = (te > 0.05).nonzero()
            print(M, K, N, plan, rep, "max err %.3f" % float(err.max()), "bad tiles", len(bad), "of", te.numel(),
                  bad[:6].tolist(), flush=True)
            if len(bad):
                mt, nt = bad[0].tolist()
                e = err[mt * bm:(mt + 1) * bm, nt * bn:(nt + 1) * bn]
                rows = (e.amax(1) > 0.05).nonzero().flatten().tolist()
                cols = (e.amax(0) > 0.05).nonzero().flatten().tolist()
                print("   rows", rows[:40], "cols", cols[:40])
