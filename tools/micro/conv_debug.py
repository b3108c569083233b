#!/usr/bin/env python3
"""conv3x3_nhwc against fp64 for one shape over plans: where are the wrong elements?  usage: conv_debug.py B Cin Cout H W"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from guided_attention_amd import ops  # noqa: E402

B, Cin, Cout, H, W = [int(a) for a in sys.argv[1:6]]
torch.manual_seed(0)
x = torch.randn(B, Cin, H, W, device="cuda", dtype=torch.half).contiguous(memory_format=torch.channels_last)
w = (torch.randn(Cout, Cin, 3, 3, device="cuda") * (9 * Cin) ** -0.5).half()
ref = torch.nn.functional.conv2d(x.double().cpu(), w.double().cpu(), padding=1)
wp = ops.conv3x3_packed_weights(w, False)
for bm, bn in ((128, 64), (64, 64)):
    for splits in (1, 2, 3, 4, 5, 8, 16):
        if splits > 9 * Cin // 64:
            continue
        ws = splits * B * H * W * Cout if splits > 1 else 0
        errs = []
        for rep in range(4):
            y = ops.conv3x3_nhwc(x, wp, Cout, 1, None, None, plan=(bm, bn, splits, ws))
            d = (y.double().cpu() - ref).abs()
            errs.append(float(d.max()))
        bad = (d > 0.05).nonzero()
        where = ""
        if len(bad):
            ch = sorted(set(int(v) for v in bad[:, 1]))
            rows = sorted(set(int(v) for v in bad[:, 2]))
            cols = sorted(set(int(v) for v in bad[:, 3]))
            where = f"  {len(bad)} bad: channels {ch[:6]}..{ch[-1]} ({len(ch)}), rows {rows[:8]}..{rows[-1]} ({len(rows)}), cols {cols[:8]}..{cols[-1]} ({len(cols)})"
        print(f"tile {bm}x{bn} splits {splits:2d}: max err per launch {['%.3g' % e for e in errs]}{where}")
