#!/usr/bin/env python3
"""Where a launch of conv3x3_patch_dma_kernel spends its time, from in-kernel clock stamps (diagnostic library: `make stamps` ->
tools/micro/libga_conv_stamps.so; wave 0 of every workgroup stamps its phase boundaries and adds up, over its k-steps, the time
in front of the step barrier, in the refill issue, in the MFMA body and in the patch hand-over).

  python3 tools/micro/conv_stamps.py B Cin H W Cout [bm bn splits] [reps]

Median / 10th / 90th percentile over the workgroups of the LAST of `reps` launches (cold weights: the launches rotate over weight
copies).  The diagnostic build's fences forbid overlaps the real kernel has: read the SHARES (the product kernel's graph-replay
time is printed last)."""
import ctypes
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
os.environ["GA_HIP_LIB"] = os.environ.get("GA_STAMPS_LIB", str(ROOT / "tools" / "micro" / "libga_conv_stamps.so"))   # GA_STAMPS_LIB: an ablation build
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from guided_attention_amd import ops  # noqa: E402
from guided_attention_amd._lib import LIB_PATH  # noqa: E402


def main():
    nums = [int(a) for a in sys.argv[1:]]
    B, Cin, H, W, Cout = nums[:5]
    plan = tuple(nums[5:8]) if len(nums) >= 8 else None
    reps = nums[8] if len(nums) >= 9 else (nums[5] if len(nums) == 6 else 12)
    dev = torch.device("cuda")
    lib = ops.load()
    assert "conv_stamps" in str(LIB_PATH)
    lib.ga_conv_set_stamps.argtypes, lib.ga_conv_set_stamps.restype = [ctypes.c_void_p], ctypes.c_int
    bm, bn, splits = (plan or ops.conv3x3_plan(B, H, W, Cin, Cout, 1))[:3]
    M = B * H * W
    wgs = -(-M // bm) * -(-Cout // bn) * splits
    stamps = torch.zeros(wgs, 12, dtype=torch.int64, device=dev)
    assert lib.ga_conv_set_stamps(ctypes.c_void_p(stamps.data_ptr())) == 0
    x = torch.randn(B, Cin, H, W, device=dev, dtype=torch.half).contiguous(memory_format=torch.channels_last)
    n_copies = max(2, min(24, -(-320 * 2 ** 20 // (9 * Cout * Cin * 2))))
    wps = [ops.conv3x3_packed_weights(torch.randn(Cout, Cin, 3, 3, device=dev, dtype=torch.half) * (9 * Cin) ** -0.5, False)
           for _ in range(n_copies)]
    bias = torch.randn(Cout, device=dev, dtype=torch.half)
    ws = splits * M * Cout if splits > 1 else 0
    for r in range(reps):
        ops.conv3x3_nhwc(x, wps[r % n_copies], Cout, 1, bias, None, plan=(bm, bn, splits, ws))
    torch.cuda.synchronize()
    t = stamps.cpu().numpy().astype(np.float64)
    if not t[:, 0].any():
        print(f"shape B={B} Cin={Cin} {H}x{W} Cout={Cout} plan {bm}x{bn} x{splits}: not served by conv3x3_patch_dma_kernel (no stamps)")
        return
    clk_mhz = np.median((t[:, 10] - t[:, 0]) / np.maximum(t[:, 9] - t[:, 8], 1)) * 100.0
    us = lambda cyc: cyc / clk_mhz                                                         # noqa: E731
    steps = np.maximum(t[:, 6], 1)
    print(f"shape B={B} Cin={Cin} {H}x{W} Cout={Cout}  tile {bm}x{bn}, {splits} k-slices: {wgs} workgroups, {np.median(steps):.0f} k-steps each, "
          f"stamp clock {clk_mhz:.0f} MHz")
    rows = [("prologue (arguments, first weight DMAs, patch offsets, first patch -> LDS)", us(t[:, 1] - t[:, 0])),
            ("k-steps: wait for the step's weights + step barrier, sum", us(t[:, 2])),
            ("k-steps: refill issue (weight DMA pieces, next chunk's patch loads), sum", us(t[:, 3])),
            ("k-steps: fragment reads + MFMAs, sum", us(t[:, 4])),
            ("patch hand-over per chunk (barrier, registers -> LDS), sum", us(t[:, 5])),
            ("drain + epilogue (split-K hand-off, bias, stores retired)", us(t[:, 10] - t[:, 7]))]
    for name, d in rows:
        print(f"  {np.median(d):7.2f} us  (p10 {np.percentile(d, 10):6.2f}, p90 {np.percentile(d, 90):6.2f})  {name}")
    life = us(t[:, 10] - t[:, 0])
    per = us(t[:, 2] + t[:, 3] + t[:, 4] + t[:, 5]) / steps
    print(f"  {np.median(life):7.2f} us  workgroup lifetime (p90 {np.percentile(life, 90):.2f}); {np.median(per):.3f} us per k-step "
          f"(wait {np.median(us(t[:, 2]) / steps):.3f}, issue {np.median(us(t[:, 3]) / steps):.3f}, MFMA body {np.median(us(t[:, 4]) / steps):.3f})")
    start = (t[:, 8] - t[:, 8].min()) / 100.0
    print(f"  workgroup starts: last {start.max():.2f} us after the first; span first start -> last end {((t[:, 9] - t[:, 8].min()) / 100.0).max():.2f} us")
    assert lib.ga_conv_set_stamps(None) == 0
    key = ("conv3x3", B, Cin, H * W, 1, Cout, True, "torch.float16")
    print(f"  graph replay of the same call (stamps build, cold weights): {ops.replay_launch_us(key):.2f} us")


if __name__ == "__main__":
    main()
