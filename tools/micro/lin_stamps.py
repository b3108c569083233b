#!/usr/bin/env python3
"""Where a launch of linear_kernel spends its time, from in-kernel clock stamps (diagnostic library: `make stamps` ->
tools/micro/libga_stamps.so; wave 0 of every workgroup stamps the phase boundaries, cdna_hip_programming.md section 7).

  python3 tools/micro/lin_stamps.py M K N flags [bm bn splits stages] [reps]        flags = geglu | 2 LayerNorm fold | 4 residual

Prints, over the workgroups of the LAST of `reps` launches (cold weights: the launches rotate over weight copies), the median /
10th / 90th percentile of each phase in microseconds, the spread of workgroup start times and the span first start -> last end.
The diagnostic build's fences forbid overlaps the real kernel has: read the SHARES, not the total (compare `span` with the
graph-replay time of the product kernel printed last)."""
import ctypes
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
os.environ["GA_HIP_LIB"] = str(ROOT / "tools" / "micro" / "libga_stamps.so")
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from guided_attention_amd import ops  # noqa: E402
from guided_attention_amd._lib import LIB_PATH  # noqa: E402

PHASES = ["prologue: tile map, addresses, first ring issues, epilogue prefetch, LN statistics",
          "wait for the first k-step (first operand round trip)",
          "main loop (remaining k-steps)",
          "split-K hand-off (store slabs, ticket, re-read)",
          "epilogue 1: bias / LN algebra, accumulators -> LDS, barrier",
          "epilogue 2: LDS -> GEGLU / residual -> global stores issued",
          "drain: stores retired"]


def report_stream(t, M, K, N, flags, plan, wgs):
    """linear_stream_kernel: [0] start [1] prologue done [2..5] sums over the workgroup's stream (waits + barriers, k-step bodies,
    row statistics, epilogues) [6] tiles of the workgroup [7] end."""
    clk_mhz = np.median((t[:, 7] - t[:, 0]) / np.maximum(t[:, 9] - t[:, 8], 1)) * 100.0
    us = lambda cyc: cyc / clk_mhz                                                         # noqa: E731
    life = us(t[:, 7] - t[:, 0])
    tiles = np.maximum(t[:, 6], 1)
    print(f"shape M={M} K={K} N={N} flags={flags} stream form: {wgs} workgroups, {np.median(tiles):.0f} tiles each (median), stamp clock {clk_mhz:.0f} MHz")
    rows = [("prologue (tile map, first three k-steps issued)", us(t[:, 1] - t[:, 0])),
            ("waits for the ring + k-step barriers, per tile", us(t[:, 2]) / tiles),
            ("k-step bodies (fragment reads, MFMAs, refill pieces), per tile", us(t[:, 3]) / tiles),
            ("constants issue + row statistics, per tile", us(t[:, 4]) / tiles),
            ("epilogues, per tile", us(t[:, 5]) / tiles)]
    for name, d in rows:
        print(f"  {np.median(d):7.2f} us  (p10 {np.percentile(d, 10):6.2f}, p90 {np.percentile(d, 90):6.2f})  {name}")
    print(f"  {np.median(life):7.2f} us  workgroup lifetime (median; p90 {np.percentile(life, 90):.2f}) = {np.median(life / tiles):.2f} us per tile")
    start = (t[:, 8] - t[:, 8].min()) / 100.0
    print(f"  workgroup starts: last {start.max():.2f} us after the first; span first start -> last end {((t[:, 9] - t[:, 8].min()) / 100.0).max():.2f} us")


def main():
    nums = [int(a) for a in sys.argv[1:]]
    M, K, N, flags = nums[:4]
    plan = tuple(nums[4:8]) if len(nums) >= 8 else None
    reps = nums[8] if len(nums) >= 9 else (nums[4] if len(nums) == 5 else 12)
    dev = torch.device("cuda")
    lib = ops.load()
    assert "stamps" in str(LIB_PATH)
    lib.ga_lin_set_stamps.argtypes, lib.ga_lin_set_stamps.restype = [ctypes.c_void_p], ctypes.c_int
    geglu, ln_, res_ = bool(flags & 1), bool(flags & 2), bool(flags & 4)
    plan = plan or ops.linear_plan(M, K, N, geglu)
    bm, bn, splits = plan[:3]
    outc = bn // 2 if geglu else bn
    n_out = N // 2 if geglu else N
    wgs = -(-M // bm) * -(-n_out // outc) * splits
    stream = len(plan) > 3 and plan[3] == ops.LINEAR_STREAM
    if stream:
        wgs = min(-(-M // 128) * -(-n_out // (64 if geglu else 128)), torch.cuda.get_device_properties(0).multi_processor_count)
    stamps = torch.zeros(wgs, 10, dtype=torch.int64, device=dev)
    assert lib.ga_lin_set_stamps(ctypes.c_void_p(stamps.data_ptr())) == 0
    x = torch.randn(M, K, device=dev, dtype=torch.half)
    n_copies = max(2, min(64, -(-320 * 2 ** 20 // (N * K * 2))))
    ws = [torch.randn(N, K, device=dev, dtype=torch.half) * K ** -0.5 for _ in range(n_copies)]
    bias = torch.randn(N, device=dev, dtype=torch.half)
    res = torch.randn(M, n_out, device=dev, dtype=torch.half) if res_ else None
    ln = (torch.rand(M, 5, 2, device=dev) * K, torch.randn(N, device=dev), torch.randn(N, device=dev), 1e-5) if ln_ else None
    want_partials = res_ and not stream
    ops.prepare_device(dev)
    for i in range(reps):
        ops.linear_fused(x, ws[i % n_copies], None if ln_ else bias, residual=res, geglu=geglu, ln=ln, want_row_partials=want_partials,
                         plan=plan)
    torch.cuda.synchronize()
    t = stamps.cpu().numpy().astype(np.float64)
    if stream:
        return report_stream(t, M, K, N, flags, plan, wgs)
    real0, real1 = t[:, 8], t[:, 9]
    clk_mhz = np.median((t[:, 7] - t[:, 0]) / np.maximum(real1 - real0, 1)) * 100.0      # s_memrealtime ticks at 100 MHz
    us = lambda cyc: cyc / clk_mhz                                                         # noqa: E731
    print(f"shape M={M} K={K} N={N} flags={flags} plan={plan}: {wgs} workgroups, stamp clock {clk_mhz:.0f} MHz")
    total = t[:, 7] - t[:, 0]
    for i, name in enumerate(PHASES):
        d = us(t[:, i + 1] - t[:, i])
        print(f"  {np.median(d):7.2f} us  (p10 {np.percentile(d, 10):6.2f}, p90 {np.percentile(d, 90):6.2f})  {100 * np.median(d) / np.median(us(total)):5.1f} %  {name}")
    print(f"  {np.median(us(total)):7.2f} us  workgroup lifetime (median; p10 {np.percentile(us(total), 10):.2f}, p90 {np.percentile(us(total), 90):.2f})")
    start = (real0 - real0.min()) / 100.0
    end = (real1 - real0.min()) / 100.0
    print(f"  workgroup starts: median {np.median(start):.2f} us after the first, last {start.max():.2f} us; span first start -> last end {end.max():.2f} us")
    # the product kernel of the same plan, graph replay at cold weights, for scale
    turn = [0]

    def fn():
        turn[0] += 1
        ops.linear_fused(x, ws[turn[0] % n_copies], None if ln_ else bias, residual=res, geglu=geglu, ln=ln,
                         want_row_partials=want_partials, plan=plan)
    side = ops.side_stream(dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(40):
                fn()
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        side.synchronize()
        e0.record(side)
        g.replay()
        e1.record(side)
        side.synchronize()
    print(f"  this (stamped) build, graph replay: {e0.elapsed_time(e1) * 1e3 / 40:.2f} us per launch")


if __name__ == "__main__":
    main()
