#!/usr/bin/env python3
"""GPU idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV (the timed region only if --after is the
name of a marker kernel): how much of the wall time the device spends waiting for the host or for launch boundaries.
usage: gap_report.py <kernel_trace.csv> [skip_first_seconds | between=<kernel substring>,<first occurrence>,<last occurrence>]
  between=cfg_ddim,51,150 restricts the report to the span from the 51st to the 150th launch of the DDIM step kernel: the two
  timed images of `bench.py --steps 2 --warmup 1` (50 steps each, after the 50 of the warm-up image)."""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
t0 = rows[0][0]
if len(sys.argv) > 2 and sys.argv[2].startswith("between="):
    name, a, b = sys.argv[2][8:].split(",")
    marks = [i for i, r in enumerate(rows) if name in r[2]]
    rows = rows[marks[int(a) - 1]:marks[int(b) - 1] + 1]
else:
    skip = float(sys.argv[2]) * 1e9 if len(sys.argv) > 2 else 0.0
    rows = [r for r in rows if r[0] - t0 >= skip]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = []
end = rows[0][1]
for i in range(1, len(rows)):
    s, e, n = rows[i]
    if s > end:
        gaps.append((s - end, rows[i - 1][2][:60], n[:60]))
    end = max(end, e)
tot = sum(g[0] for g in gaps)
print(f"kernels {len(rows)}  span {span / 1e6:.1f} ms  busy {busy / 1e6:.1f} ms ({100 * busy / span:.1f} %)  idle {tot / 1e6:.1f} ms")
for lo, hi in ((0, 2e3), (2e3, 5e3), (5e3, 20e3), (20e3, 100e3), (100e3, 1e6), (1e6, 1e12)):
    sel = [g[0] for g in gaps if lo <= g[0] < hi]
    print(f"  gaps {lo / 1e3:7.0f} .. {hi / 1e3:9.0f} us: {len(sel):7d}  total {sum(sel) / 1e6:8.2f} ms")
print("largest gaps (us, after kernel -> before kernel):")
for g in sorted(gaps, reverse=True)[:12]:
    print(f"  {g[0] / 1e3:9.1f}  {g[1]}  ->  {g[2]}")

print("gaps >= 100 us by (kernel before -> kernel after):")
pairs = {}
for g in gaps:
    if g[0] >= 100e3:
        k = (g[1][:48], g[2][:48])
        c = pairs.setdefault(k, [0, 0.0])
        c[0] += 1
        c[1] += g[0]
for k, (n, t) in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"  {n:5d} x  avg {t / n / 1e3:7.1f} us  total {t / 1e6:7.2f} ms   {k[0]}  ->  {k[1]}")
