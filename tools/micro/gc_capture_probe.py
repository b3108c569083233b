#!/usr/bin/env python3
"""Settles ONE question (round-3 review, ADVICE): does a cycle collection that runs INSIDE an open hipGraph capture and destroys
an OLDER, already unreachable torch.cuda.CUDAGraph (with its private pool, streams and events) abort the process?

Round 3 saw one process abort in the full GPU suite, at a capture after ~500 tests, kept no log of it, and wrapped every capture in
`ops.no_gc` on that hypothesis.  This probe makes the event deterministic, each case in a CHILD process that exits 0 when it
survives:

  graph     an older captured + replayed CUDAGraph made unreachable through a reference cycle; new capture opened; gc.collect()
            inside the capture (what an automatic collection at an unlucky allocation count would do)
  runner    the same with the objects a GraphRunner owns: four graphs sharing one pool, a side stream, events, static tensors
  tensors   only tensors of an older graph's private pool die inside the new capture (no graph object destroyed)
  control   the same allocation pattern with the collection BEFORE the capture opens (what torch.cuda.graph does on entry)

  python3 tools/micro/gc_capture_probe.py            -> one line per case: survived / died (exit code, last stderr line)
"""
import subprocess
import sys

CHILD = r"""
import gc, sys
import torch
case = sys.argv[1]
dev = torch.device("cuda")
gc.disable()

def work(x, w):
    for _ in range(4):
        x = torch.relu(x @ w)
    return x

class Holder:
    pass

def make(kind):
    h = Holder()
    h.me = h                                   # the reference cycle: only the cycle collector can free this
    h.x = torch.randn(256, 256, device=dev)
    h.w = torch.randn(256, 256, device=dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        work(h.x, h.w)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    h.g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(h.g, stream=side):
        h.y = work(h.x, h.w)
    h.g.replay()
    if kind == "runner":
        h.more = []
        for _ in range(3):
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, pool=h.g.pool(), stream=side):
                y2 = work(h.y, h.w)
            g2.replay()
            h.more.append((g2, y2))
        h.side, h.ev = side, [torch.cuda.Event() for _ in range(4)]
        for e in h.ev:
            e.record()
    if kind == "tensors":
        keep = h.g                              # the graph object itself stays alive; only its pool's tensors die
        h.g = None
        return h, keep
    torch.cuda.synchronize()
    return h, None

old, keep = make("graph" if case == "control" else case)
del old                                          # unreachable now, but alive until a cycle collection
x = torch.randn(256, 256, device=dev)
w = torch.randn(256, 256, device=dev)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    work(x, w)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
if case == "control":
    gc.collect()
# torch.cuda.graph.__enter__ collects by itself: open the capture by hand so that the garbage SURVIVES into it
torch.cuda.synchronize()
with torch.cuda.stream(side):
    g.capture_begin()
    y = work(x, w)
    if case != "control":
        n = gc.collect()                         # destroys the older graph / pool tensors / streams / events right here
    y = work(y, w)
    g.capture_end()
g.replay()
torch.cuda.synchronize()
print("survived", case, float(y.abs().sum()) > 0)
"""


def main():
    for case in ("control", "tensors", "graph", "runner"):
        r = subprocess.run([sys.executable, "-c", CHILD, case], capture_output=True, text=True, timeout=300)
        lines = r.stderr.strip().splitlines()
        why = [ln.strip()[:220] for ln in lines if any(k in ln for k in ("what()", "rror", "HIP", "hip", "capture"))][:4]
        print(f"{case:8s} exit {r.returncode:4d}  {'survived' if r.returncode == 0 else 'DIED'}  {r.stdout.strip()[-60:]}", flush=True)
        for ln in (why if r.returncode != 0 else []):
            print(f"         | {ln}", flush=True)


if __name__ == "__main__":
    main()
