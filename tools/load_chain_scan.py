#!/usr/bin/env python3
"""Counts DEPENDENT-LOAD CHAINS in the gfx950 code objects of libga_hip.so: a vector-memory load that is followed by an
`s_waitcnt vmcnt(0)` before the next vector-memory load of the same straight-line stretch is issued — the load is waited
for alone, and every such pair in a row is one more serial round trip to memory (0.5 - 2 us each on the MI355X).

Why this exists (round 3): hipcc turns `x[k] = cond ? load(p) : 0` / `if (cond) x[k] = load(p)` over a register array into
load, wait, copy per element (each load sits in a block of its own and the value is merged behind a wait), and it converts a
freshly loaded per-channel constant right behind its load.  The Linear epilogue (4 bias loads + 1 residual load per output
vector, each waited for on its own), the convolution epilogue (2 per output vector: 8 - 16 in a row), the small-slab GroupNorm
(bias, 4 pieces per round trip, gamma / beta behind the reduction), the Q / dO / O fragment loads of the attention kernels
(NK in a row, 30 in the dQ kernel at head size 160), the upstream map gradient of the capture backward (4 NT in a row) and
the head-map mean of the loss launch (32 in a row) all had this shape; the fix each time is unconditional loads from
clamped addresses, issued in one batch, with the select where the value is consumed.

  python tools/load_chain_scan.py [lib] [--min N] [--json out.json]
      prints kernels whose longest run of serial steps (see chains()) is >= N (default 3), longest first."""
import json
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
from code_object_check import extract_code_objects  # noqa: E402

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
LOAD = re.compile(r"^\s*(global_load|buffer_load|flat_load)\w*\s")
LDS_DMA = re.compile(r"\blds\b")
WAITN = re.compile(r"^\s*s_waitcnt\b.*vmcnt\((\d+)\)")


def chains(lib):
    """{kernel: (longest run of serial steps, serial steps, loads)}.  A SERIAL STEP is a wait that retires a load while at
    most two loads were outstanding (the compiler's two usual shapes: load, wait vmcnt(0), load, ...  and the depth-two
    form load, load, wait vmcnt(1), load, wait vmcnt(1), ...): the next load is issued only after a round trip.  A run ends
    where three or more loads go out between two waits (a batch) or a wait finds more than two outstanding."""
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for co in extract_code_objects(lib, tmp):
            text = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", str(co)], capture_output=True, text=True, check=True).stdout
            func, pending, since, run, best, total, loads = None, 0, 0, 0, 0, 0, 0
            for line in text.splitlines():
                if line.endswith(">:"):
                    if func is not None:
                        out[func] = (max(best, run), total, loads)
                    func, pending, since, run, best, total, loads = line.split("<")[1][:-2], 0, 0, 0, 0, 0, 0
                    continue
                body = line.split("//")[0]
                if LOAD.match(body) and not LDS_DMA.search(body):
                    loads += 1
                    pending += 1
                    since += 1
                    if since > 2:
                        best, run = max(best, run), 0
                    continue
                m = WAITN.match(body)
                if m:
                    left = int(m.group(1))
                    if left < pending:
                        if pending <= 2:
                            run += 1
                            total += 1
                        else:
                            best, run = max(best, run), 0
                        pending = left
                    since = 0
            if func is not None:
                out[func] = (max(best, run), total, loads)
    return out


def main():
    argv, minimum, out_json, args = sys.argv[1:], 3, None, []
    while argv:
        a = argv.pop(0)
        if a == "--min":
            minimum = int(argv.pop(0))
        elif a == "--json":
            out_json = argv.pop(0)
        else:
            args.append(a)
    lib = args[0] if args else str(ROOT / "guided-attention_amd" / "libga_hip.so")
    res = chains(lib)
    rows = sorted(((v[0], v[1], v[2], k) for k, v in res.items() if v[0] >= minimum), reverse=True)
    print(f"{len(res)} kernels, {len(rows)} with a run of >= {minimum} serial load steps")
    for run, total, loads, name in rows:
        print(f"  run {run:3d}  serial {total:3d} of {loads:3d} loads  {name[:150]}")
    if out_json:
        Path(out_json).write_text(json.dumps({k: {"longest_run": v[0], "serial_loads": v[1], "loads": v[2]}
                                                                            for k, v in res.items()}, indent=1))
    return 0


if __name__ == "__main__":
    sys.exit(main())
