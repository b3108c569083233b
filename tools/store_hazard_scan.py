#!/usr/bin/env python3
"""Scans the gfx950 code objects of libga_hip.so for a wide VMEM store (buffer/global/flat store of 3 - 4 dwords) whose data
registers are overwritten by one of the next `--window` instructions.  The toolchain leaves no wait state there when the
store carries an SGPR offset; on the MI355X such a pair was seen to store the NEW value from a few lanes (round 3: the split-K
slab store of linear_kernel<128, 64, 3>, one tile in a few hundred wrong in rows 12 - 15 / 28 - 31 of one accumulator dword).
The kernels keep the stored registers live across the following wait instead; tests/test_abi.py runs this scan.

  python tools/store_hazard_scan.py [lib] [--window N]   -> prints offenders, exit code 1 if any"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
from code_object_check import extract_code_objects  # noqa: E402

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
STORE = re.compile(r"^\s*(buffer_store_dwordx[34]|global_store_dwordx[34]|flat_store_dwordx[34]|scratch_store_dwordx[34])\s+(.*)$")
VREG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")


def regs(tok):
    m = VREG.fullmatch(tok.strip())
    if not m:
        return set()
    if m.group(3) is not None:
        return {int(m.group(3))}
    return set(range(int(m.group(1)), int(m.group(2)) + 1))


def store_data(mnemonic, ops):
    parts = [p.strip() for p in ops.split(",")]
    if mnemonic.startswith("buffer"):
        return regs(parts[0])
    return regs(parts[1]) if len(parts) > 1 else set()     # global / flat / scratch: vaddr, vdata, ...


def written(line):
    """VGPRs an instruction writes (first operand of VALU / load instructions; stores and waits write none)."""
    line = line.split("//")[0].strip()
    if not line or line.startswith(("s_", "buffer_store", "global_store", "flat_store", "ds_write", "scratch_store", ";")):
        return set()
    parts = line.split(None, 1)
    if len(parts) < 2:
        return set()
    return regs(parts[1].split(",")[0])


def write_through_offenders(lib, window=2):
    """The observed failure class only: an sc1 (write-through) buffer store overwritten by a VALU instruction."""
    return [o for o in offenders(lib, window) if "sc1" in o[1] and o[2].startswith("v_")]


def offenders(lib, window=2):
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        for co in extract_code_objects(lib, tmp):
            text = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", str(co)], capture_output=True, text=True, check=True).stdout
            func, lines = None, text.splitlines()
            for n, line in enumerate(lines):
                if line.endswith(">:"):
                    func = line.split("<")[1][:-2]
                    continue
                body = line.split("//")[0]
                m = STORE.match(body)
                if not m:
                    continue
                data = store_data(m.group(1), m.group(2))
                seen = 0
                for nxt in lines[n + 1:]:
                    b = nxt.split("//")[0].strip()
                    if not b:
                        continue
                    if b.startswith(("s_nop", "s_waitcnt")) or nxt.endswith(">:"):
                        break
                    hit = written(b) & data
                    if hit:
                        out.append((func, body.strip(), b))
                        break
                    seen += 1
                    if seen >= window:
                        break
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    window = int(sys.argv[sys.argv.index("--window") + 1]) if "--window" in sys.argv else 2
    if "--window" in sys.argv:
        args = [a for a in args if a != str(window)]
    lib = args[0] if args else str(ROOT / "guided-attention_amd" / "libga_hip.so")
    bad = offenders(lib, window)
    for f, s, w in bad:
        print(f"{f}\n    {s}\n    {w}")
    print(f"{len(bad)} wide store(s) with their data registers overwritten within {window} instruction(s)")
    sys.exit(1 if bad else 0)
