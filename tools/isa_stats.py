#!/usr/bin/env python3
"""Instruction-class histogram per basic block of one kernel in a hipcc -S listing (blocks that contain MFMAs only).
  python tools/isa_stats.py file.s 'self_attn_fwd_kernelIDF16_Li3ELi1ELi2ELi128'"""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt")):
        return "trans"
    if op.startswith("v_pk_"):
        return "valu_pk"
    if op.startswith("v_cvt"):
        return "cvt"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "lds_rd"
    if op.startswith("ds_"):
        return "lds_wr"
    if op.startswith(("buffer_load", "global_load", "flat_load")):
        return "vm_ld"
    if op.startswith(("buffer_store", "global_store", "flat_store")):
        return "vm_st"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(pat) + r"\S*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    blocks, cur, name = [], Counter(), "entry"
    ops = Counter()
    for l in lines[start + 1:end + 1]:
        t = l.strip()
        if not t or t.startswith((";", "//")):
            continue
        m = re.match(r"^(\.LBB\S+):", t)
        if m:
            blocks.append((name, cur, ops))
            cur, ops, name = Counter(), Counter(), m.group(1)
            continue
        if t.startswith("."):
            continue
        op = t.split()[0]
        cur[classify(op)] += 1
        ops[op] += 1
    blocks.append((name, cur, ops))
    for meta in ("vgpr_count", "sgpr_count", "NumVgprs", "ScratchSize", "Occupancy", "LDSByteSize"):
        for l in lines[end:end + 120]:
            if meta in l:
                print(l.strip())
                break
    for name, c, ops in blocks:
        if c["mfma"] >= 4:
            tot = sum(c.values())
            print(f"{name}: {tot} instr  " + "  ".join(f"{k}={v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1])))
            if "-v" in sys.argv:
                print("    " + "  ".join(f"{k}:{v}" for k, v in sorted(ops.items(), key=lambda kv: -kv[1])[:40]))


if __name__ == "__main__":
    main()
