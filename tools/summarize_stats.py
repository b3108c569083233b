#!/usr/bin/env python3
"""Condense a rocprofv3 `*_kernel_stats.csv` into a small markdown table (top kernels by total time, the
hand-written ga_* kernels always listed) for profiles/.  usage: summarize_stats.py kernel_stats.csv out.md [title]"""
import csv
import re
import sys

OURS = ("attn_capture", "smooth_loss", "aggregate_kernel", "axpy", "axpby", "cfg_ddim", "self_attn", "group_norm", "gn_", "conv3x3", "conv_splitk", "geglu", "add_ln", "bias_residual")


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("at::native::", "").replace("(anonymous namespace)::", "")
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9]+kernel)I(.*?)EEv", name)
    if m:
        args = m.group(2).replace("DF16_", "f16,").replace("N2ga6bf16_tE", "bf16,").replace("Li", "").replace("E", ",")
        return f"ga::{m.group(1)}<{args.strip(',')}>"
    return re.sub(r"\(.*", "", name)[:90]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else src
    rows = list(csv.DictReader(open(src)))
    find = [r for r in rows if r["Name"].startswith("naive_conv")]
    rows = [r for r in rows if not r["Name"].startswith("naive_conv")]
    total = sum(int(r["TotalDurationNs"]) for r in rows)
    calls = sum(int(r["Calls"]) for r in rows)
    out = [f"# {title}", "",
           f"`rocprofv3 --kernel-trace --stats`; {len(rows)} kernel symbols, {calls} launches, {total / 1e9:.3f} s of kernel time "
           f"(MIOpen's one-time `naive_conv_*` find/verification kernels excluded: {sum(int(r['TotalDurationNs']) for r in find) / 1e9:.2f} s).", "",
           "| kernel | calls | total ms | % | avg us | min us | max us |", "|---|---:|---:|---:|---:|---:|---:|"]
    acc = 0
    for i, r in enumerate(rows):
        d = int(r["TotalDurationNs"])
        ours = any(k in r["Name"] for k in OURS)
        if i < 40 or ours:
            out.append(f"| {'**' if ours else ''}{short(r['Name'])}{'**' if ours else ''} | {r['Calls']} | {d / 1e6:.1f} | "
                       f"{100 * d / total:.2f} | {float(r['AverageNs']) / 1e3:.2f} | {int(r['MinNs']) / 1e3:.2f} | {int(r['MaxNs']) / 1e3:.2f} |")
        acc += d
    open(dst, "w").write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
