#!/usr/bin/env python3
"""Kernel-time shares by category from a rocprofv3 `*_kernel_stats.csv`.  usage: stats_categories.py kernel_stats.csv"""
import csv
import sys


def cat(n):
    if "conv3x3_patch_kernel" in n: return "ga conv3x3 (patch variant)"
    if "conv3x3_kernel" in n: return "ga conv3x3 (per-tap variant)"
    if "conv_splitk" in n: return "ga conv split-K sum"
    if "linear_kernel" in n: return "ga Linear (LayerNorm / GEGLU / residual folded in)"
    if "conv_pack" in n: return "ga conv weight pack"
    if n.startswith("Cijk") or n.startswith("Custom_Cijk"): return "hipBLASLt GEMM"
    if "igemm" in n or "ck16tensor" in n or "ck::" in n or "conv" in n.lower(): return "MIOpen/CK conv"
    if "self_attn" in n: return "ga self-attention"
    if "gn_" in n: return "ga GroupNorm"
    if "attn_capture" in n or "attn_scores" in n: return "ga cross-attention capture"
    if "add_ln" in n: return "ga residual+LayerNorm"
    if "geglu" in n: return "ga GEGLU"
    if "bias_residual" in n: return "ga bias+residual"
    if "cat_rows" in n: return "ga channel concatenation"
    if "smooth_loss" in n or "aggregate" in n or "axp" in n or "cfg_ddim" in n: return "ga loss / aggregate / latent ops"
    if "elementwise" in n or "CatArray" in n or "upsample" in n.lower() or "copy" in n.lower() or "fill" in n.lower():
        return "torch element-wise / cat / copy / upsample"
    if "SubTensor" in n: return "MIOpen SubTensorOp"
    return "other"


rows = [r for r in csv.DictReader(open(sys.argv[1])) if not r["Name"].startswith("naive_conv")]
tot = sum(int(r["TotalDurationNs"]) for r in rows)
cats, calls = {}, {}
for r in rows:
    c = cat(r["Name"])
    cats[c] = cats.get(c, 0) + int(r["TotalDurationNs"])
    calls[c] = calls.get(c, 0) + int(r["Calls"])
print(f"| category | total ms | % | launches | avg us |\n|---|---:|---:|---:|---:|")
for c, v in sorted(cats.items(), key=lambda kv: -kv[1]):
    print(f"| {c} | {v / 1e6:.1f} | {100 * v / tot:.2f} | {calls[c]} | {v / calls[c] / 1e3:.2f} |")
print(f"| **total** | {tot / 1e6:.1f} | 100 | {sum(calls.values())} | |")
