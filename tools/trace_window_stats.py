#!/usr/bin/env python3
"""Per-kernel and per-category time of the TIMED IMAGES ONLY from a rocprofv3 --kernel-trace CSV of bench.py:
the window runs from the end of the last warm-up image to the end of the last timed image, found by counting the
`cfg_ddim_step` launches (one per CFG pass: `per_image` of them per image — unet_calls_per_image.fwd_b2).  Warm-up images,
hipGraph capture, library algorithm searches and the roofline's micro-replays all fall outside the window.

  trace_window_stats.py <kernel_trace.csv> <warmup images> <timed images> <cfg passes per image> <out.md> [title]
  trace_window_stats.py <kernel_trace.csv> marker <kernel substring> <replays> <out.md> [title]
      the window after the LAST launch whose name holds the substring (tools/unet_bench.py only=<pass> launches a latent
      axpby as the marker right before its timed replays): the kernels of `replays` replays of one captured pass"""
import csv
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from summarize_stats import short  # noqa: E402


def cat(n):
    if "conv3x3_patch" in n: return "ga conv3x3 (patch variants)"
    if "conv3x3_kernel" in n: return "ga conv3x3 (per-tap variant)"
    if "conv_pack" in n: return "ga conv weight pack"
    if "linear_kernel" in n or "linear_stream_kernel" in n: return "ga Linear (LayerNorm / GEGLU / residual folded in)"
    if n.startswith("Cijk") or n.startswith("Custom_Cijk"): return "hipBLASLt GEMM"
    if "igemm" in n or "ck16tensor" in n or "ck::" in n or "naive_conv" in n or "Conv" in n: return "MIOpen/CK conv"
    if "self_attn" in n: return "ga self-attention"
    if "gn_" in n: return "ga GroupNorm"
    if "attn_capture" in n or "attn_scores" in n: return "ga cross-attention capture"
    if "add_ln" in n: return "ga LayerNorm backward"
    if "geglu" in n: return "ga GEGLU backward"
    if "bias_residual" in n: return "ga bias+residual"
    if "cat_rows" in n: return "ga channel concatenation"
    if "smooth_loss" in n or "aggregate" in n or "axp" in n or "cfg_ddim" in n: return "ga loss / aggregate / latent ops"
    if "elementwise" in n or "CatArray" in n or "upsample" in n.lower() or "copy" in n.lower() or "fill" in n.lower():
        return "torch element-wise / cat / copy / upsample"
    if "SubTensor" in n: return "MIOpen SubTensorOp"
    return "other"


def main():
    path, out = sys.argv[1], sys.argv[5]
    title = sys.argv[6] if len(sys.argv) > 6 else "kernel time inside the timed images"
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    if sys.argv[2] == "marker":
        marks = [i for i, r in enumerate(rows) if sys.argv[3] in r[2]]
        if not marks:
            raise SystemExit(f"no launch named *{sys.argv[3]}* in the trace")
        timed = int(sys.argv[4])
        win = rows[marks[-1] + 1:]
        header = f"Window: {timed} replays of the captured pass (everything after the marker launch), {len(win)} launches"
        unit = "replay"
    else:
        warm, timed, per_image = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
        marks = [i for i, r in enumerate(rows) if "cfg_ddim" in r[2]]
        need = (warm + timed) * per_image
        if len(marks) < need:
            raise SystemExit(f"only {len(marks)} cfg_ddim launches in the trace, expected at least {need}")
        lo = marks[warm * per_image - 1] + 1 if warm else 0
        hi = marks[need - 1]
        win = rows[lo:hi + 1]
        header = (f"Window: {timed} timed image(s) after {warm} warm-up image(s) ({per_image} CFG passes per image), "
                  f"{len(win)} launches")
        unit = "image"
    span = win[-1][1] - win[0][0]
    per, cats = {}, {}
    for s, e, n in win:
        d = per.setdefault(n, [0, 0, 1 << 62, 0])
        d[0] += 1
        d[1] += e - s
        d[2] = min(d[2], e - s)
        d[3] = max(d[3], e - s)
        c = cats.setdefault(cat(n), [0, 0])
        c[0] += 1
        c[1] += e - s
    total = sum(d[1] for d in per.values())
    lines = [f"# {title}", "",
             f"{header}, span {span / 1e6:.1f} ms = {span / 1e6 / timed:.3f} ms per {unit}, kernel time {total / 1e6:.1f} ms "
             f"({100 * total / span:.1f} % of the span; durations under the profiler read high).", "",
             "| category | total ms | % of kernel time | launches | avg us |", "|---|---:|---:|---:|---:|"]
    for c, (n, t) in sorted(cats.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"| {c} | {t / 1e6:.1f} | {100 * t / total:.2f} | {n} | {t / n / 1e3:.2f} |")
    lines += ["", "| kernel | calls | total ms | % | avg us | min us | max us |", "|---|---:|---:|---:|---:|---:|---:|"]
    for n, (c, t, mn, mx) in sorted(per.items(), key=lambda kv: -kv[1][1])[:60]:
        lines.append(f"| {short(n)} | {c} | {t / 1e6:.1f} | {100 * t / total:.2f} | {t / c / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} |")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:24]))


if __name__ == "__main__":
    main()
