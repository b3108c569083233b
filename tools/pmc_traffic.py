#!/usr/bin/env python3
"""HBM traffic per launch of the hand-written kernels at the benched model's dominant shapes ->
gpurun_out/r4_pmc_traffic_<model>.json (copy to profiles/; bench.py's roofline.traffic reads it).

For every (kernel, shape) two rocprofv3 runs of tools/kernel_once.py — FETCH_SIZE and WRITE_SIZE cannot share a pass —
each with --kernel-trace only and the program directly after `--`.  This driver never touches the GPU itself (plain
child processes).  Corrections of MI355X_MICROARCH.md (HBM section): both counters are in KB; on gfx950 FETCH_SIZE
reports half the bytes of wide coalesced streaming reads -> x 2; WRITE_SIZE is exact for 16-byte stores.
usage: pmc_traffic.py sd15|sd21|sdxl"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
E = 2  # bytes per element (f16 / bf16)


def conv_bytes(B, cin, hw, stride, cout):      # X + packed W + bias + residual + Y
    m = B * hw // (stride * stride)
    return E * (B * hw * cin + 9 * cin * cout + cout + 2 * m * cout)


def sa_bytes(B, H, N, D, bwd):                 # fwd: Q K V O (+ LSE f32); bwd: Q K V O dO dQ dK dV + LSE, delta
    c = H * D
    return E * B * N * c * (8 if bwd else 4) + 4 * B * H * N * (3 if bwd else 1)


def cap_bytes(B, H, N, D, bwd):
    c = H * D
    return E * ((3 if bwd else 2) * B * N * c + 2 * B * 77 * c + (N * 77 if bwd else B * H * N * 77))


def gn_bytes(B, C, HW, bwd):
    return E * B * C * HW * (3 if bwd else 2)


def lin_bytes(M, K, N, flags):                 # X + W + Y (GEGLU: half the columns) (+ residual) (+ the LayerNorm partial sums)
    n_out = N // 2 if flags & 1 else N
    return E * (M * K + N * K + M * n_out * (2 if flags & 4 else 1)) + (M * 5 * 8 if flags & 2 else 0)


# (entry point, kernel_once arguments, shape key of bench.py's census, algorithmic bytes, kernel-name filter)
CASES = {
    "sd15": [
        ("ga_conv3x3", ["conv", 3, 1280, 256, 1, 1280], dict(B=3, H=1280, N=256, D=1280), conv_bytes(3, 1280, 256, 1, 1280), "conv"),
        ("ga_conv3x3", ["conv", 3, 640, 1024, 1, 640], dict(B=3, H=640, N=1024, D=640), conv_bytes(3, 640, 1024, 1, 640), "conv"),
        ("ga_conv3x3", ["conv", 3, 320, 4096, 1, 320], dict(B=3, H=320, N=4096, D=320), conv_bytes(3, 320, 4096, 1, 320), "conv"),
        ("ga_conv3x3", ["conv", 1, 1280, 64, 1, 1280], dict(B=1, H=1280, N=64, D=1280), conv_bytes(1, 1280, 64, 1, 1280), "conv"),
        ("ga_linear", ["lin", 12288, 320, 2560, 3], dict(B=12288, H=320, N=0, D=2560), lin_bytes(12288, 320, 2560, 3), "linear_"),
        ("ga_linear", ["lin", 4096, 320, 960, 2], dict(B=4096, H=320, N=0, D=960), lin_bytes(4096, 320, 960, 2), "linear_"),
        ("ga_linear", ["lin", 256, 5120, 1280, 4], dict(B=256, H=5120, N=0, D=1280), lin_bytes(256, 5120, 1280, 4), "linear_"),
        ("ga_self_attn_fwd", ["sa_fwd", 3, 8, 4096, 40], dict(B=3, H=8, N=4096, D=40), sa_bytes(3, 8, 4096, 40, False), "self_attn_fwd"),
        ("ga_self_attn_fwd", ["sa_fwd", 1, 8, 4096, 40], dict(B=1, H=8, N=4096, D=40), sa_bytes(1, 8, 4096, 40, False), "self_attn_fwd"),
        ("ga_self_attn_bwd", ["sa_bwd", 1, 8, 4096, 40], dict(B=1, H=8, N=4096, D=40), sa_bytes(1, 8, 4096, 40, True), "self_attn_bwd"),
        ("ga_attn_capture_fwd", ["cap_fwd", 1, 8, 4096, 40], dict(B=1, H=8, N=4096, D=40), cap_bytes(1, 8, 4096, 40, False), "attn_capture_fwd"),
        ("ga_attn_capture_fwd", ["cap_fwd", 1, 8, 256, 160], dict(B=1, H=8, N=256, D=160), cap_bytes(1, 8, 256, 160, False), "attn_capture_fwd"),
        ("ga_attn_capture_bwd", ["cap_bwd", 1, 8, 256, 160], dict(B=1, H=8, N=256, D=160), cap_bytes(1, 8, 256, 160, True), "attn_capture_bwd"),
        ("ga_group_norm_fwd", ["gn_fwd", 1, 320, 4096], dict(B=1, H=32, N=4096, D=320), gn_bytes(1, 320, 4096, False), "gn_!bwd"),
        ("ga_group_norm_bwd", ["gn_bwd", 1, 320, 4096], dict(B=1, H=32, N=4096, D=320), gn_bytes(1, 320, 4096, True), "gn_+bwd"),
    ],
    "sd21": [
        ("ga_conv3x3", ["conv", 3, 1280, 576, 1, 1280], dict(B=3, H=1280, N=576, D=1280), conv_bytes(3, 1280, 576, 1, 1280), "conv"),
        ("ga_conv3x3", ["conv", 3, 320, 9216, 1, 320], dict(B=3, H=320, N=9216, D=320), conv_bytes(3, 320, 9216, 1, 320), "conv"),
        ("ga_conv3x3", ["conv", 3, 640, 2304, 1, 640], dict(B=3, H=640, N=2304, D=640), conv_bytes(3, 640, 2304, 1, 640), "conv"),
        ("ga_self_attn_fwd", ["sa_fwd", 3, 5, 9216, 64], dict(B=3, H=5, N=9216, D=64), sa_bytes(3, 5, 9216, 64, False), "self_attn_fwd"),
        ("ga_self_attn_fwd", ["sa_fwd", 1, 5, 9216, 64], dict(B=1, H=5, N=9216, D=64), sa_bytes(1, 5, 9216, 64, False), "self_attn_fwd"),
    ],
    "sdxl": [
        ("ga_conv3x3", ["conv", 3, 1280, 1024, 1, 1280, "bf16"], dict(B=3, H=1280, N=1024, D=1280), conv_bytes(3, 1280, 1024, 1, 1280), "conv"),
        ("ga_conv3x3", ["conv", 3, 320, 16384, 1, 320, "bf16"], dict(B=3, H=320, N=16384, D=320), conv_bytes(3, 320, 16384, 1, 320), "conv"),
        ("ga_conv3x3", ["conv", 3, 640, 4096, 1, 640, "bf16"], dict(B=3, H=640, N=4096, D=640), conv_bytes(3, 640, 4096, 1, 640), "conv"),
        ("ga_self_attn_fwd", ["sa_fwd", 3, 10, 4096, 64, "bf16"], dict(B=3, H=10, N=4096, D=64), sa_bytes(3, 10, 4096, 64, False), "self_attn_fwd"),
    ],
}


def one_pass(counter, args, tag):
    out = f"/tmp/pmc_traffic_{tag}_{counter}"
    shutil.rmtree(out, ignore_errors=True)
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "p", "--",
           sys.executable, str(ROOT / "tools" / "kernel_once.py"), *[str(a) for a in args]]
    r = subprocess.run(cmd, env=env, cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        print(r.stdout[-2000:], file=sys.stderr)
        raise SystemExit(f"rocprofv3 pass failed: {' '.join(cmd)}")
    per = defaultdict(float)      # (dispatch, kernel) -> counter value summed over the XCDs / instances
    for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                per[(row["Dispatch_Id"], row["Kernel_Name"])] += float(row["Counter_Value"])
    return per


def main():
    model = sys.argv[1] if len(sys.argv) > 1 else "sd15"
    doc = {"model": model, "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over tools/kernel_once.py "
           "(5 launches, inputs resident in HBM); bytes = FETCH_SIZE KB x 1024 x 2 (gfx950 counts 128-byte read requests as "
           "64) + WRITE_SIZE KB x 1024; a backward entry point = the sum of its launches; Infinity-Cache hits are counted, "
           "so bytes above the algorithmic figure are re-reads that left the XCD's L2, not necessarily HBM", "kernels": []}
    for i, (entry, args, shape, alg, flt) in enumerate(CASES[model]):
        vals = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            per = one_pass(counter, args, f"{model}{i}")
            by_kernel = defaultdict(list)
            for (_, k), v in per.items():
                inc, _, exc = flt.partition("!")
                inc, _, also = inc.partition("+")
                if inc in k and also in k and "pack" not in k and not (exc and exc in k):
                    by_kernel[k].append(v)
            # mean per launch of each kernel symbol, summed over the symbols of one entry-point call
            vals[counter] = sum(sum(v) / len(v) * (len(v) / 5.0 if len(v) % 5 == 0 else 1.0) for v in by_kernel.values())
            vals[counter + "_kernels"] = {k[:80]: round(sum(v) / len(v), 1) for k, v in by_kernel.items()}
        hbm = int(vals["FETCH_SIZE"] * 1024 * 2 + vals["WRITE_SIZE"] * 1024)
        doc["kernels"].append({"kernel": entry, "shape": shape, "args": [str(a) for a in args], "algorithmic_bytes": alg,
                               "FETCH_SIZE_KB": round(vals["FETCH_SIZE"], 1), "WRITE_SIZE_KB": round(vals["WRITE_SIZE"], 1),
                               "hbm_bytes_corrected": hbm, "ratio_to_algorithmic": round(hbm / alg, 2),
                               "per_symbol_KB": {"fetch": vals["FETCH_SIZE_kernels"], "write": vals["WRITE_SIZE_kernels"]}})
        print(entry, shape, "alg", alg, "measured", hbm, f"x{hbm / alg:.2f}", flush=True)
    dst = ROOT / "gpurun_out" / f"r4_pmc_traffic_{model}.json"
    dst.parent.mkdir(exist_ok=True)
    dst.write_text(json.dumps(doc, indent=1))
    print("wrote", dst)


if __name__ == "__main__":
    main()
