#!/usr/bin/env python3
"""Per-shape time of the UNet's 3x3 convolutions: the library (MIOpen, benchmark mode, channels-last fp16) against
ga_conv3x3_nhwc for every (tile, split-K) plan — forward; the stride-1 backward-to-input is the same kernel with
Cin / Cout swapped, listed as its own shape.  hipGraph replay timing.  Prints a table and a JSON line.

  conv_tune.py [batches, default 1,3] [cold] [small] [sd21 | sdxl] [--write] [variants [all]]
     cold    : own-kernel launches rotate over enough packed-weight copies to exceed the 256 MB Infinity Cache — in the
               pipeline a convolution's weights are cold (1.7 GB of UNet weights stream through between two uses) while its
               input was just written; warm timings favour plans with too few bytes in flight per CU
     --write : update guided-attention_amd/conv_plans.json {"M,Cin,Cout,stride": [bm, bn, splits]} with the best plans"""
import json
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from guided_attention_amd import ops  # noqa: E402

torch.backends.cudnn.benchmark = True
BASE_ALL = [  # (Cin, Cout, H, stride)
    (320, 320, 64, 1), (640, 320, 64, 1), (960, 320, 64, 1), (320, 640, 64, 1), (320, 960, 64, 1), (320, 320, 64, 2), (640, 640, 64, 1),
    (320, 640, 32, 1), (640, 640, 32, 1), (1280, 640, 32, 1), (1920, 640, 32, 1), (960, 640, 32, 1), (640, 1280, 32, 1),
    (640, 1920, 32, 1), (640, 960, 32, 1), (640, 640, 32, 2), (1280, 1280, 32, 1),
    (640, 1280, 16, 1), (1280, 1280, 16, 1), (2560, 1280, 16, 1), (1920, 1280, 16, 1), (1280, 2560, 16, 1), (1280, 1920, 16, 1),
    (1280, 1280, 16, 2), (1280, 1280, 8, 1), (2560, 1280, 8, 1), (1280, 2560, 8, 1),
]


# "variants" runs compare extra builds on a few representative shapes only
BASE = BASE_ALL if "variants" not in sys.argv or "all" in sys.argv else [(320, 320, 64, 1), (640, 640, 32, 1), (1280, 1280, 16, 1), (1280, 1280, 8, 1)]
if "small" in sys.argv:        # the weight-bound levels only (16 x 16 and 8 x 8 maps)
    BASE = [b for b in BASE_ALL if b[2] <= 16]
if "sd21" in sys.argv:         # BASELINE config 4: the same UNet on 96 x 96 latents (768^2 images): maps of 96 / 48 / 24 / 12
    BASE = [(ci, co, h * 3 // 2, st) for ci, co, h, st in BASE]
if "sdxl" in sys.argv:         # BASELINE config 5: 128 x 128 latents, three levels (320 / 640 / 1280 channels at 128 / 64 / 32)
    BASE = [(320, 320, 128, 1), (640, 320, 128, 1), (960, 320, 128, 1), (320, 960, 128, 1), (320, 640, 128, 1), (320, 320, 128, 2),
            (320, 640, 64, 1), (640, 640, 64, 1), (1280, 640, 64, 1), (1920, 640, 64, 1), (960, 640, 64, 1), (640, 1280, 64, 1),
            (640, 1920, 64, 1), (640, 960, 64, 1), (640, 640, 64, 2), (640, 640, 128, 1),
            (640, 1280, 32, 1), (1280, 1280, 32, 1), (2560, 1280, 32, 1), (1920, 1280, 32, 1), (1280, 2560, 32, 1),
            (1280, 1920, 32, 1), (1280, 1280, 64, 1)]


def replay_us(fn, iters=20):
    s = ops.side_stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(iters):
                fn()
        g.replay()
        s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(3):
            g.replay()
        e1.record(s)
        e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * iters)


def bind(path):
    import ctypes
    lib = ctypes.CDLL(str(path))
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
    lib.ga_conv3x3_pack_weights.argtypes = [vp, vp, i32, i32, i64, i64, i64, i64, i32, i32, vp]
    lib.ga_conv3x3_nhwc.argtypes = [vp, vp, vp, vp, vp, vp, vp] + [i32] * 10 + [vp]
    return lib


def variant_times(lib, x, w, co, st, plans):
    """{plan: us} for one extra build of conv3x3.hip (tools/micro/sa_variants/libconv_*.so), same inputs."""
    import ctypes
    B, ci, h, _ = x.shape
    ho = (h - 1) // st + 1
    wp = torch.empty(9 * (-(-co // 64) * 64) * ci, device="cuda", dtype=x.dtype)      # the blocked pack (include/ga_hip.h)
    sp_ = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
    so, si, sy, sx = w.stride()
    assert lib.ga_conv3x3_pack_weights(P(w), P(wp), co, ci, so, si, sy, sx, 0, 0, sp_()) == 0
    y = torch.empty(B, co, ho, ho, device="cuda", dtype=x.dtype).contiguous(memory_format=torch.channels_last)
    out = {}
    for (bm, bn, sp) in plans:
        def call():
            ws, tk = ops.splitk_workspace(x.device, B * ho * ho, co, bm, bn, sp)   # the scratch of the stream this runs on
            rc = lib.ga_conv3x3_nhwc(P(x), P(wp), P(y), P(ws), P(tk), None, None, B, h, h, ci, co, st, bm, bn, sp, 0, sp_())
            assert rc == 0, rc
        out[(bm, bn, sp)] = replay_us(call, iters=10)
    return out


def main():
    batches = [int(b) for b in (sys.argv[1] if len(sys.argv) > 1 and sys.argv[1][0].isdigit() else "1,3").split(",")]
    vdir = Path(__file__).resolve().parent / "micro" / "sa_variants"
    variants = {p.stem[3:]: bind(p) for p in sorted(vdir.glob("libconv_*.so"))} if "variants" in sys.argv else {}
    table = {}
    print(f"{'B':>2} {'Cin':>5} {'Cout':>5} {'HW':>3} {'s':>1} {'lib us':>8} {'TF/s':>6} | {'best plan':>14} {'us':>8} {'TF/s':>6} {'x':>5} | heuristic")
    for B in batches:
        for ci, co, h, st in BASE:
            x = torch.randn(B, ci, h, h, device="cuda", dtype=torch.half).contiguous(memory_format=torch.channels_last)
            w = (torch.randn(co, ci, 3, 3, device="cuda", dtype=torch.half) * 0.02).contiguous(memory_format=torch.channels_last)
            ho = (h - 1) // st + 1
            flop = 2.0 * B * ho * ho * co * ci * 9
            t_lib = replay_us(lambda: F.conv2d(x, w, None, stride=st, padding=1))
            wp = ops.conv3x3_packed_weights(w, False)
            n_copies = max(1, min(48, -(-320 * 2 ** 20 // (9 * co * ci * 2)))) if "cold" in sys.argv else 1
            wps = [wp] + [wp.clone() for _ in range(n_copies - 1)]
            turn = [0]

            def nxt():
                turn[0] += 1
                return wps[turn[0] % n_copies]
            steps = 9 * ci // 64
            res = {}
            for bm, bn in ((128, 128), (128, 64), (64, 64)):
                for sp in (1, 2, 3, 4, 6, 8, 12, 16):
                    if steps // sp < 3:
                        continue
                    tiles = -(-B * ho * ho // bm) * -(-co // bn)
                    if tiles * sp > 4096 or (tiles * sp < 96 and sp < 16):
                        continue
                    if sp > 1 and sp * tiles * bm * bn > ops.LIN_SLAB_FLOATS:
                        continue
                    plan = (bm, bn, sp, 0)
                    res[(bm, bn, sp)] = replay_us(lambda: ops.conv3x3_nhwc(x, nxt(), co, st, None, None, plan=plan), iters=max(10, min(n_copies, 40)))
            best = min(res, key=res.get)
            extra = ""
            for vname, vlib in variants.items():
                vt = variant_times(vlib, x, w, co, st, [k for k in res if (9 * ci // 64) // k[2] >= 3])
                vb = best if best in vt else min(vt, key=vt.get)      # the same plan as the product build's best
                extra += f" | {vname} {vt[vb]:.1f}"
            heur = ops.conv3x3_plan(B, h, h, ci, co, st)[:3]
            t_h = res.get(tuple(heur))
            print(f"{B:>2} {ci:>5} {co:>5} {h:>3} {st:>1} {t_lib:8.1f} {flop / t_lib / 1e6:6.0f} | {str(best):>14} {res[best]:8.1f} "
                  f"{flop / res[best] / 1e6:6.0f} {t_lib / res[best]:5.2f} | {heur} {t_h if t_h is None else round(t_h, 1)}{extra}", flush=True)
            table[f"{B},{ci},{co},{h},{st}"] = {"lib_us": round(t_lib, 1), "best": list(best), "best_us": round(res[best], 1),
                                                "all": {f"{k[0]}x{k[1]}x{k[2]}": round(v, 1) for k, v in res.items()}}
    print(json.dumps(table))
    if "--write" in sys.argv:
        path = Path(__file__).resolve().parent.parent / "guided-attention_amd" / "conv_plans.json"
        plans = json.loads(path.read_text()) if path.exists() else {}
        for key, row in table.items():
            B, ci, co, h, st = (int(v) for v in key.split(","))
            ho = (h - 1) // st + 1
            plans[f"{B * ho * ho},{ci},{co},{st}"] = row["best"]
        path.write_text(json.dumps(plans, indent=0, sort_keys=True))


if __name__ == "__main__":
    main()
