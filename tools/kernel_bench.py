#!/usr/bin/env python3
"""Per-shape launch time of the capture kernels on the SD-1.x layer shapes (GPU box only):
back-to-back replay between two HIP events, algorithmic bytes / time vs the 8 TB/s HBM peak."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from guided_attention_amd import ops  # noqa: E402


def capture_bytes(key):
    kind, B, H, N, Kt, D, flag, dt = key
    esz, C = 2, H * D
    kv = 2 * B * Kt * C
    if kind == "attn_capture_fwd":
        return esz * (2 * B * N * C + kv + (B * H * N * Kt if flag else 0))
    return esz * (3 * B * N * C + kv + (N * Kt if flag else 0))


def main():
    ops.load()
    print(f"{'kernel':18s} {'B':>2s} {'N':>5s} {'D':>4s} {'flag':>5s} {'us':>8s} {'GB/s':>8s} {'frac':>6s}")
    for B in (1, 2):
        for N, D in ((4096, 40), (1024, 80), (256, 160), (64, 160)):
            for kind, flag in (("attn_capture_fwd", False), ("attn_capture_fwd", True), ("attn_capture_bwd", False),
                               ("attn_capture_bwd", True)):
                if B == 2 and kind == "attn_capture_bwd":
                    continue
                key = (kind, B, 8, N, 77, D, flag, "torch.float16")
                us = ops.replay_launch_us(key, 300)
                gbs = capture_bytes(key) / us / 1e3
                print(f"{kind:18s} {B:2d} {N:5d} {D:4d} {str(flag):>5s} {us:8.2f} {gbs:8.1f} {gbs / 8000:6.3f}", flush=True)


if __name__ == "__main__" and "--self-attn" not in sys.argv and "--group-norm" not in sys.argv:
    main()


def self_attn_bench():
    import torch
    print(f"\n{'self-attn':12s} {'B':>2s} {'N':>5s} {'D':>4s} {'fwd us':>8s} {'TF/s':>7s} {'bwd us':>8s} {'TF/s':>7s}   (SDPA fwd us, bwd us)")
    for B, N, D in ((1, 4096, 40), (2, 4096, 40), (3, 4096, 40), (1, 1024, 80), (2, 1024, 80), (3, 1024, 80)) + (() if "--no160" in sys.argv else ((1, 256, 160), (1, 64, 160))):
        H = 8
        q, k, v, do = (torch.randn(B, N, H * D, device="cuda", dtype=torch.half) for _ in range(4))
        o, lse = ops.self_attn_fwd(q, k, v, H, D ** -0.5)

        def timed(fn, iters=50):
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    fn()
                side.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    for _ in range(iters):
                        fn()
                g.replay()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                side.synchronize()
                e0.record(side)
                g.replay()
                e1.record(side)
                side.synchronize()
            torch.cuda.current_stream().wait_stream(side)
            return e0.elapsed_time(e1) * 1e3 / iters

        f_us = timed(lambda: ops.self_attn_fwd(q, k, v, H, D ** -0.5))
        try:
            b_us = timed(lambda: ops.self_attn_bwd(q, k, v, o, do, lse, H, D ** -0.5))
        except Exception as e:  # experiment builds may not fit every shape
            b_us = float("nan")
        qh, kh, vh = (t.view(B, N, H, D).transpose(1, 2).detach().requires_grad_(True) for t in (q, k, v))
        def eager(fn, iters=30):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / iters

        with torch.no_grad():
            sf = eager(lambda: torch.nn.functional.scaled_dot_product_attention(qh, kh, vh))
        so = torch.nn.functional.scaled_dot_product_attention(qh, kh, vh)
        gout = torch.randn_like(so)
        sb = eager(lambda: torch.autograd.grad(so, [qh, kh, vh], gout, retain_graph=True))
        fl = 4.0 * B * H * N * N * D
        print(f"{'':12s} {B:2d} {N:5d} {D:4d} {f_us:8.1f} {fl / f_us / 1e6:7.1f} {b_us:8.1f} {2.5 * fl / b_us / 1e6:7.1f}   ({sf:.1f}, {sb:.1f})",
              flush=True)


if __name__ == "__main__" and "--self-attn" in sys.argv:
    self_attn_bench()


def group_norm_bench():
    print(f"\n{'group_norm':12s} {'B':>2s} {'C':>5s} {'HW':>5s} {'fwd us':>8s} {'GB/s':>7s} {'bwd us':>8s} {'GB/s':>7s}")
    for B, C, HW in ((1, 320, 4096), (2, 320, 4096), (1, 640, 4096), (1, 960, 4096), (1, 640, 1024), (1, 1280, 1024),
                     (1, 1920, 1024), (1, 1280, 256), (1, 2560, 256), (1, 1280, 64), (1, 2560, 64)):
        f = ops.replay_launch_us(("group_norm_fwd", B, 32, HW, 0, C, True, "torch.float16"))
        b = ops.replay_launch_us(("group_norm_bwd", B, 32, HW, 0, C, True, "torch.float16"))
        nbytes = 2 * B * C * HW
        print(f"{'':12s} {B:2d} {C:5d} {HW:5d} {f:8.2f} {2 * nbytes / f / 1e3:7.1f} {b:8.2f} {3 * nbytes / b / 1e3:7.1f}", flush=True)


if __name__ == "__main__" and "--group-norm" in sys.argv:
    group_norm_bench()
