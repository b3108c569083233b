#!/usr/bin/env python3
"""Per-shape time of the UNet's Linear layers / 1x1 convolutions at COLD weights: torch.nn.functional.linear (hipBLASLt)
against ga_linear_fused for every (tile, split-K) plan.  In the pipeline a layer's weights are cold (1.7 GB of UNet weights
cycle through the 256 MB Infinity Cache between two uses) while its input was just written: both contenders rotate over
enough weight copies to exceed the cache.  hipGraph replay timing between two HIP events.

  python tools/linear_tune.py [batches, default 1,3] [--write]     # --write: guided-attention_amd/linear_plans.json
"""
import json
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from guided_attention_amd import ops  # noqa: E402
from gemm_tune import BASE  # noqa: E402


def replay_us(fn, iters=20, reps=3):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    best = None
    with torch.cuda.stream(side):
        for _ in range(2):
            fn()
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(iters):
                fn()
        g.replay()
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            side.synchronize()
            e0.record(side)
            g.replay()
            e1.record(side)
            side.synchronize()
            t = e0.elapsed_time(e1) * 1e3 / iters
            best = t if best is None else min(best, t)
    torch.cuda.current_stream().wait_stream(side)
    return best


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    batches = [int(b) for b in (args[0] if args else "1,3").split(",")]
    dev = torch.device("cuda")
    ops.prepare_device(dev)
    table, plans, wins = {}, {}, 0
    print(f"{'M':>6} {'K':>5} {'N':>6} {'lib us':>8} {'TF/s':>6} | {'best plan':>14} {'us':>8} {'TF/s':>6} {'x':>5}")
    for B in batches:
        for tok, K, N in BASE:
            M = B * tok
            x = torch.randn(M, K, device=dev, dtype=torch.half)
            n_copies = max(2, min(64, -(-320 * 2 ** 20 // (N * K * 2))))
            ws = [torch.randn(N, K, device=dev, dtype=torch.half) * K ** -0.5 for _ in range(n_copies)]
            bias = torch.randn(N, device=dev, dtype=torch.half)
            turn = [0]

            def nxt():
                turn[0] += 1
                return ws[turn[0] % n_copies]

            flop = 2.0 * M * K * N
            t_lib = replay_us(lambda: F.linear(x, nxt(), bias))
            steps = K // 64
            res = {}
            for bm, bn in ((128, 128), (128, 64), (64, 128), (64, 64)):
                for sp in (1, 2, 3, 4, 6, 8, 12, 16):
                    if sp > 1 and steps // sp < 2:
                        continue
                    tiles = -(-M // bm) * -(-N // bn)
                    if (tiles * sp > 2048 and sp > 1) or (tiles * sp < 96 and sp < 16 and steps // (sp + 1) >= 2):
                        continue
                    if sp > 1 and sp * tiles * bm * bn > ops.LIN_SLAB_FLOATS:
                        continue
                    plan = (bm, bn, sp)
                    res[plan] = replay_us(lambda: ops.linear_fused(x, nxt(), bias, plan=plan), iters=20, reps=2)
            best = min(res, key=res.get)
            y = ops.linear_fused(x, ws[0], bias, plan=best)["y"]
            ref = F.linear(x, ws[0], bias)
            err = float((y.float() - ref.float()).abs().max() / ref.float().abs().max())
            wins += res[best] <= t_lib
            print(f"{M:>6} {K:>5} {N:>6} {t_lib:8.1f} {flop / t_lib / 1e6:6.0f} | {str(best):>14} {res[best]:8.1f} "
                  f"{flop / res[best] / 1e6:6.0f} {t_lib / res[best]:5.2f}  err {err:.1e}", flush=True)
            table[f"{M},{K},{N}"] = {"lib_us": round(t_lib, 1), "best": list(best), "best_us": round(res[best], 1), "err": err,
                                     "all": {str(k): round(v, 1) for k, v in sorted(res.items(), key=lambda kv: kv[1])[:4]}}
            plans[f"{M},{K},{N},0"] = list(best)
    print(f"own kernel at least as fast as the library on {wins} of {len(table)} shapes (cold weights)")
    print(json.dumps(table))
    if "--write" in sys.argv:
        path = Path(__file__).resolve().parent.parent / "guided-attention_amd" / "linear_plans.json"
        old = json.loads(path.read_text()) if path.exists() else {}
        old.update(plans)
        path.write_text(json.dumps(old, indent=0, sort_keys=True))


if __name__ == "__main__":
    main()
