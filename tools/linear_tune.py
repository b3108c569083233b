#!/usr/bin/env python3
"""Per-shape time of the UNet's Linear layers / 1x1 convolutions at COLD weights: the library form the round-2 build ran
(torch.nn.functional.linear on hipBLASLt plus the separate LayerNorm / GEGLU / residual-add launches around it) against
ga_linear_fused for every (tile, split-K, ring depth) plan.  In the pipeline a layer's weights are cold (1.7 GB of UNet
weights cycle through the 256 MB Infinity Cache between two uses) while its input was just written: both contenders rotate
over enough weight copies to exceed the cache.  hipGraph replay timing between two HIP events.

  python tools/linear_tune.py [batches, default 1,3] [--write] [--mode plain|fused|all]
     plain : Y = X W^T + b                         vs F.linear                       (the 38 shapes per batch of round 2's table)
     fused : the transformer block's real calls    vs the library + ga element-wise kernels:
             ln    LayerNorm -> Linear                          (qkv, to_q)
             res   Linear + bias + residual                     (to_out, FF out, proj_out)
             geglu LayerNorm -> Linear -> GEGLU                 (FF in)
  --write: guided-attention_amd/linear_plans.json  {"M,K,N,geglu": [bm, bn, splits, stages]}
"""
import json
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from guided_attention_amd import ops  # noqa: E402
from gemm_tune import BASE  # noqa: E402

TILES = ((128, 128, 3), (128, 128, 2), (128, 64, 4), (128, 64, 3), (64, 128, 4), (64, 128, 3), (64, 64, 4), (64, 64, 5))


def replay_us(fn, iters=20, reps=3):
    side = ops.side_stream()
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


def candidates(M, K, n_out, geglu):
    steps = K // 64
    for bm, bn, st in TILES:
        outc = bn // 2 if geglu else bn
        tiles = -(-M // bm) * -(-n_out // outc)
        for sp in (1, 2, 3, 4, 6, 8, 12, 16):
            if sp > 1 and steps // sp < 2:
                continue
            if (tiles * sp > 2048 and sp > 1) or (tiles * sp < 96 and sp < 16 and steps // (sp + 1) >= 2):
                continue
            if sp > 1 and sp * tiles * bm * bn > ops.LIN_SLAB_FLOATS:
                continue
            yield (bm, bn, sp, st)


def cold(N, K, dev):
    n_copies = max(2, min(64, -(-320 * 2 ** 20 // (N * K * 2))))
    ws = [torch.randn(N, K, device=dev, dtype=torch.half) * K ** -0.5 for _ in range(n_copies)]
    turn = [0]

    def nxt():
        turn[0] += 1
        return ws[turn[0] % n_copies]
    return ws, nxt


LEVELS = ((4096, 320), (1024, 640), (256, 1280), (64, 1280))          # SD-1.x at 512^2: (tokens, channels) per level
if "sd21" in sys.argv:                                                   # BASELINE config 4: 96 x 96 latents
    LEVELS = ((9216, 320), (2304, 640), (576, 1280), (144, 1280))
if "sdxl" in sys.argv:                                                   # BASELINE config 5: attention at 64^2 / 32^2 only
    LEVELS = ((4096, 640), (1024, 1280))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--") and a not in ("sd21", "sdxl")]
    batches = [int(b) for b in (args[0] if args else "1,3").split(",")]
    mode = sys.argv[sys.argv.index("--mode") + 1] if "--mode" in sys.argv else "all"
    if "--mode" in sys.argv:
        args = [a for a in args if a != mode]
        batches = [int(b) for b in (args[0] if args else "1,3").split(",")]
    dev = torch.device("cuda")
    ops.prepare_device(dev)
    table, plans, wins, total = {}, {}, 0, 0

    def report(tag, M, K, N, t_lib, res, flop, key):
        nonlocal wins, total
        best = min(res, key=res.get)
        wins += res[best] <= t_lib
        total += 1
        print(f"{tag:>6} {M:>6} {K:>5} {N:>6} {t_lib:8.1f} {flop / t_lib / 1e6:6.0f} | {str(best):>18} {res[best]:8.1f} "
              f"{flop / res[best] / 1e6:6.0f} {t_lib / res[best]:5.2f}", flush=True)
        table[f"{tag},{M},{K},{N}"] = {"lib_us": round(t_lib, 1), "best": list(best), "best_us": round(res[best], 1),
                                       "top": {str(k): round(v, 1) for k, v in sorted(res.items(), key=lambda kv: kv[1])[:4]}}
        if key not in plans or tag != "plain":     # the fused forms are what the pipeline runs: they win the table entry
            plans[key] = list(best)

    def stream(res, tag, M, K, N, g, call):
        """The persistent form (linear_stream_kernel) beside the best per-tile plan: measured, printed, and entered in the table
        as "M,K,N,geglu,8": [1 | 0] (ops.linear_plan takes it for the calls that form serves when it is at least 3 % faster).
        `partials` of this harness has ONE part per row; the stream form wants >= 2: it gets a two-part copy."""
        if K // 64 < 5:
            return
        x2 = x_two_parts.get(M)
        if x2 is None:
            return
        t = replay_us(lambda: call2(M, K, N, g), reps=3)
        best = min(res.values())
        plans[f"{M},{K},{N},{g},{ops.LINEAR_STREAM}"] = [1 if t < 0.97 * best else 0]
        print(f"{tag:>6} {M:>6} {K:>5} {N:>6} stream form {t:8.1f} us  {2.0 * M * K * N / t / 1e6:6.0f} TF/s   best per-tile plan {best:8.1f} us"
              f"  -> {'stream' if t < 0.97 * best else 'per-tile'}", flush=True)

    x_two_parts, stream_ctx = {}, {}

    def call2(M, K, N, g):
        xx, nx, cs_, sh_ = stream_ctx[(M, K, N, g)]
        return ops.linear_fused(xx, nx(), None, geglu=bool(g), ln=(x_two_parts[M], cs_, sh_, 1e-5), plan=ops.LINEAR_STREAM_PLAN)

    print(f"{'form':>6} {'M':>6} {'K':>5} {'N':>6} {'lib us':>8} {'TF/s':>6} | {'best plan':>18} {'us':>8} {'TF/s':>6} {'x':>5}")
    for B in batches:
        if mode in ("plain", "all"):
            for tok, K, N in BASE:
                M = B * tok
                x = torch.randn(M, K, device=dev, dtype=torch.half)
                bias = torch.randn(N, device=dev, dtype=torch.half)
                _, nxt = cold(N, K, dev)
                t_lib = replay_us(lambda: F.linear(x, nxt(), bias))
                res = {p: replay_us(lambda: ops.linear_fused(x, nxt(), bias, plan=p), reps=2) for p in candidates(M, K, N, False)}
                report("plain", M, K, N, t_lib, res, 2.0 * M * K * N, f"{M},{K},{N},0")
        if mode in ("fused", "all"):
            for tok, C in LEVELS:
                M = B * tok
                x = torch.randn(M, C, device=dev, dtype=torch.half)
                g, b_ = torch.ones(C, device=dev, dtype=torch.half), torch.zeros(C, device=dev, dtype=torch.half)
                partials = torch.stack([x.float().sum(-1), (x.float() ** 2).sum(-1)], -1)[:, None, :].contiguous()
                half = C // 2
                x_two_parts[M] = torch.stack([torch.stack([x[:, :half].float().sum(-1), (x[:, :half].float() ** 2).sum(-1)], -1),
                                              torch.stack([x[:, half:].float().sum(-1), (x[:, half:].float() ** 2).sum(-1)], -1)], 1).contiguous()
                for N in (3 * C, C):            # LayerNorm -> qkv / to_q
                    bias = torch.randn(N, device=dev, dtype=torch.half)
                    _, nxt = cold(N, C, dev)
                    cs, sh = torch.randn(N, device=dev), torch.randn(N, device=dev)
                    stream_ctx[(M, C, N, 0)] = (x, nxt, cs, sh)
                    t_lib = replay_us(lambda: F.linear(ops.layer_norm(x, g, b_, 1e-5), nxt(), bias))
                    res = {p: replay_us(lambda: ops.linear_fused(x, nxt(), None, ln=(partials, cs, sh, 1e-5), plan=p), reps=2)
                           for p in candidates(M, C, N, False)}
                    stream(res, "ln", M, C, N, 0, lambda p: ops.linear_fused(x, nxt(), None, ln=(partials, cs, sh, 1e-5), plan=p))
                    report("ln", M, C, N, t_lib, res, 2.0 * M * C * N, f"{M},{C},{N},0")
                for K in (C, 4 * C):            # Linear + bias + residual: to_out / proj_out, FF out
                    xk = torch.randn(M, K, device=dev, dtype=torch.half)
                    bias = torch.randn(C, device=dev, dtype=torch.half)
                    r = torch.randn(M, C, device=dev, dtype=torch.half)
                    _, nxt = cold(C, K, dev)
                    t_lib = replay_us(lambda: F.linear(xk, nxt(), bias) + r)
                    res = {p: replay_us(lambda: ops.linear_fused(xk, nxt(), bias, residual=r, want_row_partials=True, plan=p), reps=2)
                           for p in candidates(M, K, C, False)}
                    report("res", M, K, C, t_lib, res, 2.0 * M * K * C, f"{M},{K},{C},0")
                N = 8 * C                       # LayerNorm -> FF in -> GEGLU
                bias = torch.randn(N, device=dev, dtype=torch.half)
                _, nxt = cold(N, C, dev)
                cs, sh = torch.randn(N, device=dev), torch.randn(N, device=dev)
                stream_ctx[(M, C, N, 1)] = (x, nxt, cs, sh)
                t_lib = replay_us(lambda: ops.geglu(F.linear(ops.layer_norm(x, g, b_, 1e-5), nxt(), bias)))
                res = {p: replay_us(lambda: ops.linear_fused(x, nxt(), None, geglu=True, ln=(partials, cs, sh, 1e-5), plan=p), reps=2)
                       for p in candidates(M, C, N // 2, True)}
                stream(res, "geglu", M, C, N, 1, lambda p: ops.linear_fused(x, nxt(), None, geglu=True, ln=(partials, cs, sh, 1e-5), plan=p))
                report("geglu", M, C, N, t_lib, res, 2.0 * M * C * N, f"{M},{C},{N},1")
    # whole transformer block per (tokens, channels): LN->qkv, to_out, LN->to_q, to_out, LN->GEGLU, FF out.  Where the library
    # form of the block is more than 3 % faster the block keeps it (key "M,C,-1,-1", read by ops.library_block)
    if mode in ("fused", "all"):
        for B in batches:
            for tok, C in LEVELS:
                M = B * tok
                rows = [table.get(f"ln,{M},{C},{3 * C}"), table.get(f"res,{M},{C},{C}"), table.get(f"ln,{M},{C},{C}"),
                        table.get(f"res,{M},{C},{C}"), table.get(f"geglu,{M},{C},{8 * C}"), table.get(f"res,{M},{4 * C},{C}")]
                if all(rows):
                    own, lib_ = sum(r["best_us"] for r in rows), sum(r["lib_us"] for r in rows)
                    print(f"block {M:>6} x {C:>5}: folded {own:7.1f} us, library form {lib_:7.1f} us -> {'library' if lib_ < 0.97 * own else 'folded'}")
                    plans[f"{M},{C},-1,-1"] = [1 if lib_ < 0.97 * own else 0]
    print(f"own kernel at least as fast as the library form on {wins} of {total} cases (cold weights)")
    print(json.dumps(table))
    if "--write" in sys.argv:
        path = Path(__file__).resolve().parent.parent / "guided-attention_amd" / "linear_plans.json"
        old = json.loads(path.read_text()) if path.exists() else {}
        old.update(plans)
        path.write_text(json.dumps(old, indent=0, sort_keys=True))


if __name__ == "__main__":
    main()
