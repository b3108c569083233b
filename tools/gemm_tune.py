#!/usr/bin/env python3
"""Per-shape time of the UNet's Linear layers / 1x1 convolutions: torch.nn.functional.linear (hipBLASLt) against
ga_gemm_nt for every (tile, split-K) plan.  Checks the result of the best plan against the library's.  hipGraph replay
timing.  Prints a table and a JSON line.  usage: gemm_tune.py [batches, default 1,3]"""
import ctypes
import json
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from guided_attention_amd import ops  # noqa: E402
from guided_attention_amd._lib import load, dtype_code, stream_ptr  # noqa: E402
from conv_tune import replay_us  # noqa: E402

# (tokens per sample, K, N): SD-1.x transformer blocks and ResnetBlock shortcuts, forward and (swapped) backward-to-input
BASE = []
for tok, c in ((4096, 320), (1024, 640), (256, 1280), (64, 1280)):
    BASE += [(tok, c, 3 * c), (tok, c, c), (tok, c, 8 * c), (tok, 4 * c, c), (tok, 3 * c, c), (tok, 8 * c, c), (tok, c, 4 * c)]
BASE += [(4096, 640, 320), (4096, 960, 320), (1024, 320, 640), (1024, 1280, 640), (1024, 1920, 640), (1024, 960, 640),
         (256, 640, 1280), (256, 2560, 1280), (256, 1920, 1280), (64, 2560, 1280)]


def main():
    batches = [int(b) for b in (sys.argv[1] if len(sys.argv) > 1 else "1,3").split(",")]
    lib = load()
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
    table = {}
    print(f"{'M':>6} {'K':>5} {'N':>6} {'lib us':>8} {'TF/s':>6} | {'best plan':>14} {'us':>8} {'TF/s':>6} {'x':>5}")
    for B in batches:
        for tok, K, N in BASE:
            M = B * tok
            x = torch.randn(M, K, device="cuda", dtype=torch.half)
            w = torch.randn(N, K, device="cuda", dtype=torch.half) * K ** -0.5
            bias = torch.randn(N, device="cuda", dtype=torch.half)
            y = torch.empty(M, N, device="cuda", dtype=torch.half)
            flop = 2.0 * M * K * N
            t_lib = replay_us(lambda: F.linear(x, w, bias))
            ref = F.linear(x, w, bias)
            steps = K // 64
            res = {}
            for bm, bn in ((128, 128), (128, 64), (64, 64)):
                for sp in (1, 2, 3, 4, 6, 8):
                    if sp > 1 and steps // sp < 2:
                        continue
                    tiles = -(-M // bm) * -(-N // bn)
                    if tiles * sp > 4096 or (tiles * sp < 64 and sp < 8):
                        continue
                    def call():
                        ws, tk = ops.splitk_workspace(x.device, M, N, bm, bn, sp)   # the scratch of the stream this runs on
                        rc = lib.ga_gemm_nt(P(x), P(w), P(y), P(ws), P(tk), P(bias), None, M, K, N, bm, bn, sp, dtype_code(x), stream_ptr())
                        assert rc == 0, rc
                    res[(bm, bn, sp)] = replay_us(call, iters=10)
            best = min(res, key=res.get)
            bm, bn, sp = best
            ws, tk = ops.splitk_workspace(x.device, M, N, bm, bn, sp)
            assert lib.ga_gemm_nt(P(x), P(w), P(y), P(ws), P(tk), P(bias), None, M, K, N, bm, bn, sp, dtype_code(x), stream_ptr()) == 0
            err = float((y.float() - ref.float()).abs().max() / ref.float().abs().max())
            print(f"{M:>6} {K:>5} {N:>6} {t_lib:8.1f} {flop / t_lib / 1e6:6.0f} | {str(best):>14} {res[best]:8.1f} "
                  f"{flop / res[best] / 1e6:6.0f} {t_lib / res[best]:5.2f}  err {err:.1e}", flush=True)
            table[f"{M},{K},{N}"] = {"lib_us": round(t_lib, 1), "best": list(best), "best_us": round(res[best], 1), "err": err}
    print(json.dumps(table))


if __name__ == "__main__":
    main()
