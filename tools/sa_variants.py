#!/usr/bin/env python3
"""A/B harness for the self-attention kernels: several builds of csrc/self_attn.hip (ablation switches GA_ABL, older
revisions from git) side by side in ONE process, interleaved rounds, hipGraph replays between HIP events.

  python tools/sa_variants.py build            (here: cross-compiles every variant into tools/micro/sa_variants/)
  python tools/sa_variants.py run [fwd|bwd]    (on the GPU box: times them; prints a table and a JSON line)

Variants with GA_ABL != 0 compute wrong results on purpose (a piece of the loop is removed): only their time is read.
"""
import ctypes
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
OUT = ROOT / "tools" / "micro" / "sa_variants"
SRC = ROOT / "guided-attention_amd" / "csrc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-std=c++17", f"-I{ROOT / 'include'}", f"-I{SRC}",
         "-mllvm", "-amdgpu-mfma-vgpr-form", "-Wno-unused-function"]

# name -> (git revision of self_attn.hip or None for the working tree, extra -D flags)
VARIANTS = {
    "cur": (None, []),
    "prio": (None, ["-DGA_SA_PRIO=1"]),              # s_setprio 1 around the MFMA batches
    "noslp": (None, ["-fno-slp-vectorize"]),         # scalar f32 softmax arithmetic instead of the SLP-packed v_pk_* forms
    "prio_noslp": (None, ["-DGA_SA_PRIO=1", "-fno-slp-vectorize"]),
    "nw2": (None, ["-DGA_FWD_NW=2"]),
}


def build(nk=3):
    OUT.mkdir(parents=True, exist_ok=True)
    for name, (rev, defs) in VARIANTS.items():
        src = SRC / "self_attn.hip"
        micro = [f"-DGA_SA_MICRO={nk}"]
        if rev is not None:
            tmp = OUT / f"self_attn_{rev}.hip"
            tmp.write_text(subprocess.run(["git", "-C", str(ROOT), "show", f"{rev}:guided-attention_amd/csrc/self_attn.hip"],
                                          capture_output=True, text=True, check=True).stdout)
            src, micro = tmp, []      # old revisions have no micro mode: full build
        cmd = ["/opt/rocm/bin/hipcc", *FLAGS, *micro, *defs, str(src), "-o", str(OUT / f"libsa_{name}.so")]
        subprocess.run(cmd, check=True)
        print("built", name)


def run(which="fwd", rounds=5, iters=50):
    import torch
    vp, i32, f32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    libs = {}
    for name in list(VARIANTS) + ["prod"]:
        path = OUT / f"libsa_{name}.so" if name != "prod" else ROOT / "guided-attention_amd" / "libga_hip.so"
        if not path.exists():
            continue
        lib = ctypes.CDLL(str(path))
        lib.ga_self_attn_fwd.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, i32, vp]
        lib.ga_self_attn_bwd.argtypes = [vp] * 10 + [i32, i32, i32, i32, i32, f32, i32, vp]
        libs[name] = lib
    import os
    shapes = [tuple(int(v) for v in t.split('x')) for t in os.environ.get('SA_SHAPES', '1x8x4096x40,3x8x4096x40').split(',')]
    results = {}
    for (B, H, N, D) in shapes:
        dev = torch.device("cuda")
        q, k, v, do = (torch.randn(B, N, H * D, device=dev, dtype=torch.half) for _ in range(4))
        o = torch.empty_like(q)
        lse = torch.empty(B * H, N, device=dev, dtype=torch.float32)
        delta = torch.empty_like(lse)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
        P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        graphs = {}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            sp = ctypes.c_void_p(side.cuda_stream)
            for name, lib in libs.items():
                def call(lib=lib):
                    if which == "fwd":
                        rc = lib.ga_self_attn_fwd(P(q), P(k), P(v), P(o), P(lse), B, H, N, D, 0, D ** -0.5, 0, sp)
                    else:
                        rc = lib.ga_self_attn_bwd(P(q), P(k), P(v), P(o), P(do), P(lse), P(delta), P(dq), P(dk), P(dv),
                                                  B, H, N, D, 0, D ** -0.5, 0, sp)
                    assert rc == 0, (name, rc)
                if which == "bwd":   # a consistent (O, LSE) pair from the product-candidate build
                    libs["cur"].ga_self_attn_fwd(P(q), P(k), P(v), P(o), P(lse), B, H, N, D, 0, D ** -0.5, 0, sp)
                for _ in range(3):
                    call()
                side.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    for _ in range(iters):
                        call()
                graphs[name] = g
            times = {n: [] for n in graphs}
            for _ in range(rounds):          # interleaved rounds in one process (methodology rule 24)
                for name, g in graphs.items():
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    g.replay()
                    side.synchronize()
                    e0.record(side)
                    g.replay()
                    e1.record(side)
                    side.synchronize()
                    times[name].append(e0.elapsed_time(e1) * 1e3 / iters)
        torch.cuda.current_stream().wait_stream(side)
        flops = 4.0 * B * H * N * N * D * (1.0 if which == "fwd" else 2.5)
        print(f"--- {which} B={B} H={H} N={N} D={D}")
        for name, ts in times.items():
            med, best = sorted(ts)[len(ts) // 2], min(ts)
            print(f"{name:28s} median {med:8.2f} us   min {best:8.2f} us   {flops / med / 1e6:7.1f} TFLOP/s")
            results[f"{which}.B{B}.{name}"] = {"median_us": round(med, 2), "min_us": round(best, 2)}
    print(json.dumps(results))


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
    else:
        run(*(sys.argv[2:3] or ["fwd"]))
