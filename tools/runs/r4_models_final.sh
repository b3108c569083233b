#!/bin/bash
# the other two configurations and the every-step workload on the final tree
out=gpurun_out/r4models
mkdir -p $out
timeout -k 5 400 python3 bench.py --model sd21 --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_sd21_768.json 2> $out/bench_sd21.err || exit 1
timeout -k 5 400 python3 bench.py --model sdxl --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_sdxl.json 2> $out/bench_sdxl.err || exit 1
timeout -k 5 400 python3 bench.py --workload every-step --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_every_step.json 2> $out/bench_every_step.err || exit 1
timeout -k 5 400 python3 bench.py --guidance-forward truncated --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_truncated_guidance_forward.json 2> $out/bench_truncated.err || exit 1
python3 - <<'PY'
import json
for n in ("bench_sd21_768", "bench_sdxl", "bench_every_step", "bench_truncated_guidance_forward"):
    d = json.load(open(f"gpurun_out/r4models/{n}.json"))
    print(n, round(d["value"], 4), "images/s", round(d["ms_per_step"], 1), "ms")
PY
