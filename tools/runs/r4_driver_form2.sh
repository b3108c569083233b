#!/bin/bash
# the driver's own command line next to the default-argument line, same box, final tree
out=gpurun_out/r4drv2
mkdir -p $out
timeout -k 5 400 python3 bench.py --no-cpu-baseline --no-roofline > $out/bench_default.json 2> $out/bench_default.err || exit 1
timeout -k 5 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_steps20_warmup5.json 2> $out/bench_steps20.err || exit 1
python3 - <<'PY'
import json
for n in ("bench_default", "bench_steps20_warmup5"):
    d = json.load(open(f"gpurun_out/r4drv2/{n}.json"))
    print(n, round(d["value"], 4), "images/s", round(d["ms_per_step"], 1), "ms", d["roofline"]["frac"] if d.get("roofline") else "")
PY
