#!/bin/bash
# round 4, probe 17: the fused-Linear plan table re-measured on the final kernels (DPP row sums changed the epilogues) — passes
# before / after on the same box; the new table is kept only if it is faster
out=gpurun_out/r4u
mkdir -p $out
cp guided-attention_amd/linear_plans.json $out/linear_plans_before.json
for i in 1 2; do timeout -k 5 300 python3 tools/unet_bench.py 2>/dev/null | grep "ms"; done | tee $out/unet_bench_before_tune.txt
timeout -k 10 700 python3 tools/linear_tune.py 1,2,3 --mode fused --write > $out/linear_tune_fused.txt 2>&1 || { tail -5 $out/linear_tune_fused.txt; exit 1; }
grep -c "" $out/linear_tune_fused.txt
cp guided-attention_amd/linear_plans.json $out/linear_plans.json
for i in 1 2; do timeout -k 5 300 python3 tools/unet_bench.py 2>/dev/null | grep "ms"; done | tee $out/unet_bench_after_tune.txt
python3 - <<'PY'
import json
a = json.load(open("gpurun_out/r4u/linear_plans_before.json")); b = json.load(open("gpurun_out/r4u/linear_plans.json"))
ch = {k: (a.get(k), b[k]) for k in b if a.get(k) != b[k]}
print(len(ch), "of", len(b), "entries changed")
for k, v in list(ch.items())[:40]: print(" ", k, v[0], "->", v[1])
PY
