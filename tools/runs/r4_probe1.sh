#!/bin/bash
# round 4, probe 1: where linear_kernel's time goes (stamps), host gaps A/B (run-ahead refinement), per-pass times
out=gpurun_out/r4b
mkdir -p $out
{
for s in "4096 320 320 4" "1024 640 640 4" "256 1280 1280 4" "768 1280 1280 4" "12288 320 2560 3" "3072 640 5120 3" "768 1280 10240 3" "768 5120 1280 4" "4096 320 2560 3" "4096 320 960 2" "12288 320 320 4"; do
  timeout -k 5 120 python3 tools/micro/lin_stamps.py $s || exit 1
done
} > $out/lin_stamps.txt 2>&1 || { tail -20 $out/lin_stamps.txt; exit 1; }
tail -40 $out/lin_stamps.txt
timeout -k 5 300 python3 tools/unet_bench.py > $out/unet_bench.txt 2>&1 || { tail $out/unet_bench.txt; exit 1; }
cat $out/unet_bench.txt
for mode in "" "--no-run-ahead" "" "--no-run-ahead"; do
  timeout -k 5 400 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --two-pass-steps 0 $mode > $out/bench_tmp.json 2> $out/bench_err.txt || { tail $out/bench_err.txt; exit 1; }
  python3 -c "
import json,sys
l=json.loads(open('$out/bench_tmp.json').read().strip().splitlines()[-1])
print('bench', '$mode' or 'run-ahead', round(l['value'],4), 'images/s', round(l['ms_per_step'],1), 'ms', l['unet_calls_per_image'])
" | tee -a $out/bench_ab.txt
done
