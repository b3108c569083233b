#!/bin/bash
# round 4, probe 3: stream-form Linear (DMA refill interleaved with the MFMAs) — tests, tune at B = 2, 3, bench A/B of the MIOpen search scope
out=gpurun_out/r4e
mkdir -p $out
fault() { grep -q "Memory access fault" "$1" && { echo "GPU FAULT in $1"; exit 9; }; }
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -s -x -k "stream_form" > $out/stream_tests.log 2>&1; rc=$?
tail -3 $out/stream_tests.log; fault $out/stream_tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python3 tools/linear_tune.py 2,3 --mode fused --write > $out/linear_tune_fused.txt 2>&1; rc=$?
grep "stream form\|own kernel" $out/linear_tune_fused.txt; fault $out/linear_tune_fused.txt; [ $rc -eq 0 ] || exit $rc
cp guided-attention_amd/linear_plans.json $out/linear_plans.json
for mode in "" "--miopen-search" ""; do
  timeout -k 5 400 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --two-pass-steps 0 $mode > $out/bench_tmp.json 2> $out/bench_err.txt || { tail $out/bench_err.txt; exit 1; }
  fault $out/bench_err.txt
  python3 -c "
import json
l=json.loads(open('$out/bench_tmp.json').read().strip().splitlines()[-1])
print('bench', '$mode' or 'default (search scoped to conv_in / conv_out forward)', round(l['value'],4), 'images/s', round(l['ms_per_step'],1), 'ms')
" | tee -a $out/bench_ab.txt
done
timeout -k 5 300 python3 tools/unet_bench.py > $out/unet_bench.txt 2>&1; cat $out/unet_bench.txt
