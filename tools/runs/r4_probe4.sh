#!/bin/bash
# round 4, probe 4: stream form with the deferred, staggered GEGLU epilogue — tests, stamps, tune (B = 2, 3), passes
out=gpurun_out/r4g
mkdir -p $out
fault() { grep -q "Memory access fault" "$1" && { echo "GPU FAULT in $1"; exit 9; }; }
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -s -x -k "stream_form or benched_feed" > $out/stream_tests.log 2>&1; rc=$?
tail -3 $out/stream_tests.log; fault $out/stream_tests.log; [ $rc -eq 0 ] || exit $rc
for s in "12288 320 2560 3" "3072 640 5120 3" "768 1280 10240 3" "12288 320 960 2"; do timeout -k 5 120 python3 tools/micro/lin_stamps.py $s 128 128 1 8 || exit 1; done > $out/stream_stamps.txt 2>&1
grep -v amdgpu.ids $out/stream_stamps.txt
timeout -k 10 500 python3 tools/linear_tune.py 2,3 --mode fused --write > $out/linear_tune_fused.txt 2>&1; rc=$?
grep "stream form\|own kernel" $out/linear_tune_fused.txt; fault $out/linear_tune_fused.txt; [ $rc -eq 0 ] || exit $rc
cp guided-attention_amd/linear_plans.json $out/linear_plans.json
timeout -k 5 300 python3 tools/unet_bench.py > $out/unet_bench.txt 2>&1; grep -v "amdgpu.ids\|Warning\|benchmark_limit" $out/unet_bench.txt
