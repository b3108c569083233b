#!/bin/bash
# round 4, probe 5: GroupNorm statistics from the producing convolution's epilogue; stream form (immediate epilogue, statistics by all waves)
out=gpurun_out/r4h
mkdir -p $out
fault() { grep -q "Memory access fault" "$1" && { echo "GPU FAULT in $1"; exit 9; }; }
timeout -k 10 500 python -m pytest tests/test_kernels_gpu.py -m gpu -q -s -x -k "stream_form or benched_feed or consuming_group_norm or conv3x3_implicit or group_norm" > $out/kernel_tests.log 2>&1; rc=$?
tail -3 $out/kernel_tests.log; fault $out/kernel_tests.log; [ $rc -eq 0 ] || { grep -n "^E " $out/kernel_tests.log | head -20; exit $rc; }
timeout -k 5 300 python3 tools/unet_bench.py > $out/unet_bench_before_tune.txt 2>&1; grep "ms" $out/unet_bench_before_tune.txt
timeout -k 10 500 python3 tools/linear_tune.py 2,3 --mode fused --write > $out/linear_tune_fused.txt 2>&1; rc=$?
grep "stream form" $out/linear_tune_fused.txt | grep -v " 128 \| 192 \| 512 "; fault $out/linear_tune_fused.txt; [ $rc -eq 0 ] || exit $rc
cp guided-attention_amd/linear_plans.json $out/linear_plans.json
timeout -k 5 300 python3 tools/unet_bench.py > $out/unet_bench.txt 2>&1; grep "ms" $out/unet_bench.txt
timeout -k 10 600 python -m pytest tests/test_pipeline_gpu.py tests/test_unet_forward_golden.py -m gpu -q -s -x --durations=8 > $out/pipeline_tests.log 2>&1; rc=$?
tail -12 $out/pipeline_tests.log; fault $out/pipeline_tests.log; [ $rc -eq 0 ] || { grep -n "^E " $out/pipeline_tests.log | head -20; exit $rc; }
