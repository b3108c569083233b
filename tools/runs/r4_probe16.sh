#!/bin/bash
# round 4, probe 16: concatenation + norm as one launch on the 16 x 16 / 8 x 8 levels (ga_cat_group_norm_fwd) — tests, same-box arms
out=gpurun_out/r4t
mkdir -p $out
fault() { grep -q "Memory access fault" "$1" && { echo "GPU FAULT in $1"; exit 9; }; }
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "cat_channels or group_norm or consuming" > $out/gn_tests.log 2>&1; rc=$?
tail -2 $out/gn_tests.log; fault $out/gn_tests.log; [ $rc -eq 0 ] || { grep -n "^E " $out/gn_tests.log | head; exit $rc; }
for arm in "" no-cat-norm "" no-cat-norm; do
  echo "== unet_bench ${arm:-default}" | tee -a $out/ab_cat_norm.txt
  timeout -k 5 300 python3 tools/unet_bench.py $arm 2>/dev/null | grep "ms" | tee -a $out/ab_cat_norm.txt
done
timeout -k 10 600 python -m pytest tests/test_pipeline_gpu.py tests/test_unet_forward_golden.py -m gpu -q -x -k "full_width or half_precision or golden or forward" > $out/pipe_tests.log 2>&1; rc=$?
tail -2 $out/pipe_tests.log; fault $out/pipe_tests.log; [ $rc -eq 0 ] || { grep -n "^E " $out/pipe_tests.log | head; exit $rc; }
