#!/bin/bash
# round 4, probe 14: the shipped conv loop with its issue order pinned (conv_order_fence) — tests, same-box A/B against the library
# before (tools/micro/libga_prev.so)
out=gpurun_out/r4r
mkdir -p $out
fault() { grep -q "Memory access fault" "$1" && { echo "GPU FAULT in $1"; exit 9; }; }
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "conv or upsample or consuming" > $out/conv_tests.log 2>&1; rc=$?
tail -2 $out/conv_tests.log; fault $out/conv_tests.log; [ $rc -eq 0 ] || { grep -n "^E " $out/conv_tests.log | head; exit $rc; }
for arm in new prev new prev; do
  echo "== unet_bench $arm" | tee -a $out/ab_fence.txt
  if [ $arm = prev ]; then export GA_HIP_LIB=$PWD/tools/micro/libga_prev.so; else unset GA_HIP_LIB; fi
  timeout -k 5 300 python3 tools/unet_bench.py 2>/dev/null | grep "ms" | tee -a $out/ab_fence.txt
done
unset GA_HIP_LIB
