#!/bin/bash
# round 4, probe 13: conv3x3_patch_dma_kernel with the step barrier in front of the last sub-step and fragments requested across
# the step boundary — tests first (short timeouts), stamps, same-box A/B against the library before the change (tools/micro/libga_prev.so)
out=gpurun_out/r4q
mkdir -p $out
fault() { grep -q "Memory access fault" "$1" && { echo "GPU FAULT in $1"; exit 9; }; }
timeout -k 10 120 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "conv3x3_implicit_gemm and 1x64x64x16x16x1" > $out/conv_first.log 2>&1; rc=$?
tail -2 $out/conv_first.log; fault $out/conv_first.log; [ $rc -eq 0 ] || { grep -n "^E " $out/conv_first.log | head; exit $rc; }
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "conv or upsample or consuming" > $out/conv_tests.log 2>&1; rc=$?
tail -2 $out/conv_tests.log; fault $out/conv_tests.log; [ $rc -eq 0 ] || { grep -n "^E " $out/conv_tests.log | head; exit $rc; }
for s in "1 1280 16 16 1280" "3 1280 16 16 1280" "1 1280 8 8 1280" "3 640 32 32 640" "1 320 64 64 320" "3 320 64 64 320"; do
  timeout -k 5 120 python3 tools/micro/conv_stamps.py $s 2>&1 | grep -v amdgpu.ids || exit 1
done > $out/conv_stamps_spec.txt
grep "shape\|per k-step\|replay" $out/conv_stamps_spec.txt
for arm in new prev new prev; do
  echo "== unet_bench $arm" | tee -a $out/ab_spec.txt
  if [ $arm = prev ]; then export GA_HIP_LIB=$PWD/tools/micro/libga_prev.so; else unset GA_HIP_LIB; fi
  timeout -k 5 300 python3 tools/unet_bench.py 2>/dev/null | grep "ms" | tee -a $out/ab_spec.txt
done
unset GA_HIP_LIB
