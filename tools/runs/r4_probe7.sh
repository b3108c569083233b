#!/bin/bash
# round 4, probe 7: DPP row / wave sums instead of ds_bpermute shuffles — tests, same-box A/B against the library before the change
out=gpurun_out/r4k
mkdir -p $out
fault() { grep -q "Memory access fault" "$1" && { echo "GPU FAULT in $1"; exit 9; }; }
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "linear or group_norm or layer_norm or geglu or consuming" > $out/kernel_tests.log 2>&1; rc=$?
tail -2 $out/kernel_tests.log; fault $out/kernel_tests.log; [ $rc -eq 0 ] || { grep -n "^E " $out/kernel_tests.log | head; exit $rc; }
for arm in new prev new prev; do
  echo "== unet_bench $arm" | tee -a $out/ab_dpp.txt
  if [ $arm = prev ]; then export GA_HIP_LIB=$PWD/tools/micro/libga_prev.so; else unset GA_HIP_LIB; fi
  timeout -k 5 300 python3 tools/unet_bench.py 2>/dev/null | grep "ms" | tee -a $out/ab_dpp.txt
done
unset GA_HIP_LIB
for s in "4096 320 320 4" "12288 320 320 4" "768 1280 1280 4"; do timeout -k 5 120 python3 tools/micro/lin_stamps.py $s || exit 1; done > $out/lin_stamps_dpp.txt 2>&1
grep -v amdgpu.ids $out/lin_stamps_dpp.txt | grep "shape\|epilogue\|lifetime\|replay"
