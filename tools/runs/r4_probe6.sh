#!/bin/bash
# round 4, probe 6: self-attention with adjacent heads per XCD (tests, PMC traffic), same-box A/B of the GroupNorm producer and
# of the stream form, PMC traffic of the benched model's dominant shapes
out=gpurun_out/r4i
mkdir -p $out
fault() { grep -q "Memory access fault" "$1" && { echo "GPU FAULT in $1"; exit 9; }; }
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "self_attention or flash_self" > $out/sa_tests.log 2>&1; rc=$?
tail -2 $out/sa_tests.log; fault $out/sa_tests.log; [ $rc -eq 0 ] || { grep -n "^E " $out/sa_tests.log | head; exit $rc; }
for arm in "" "no-gn-producer" "no-stream" "" "no-gn-producer" "no-stream"; do
  echo "== unet_bench ${arm:-default}" | tee -a $out/ab_passes.txt
  timeout -k 5 300 python3 tools/unet_bench.py $arm 2>/dev/null | grep "ms" | tee -a $out/ab_passes.txt
done
timeout -k 10 900 python3 tools/pmc_traffic.py sd15 > $out/pmc_traffic_sd15.log 2>&1; rc=$?
tail -20 $out/pmc_traffic_sd15.log; [ $rc -eq 0 ] && cp gpurun_out/r4_pmc_traffic_sd15.json $out/
