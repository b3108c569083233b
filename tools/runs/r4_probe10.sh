#!/bin/bash
# round 4, probe 10: clock stamps inside conv3x3_patch_dma_kernel on the weight-heavy and the activation-heavy shapes
out=gpurun_out/r4n
mkdir -p $out
for s in "1 1280 16 16 1280" "3 1280 16 16 1280" "1 1280 8 8 1280" "1 640 32 32 640" "3 640 32 32 640" "1 320 64 64 320" "3 320 64 64 320" "1 2560 16 16 1280"; do
  timeout -k 5 120 python3 tools/micro/conv_stamps.py $s 2>&1 | grep -v amdgpu.ids || exit 1
done > $out/conv_stamps.txt
cat $out/conv_stamps.txt
