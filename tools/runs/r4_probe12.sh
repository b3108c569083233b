#!/bin/bash
# round 4, probe 12: what bounds a k-step of conv3x3_patch_dma_kernel — ablation builds of the stamped kernel
# (spec = producer / consumer waves; abl bits: 1 no MFMA body, 2 no patch traffic, 4 no weight DMAs)
out=gpurun_out/r4p
mkdir -p $out
for lib in spec0_abl0 spec0_abl1 spec1_abl1 spec1_abl3 spec1_abl5 spec1_abl6; do
  export GA_STAMPS_LIB=$PWD/tools/micro/libga_conv_stamps_$lib.so
  echo "=== $lib"
  for s in "1 1280 16 16 1280" "1 320 64 64 320" "3 320 64 64 320"; do
    timeout -k 5 120 python3 tools/micro/conv_stamps.py $s 2>&1 | grep "shape\|per k-step\|replay" || exit 1
  done
done > $out/conv_ablation.txt
cat $out/conv_ablation.txt
