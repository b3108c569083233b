#!/bin/bash
# round 4, probe 18: thin_out with the next segment's patch in flight, 256 workgroups — test, launch times
out=gpurun_out/r4v
mkdir -p $out
fault() { grep -q "Memory access fault" "$1" && { echo "GPU FAULT in $1"; exit 9; }; }
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "thin" > $out/thin_tests.log 2>&1; rc=$?
tail -2 $out/thin_tests.log; fault $out/thin_tests.log; [ $rc -eq 0 ] || { grep -n "^E " $out/thin_tests.log | head -20; exit $rc; }
timeout -k 5 120 python3 - > $out/thin_launch_us.txt 2>&1 <<'PY' || { tail -5 $out/thin_launch_us.txt; exit 1; }
import sys; sys.path.insert(0, ".")
from guided_attention_amd import ops
for B in (1, 2, 3):
    for kind, cin, cout in (("conv3x3_thin_in", 4, 320), ("conv3x3_thin_out", 320, 4)):
        for hw in (4096, 9216, 16384):
            us = ops.replay_launch_us((kind, B, cin, hw, 1, cout, True, "torch.float16"))
            gb = 2 * B * hw * (cin + cout) / us / 1e3
            print(f"{kind:18s} B={B} HW={hw:6d}  {us:7.2f} us  {gb:7.1f} GB/s")
PY
grep -v amdgpu.ids $out/thin_launch_us.txt
