#!/bin/bash
# round 4, probe 19: GroupNorm statistics from the Linear epilogue (proj_out + residual -> the next norm) — tests, same-box arms
out=gpurun_out/r4w
mkdir -p $out
fault() { grep -q "Memory access fault" "$1" && { echo "GPU FAULT in $1"; exit 9; }; }
timeout -k 10 120 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "linear_epilogue_takes and 1x4096x320x320x32-f16" > $out/first.log 2>&1; rc=$?
tail -2 $out/first.log; fault $out/first.log; [ $rc -eq 0 ] || { grep -n "^E " $out/first.log | head; exit $rc; }
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "linear or thin" > $out/linear_tests.log 2>&1; rc=$?
tail -2 $out/linear_tests.log; fault $out/linear_tests.log; [ $rc -eq 0 ] || { grep -n "^E " $out/linear_tests.log | head; exit $rc; }
for arm in "" no-gn-producer "" no-gn-producer; do
  echo "== unet_bench ${arm:-default}" | tee -a $out/ab_gn_all.txt
  timeout -k 5 300 python3 tools/unet_bench.py $arm 2>/dev/null | grep "ms" | tee -a $out/ab_gn_all.txt
done
timeout -k 5 200 python3 - > $out/stats_launches.txt 2>&1 <<'PY'
import sys; sys.path.insert(0, ".")
import torch
from guided_attention_amd import ops
from guided_attention_amd.pipeline_guided_attention import GuidedAttention
from guided_attention_amd.text import SyntheticTextEncoder, WordTokenizer
from guided_attention_amd.unet import UNet2DConditionModel, UNetConfig
with torch.device("cuda"):
    unet = UNet2DConditionModel(UNetConfig.sd15()).half()
pipe = GuidedAttention(unet, None, None, SyntheticTextEncoder(768), WordTokenizer()).to("cuda", torch.float16)
emb = torch.randn(3, 77, 768, device="cuda", dtype=torch.half); lat = torch.randn(3, 4, 64, 64, device="cuda", dtype=torch.half)
with torch.no_grad(), ops.census_scope() as cs:
    pipe.unet(lat, 500, encoder_hidden_states=emb)
kinds = {}
for k, n in cs.launches.items(): kinds[k[0]] = kinds.get(k[0], 0) + n
print("launch census of one batch-3 forward:", dict(sorted(kinds.items())))
PY
grep census $out/stats_launches.txt
timeout -k 10 600 python -m pytest tests/test_pipeline_gpu.py tests/test_unet_forward_golden.py -m gpu -q -x -k "full_width or half_precision or golden or forward" > $out/pipe_tests.log 2>&1; rc=$?
tail -2 $out/pipe_tests.log; fault $out/pipe_tests.log; [ $rc -eq 0 ] || { grep -n "^E " $out/pipe_tests.log | head; exit $rc; }
