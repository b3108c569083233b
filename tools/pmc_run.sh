#!/bin/bash
# usage: [PMC_FILTER=substr] tools/pmc_run.sh <out-tag> <kernel_once.py args...>     (on the GPU box; writes gpurun_out/pmc/<tag>/*.txt)
# Counter passes are separate rocprofv3 runs (FETCH_SIZE and WRITE_SIZE cannot share a pass; 8 SQ slots per pass); each
# with --kernel-trace only, the program directly after `--` (no env / bash -c hop under the profiler).
set -e
tag=$1; shift
out=gpurun_out/pmc/$tag
mkdir -p $out
export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  rm -rf /tmp/pmc_${tag}_$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_${tag}_$i -o p -- python3 tools/kernel_once.py "$@" > $out/pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/pass$i.log; }
  python3 tools/pmc_summary.py /tmp/pmc_${tag}_$i "${PMC_FILTER:-attn}" > $out/pass$i.txt 2>&1 || true
done
cat $out/pass*.txt
