#!/bin/bash
# final tree: the full GPU suite, smoke, the default bench line (what the driver runs at the round end)
out=gpurun_out/r4final
mkdir -p $out
fault() { grep -q "Memory access fault" "$1" && { echo "GPU FAULT in $1"; exit 9; }; }
( while sleep 60; do echo "[final] $(tail -c 100 $out/gpu_suite.log 2>/dev/null | tr '\n' ' ')"; done ) &
ticker=$!
timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=8 > $out/gpu_suite.log 2>&1; rc=$?
kill $ticker
tail -12 $out/gpu_suite.log; fault $out/gpu_suite.log; [ $rc -eq 0 ] || { grep -n "^E " $out/gpu_suite.log | head -20; exit $rc; }
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 || exit 1
timeout -k 5 400 python3 bench.py > $out/bench.json 2> $out/bench.err; python3 -c "import json; d=json.load(open('$out/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])"

