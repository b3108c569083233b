#!/bin/bash
# Round evidence in one GPU call (run on the GPU box from the repo root): the default bench line, the bench under rocprofv3
# with the timed-window kernel statistics and the idle-gap report, the per-pass timings, the other models' bench lines and
# the PMC traffic of the dominant shapes.  Everything lands in gpurun_out/ev/ (copy what is to be judged into profiles/).
#   tools/collect_evidence.sh [bench] [trace] [passes] [models] [pmc]      (no argument: all; a call is limited to 20 minutes)
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/ev
mkdir -p $out
want() { [ -z "$ALL" ] || return 0; case " $ARGS " in *" $1 "*) return 0;; esac; return 1; }
ARGS="$*"; [ $# -eq 0 ] && ALL=1
if want bench; then
echo "[1] default bench"; python bench.py > $out/bench_default_args.json 2> $out/bench_default_args.err || exit 1
python - <<'PY'
import json
d = json.load(open("gpurun_out/ev/bench_default_args.json"))
print("   ", d["value"], d["unit"], d["ms_per_step"], "ms/image; roofline", d["roofline"]["kernel"], d["roofline"]["frac"], "cpu", d["cpu_baseline"]["value"])
PY
fi
if want trace; then
echo "[2] bench under rocprofv3 --kernel-trace (3 timed images)"
rm -rf /tmp/ev_trace
rocprofv3 --kernel-trace --output-format csv -d /tmp/ev_trace -o b -- python3 bench.py --steps 3 --warmup 1 --no-roofline --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.err || exit 1
trace=$(find /tmp/ev_trace -name "*kernel_trace.csv" | head -1)
python tools/trace_window_stats.py $trace 1 3 52 $out/kernel_stats_top.md "Round 4 - python bench.py --steps 3 --warmup 1 (1x MI355X, fp16): kernel time inside the three TIMED images only" > $out/trace_window.log 2>&1 || { tail -5 $out/trace_window.log; exit 1; }
python tools/gap_report.py $trace between=cfg_ddim,53,208 > $out/gap_report.txt 2>&1 || true
head -12 $out/kernel_stats_top.md
fi
if want passes; then
echo "[3] per-pass timings"; python tools/unet_bench.py > $out/unet_bench.txt 2>&1 || true
grep -E "ms$|ms " $out/unet_bench.txt | head -8
for pass in eval grad joint; do
  rm -rf /tmp/ev_pass_$pass
  rocprofv3 --kernel-trace --output-format csv -d /tmp/ev_pass_$pass -o p -- python3 tools/unet_bench.py only=$pass > $out/pass_$pass.txt 2>&1 || continue
  python tools/trace_window_stats.py $(find /tmp/ev_pass_$pass -name "*kernel_trace.csv" | head -1) marker axpby 60 $out/pass_${pass}_stats.md "$pass pass: 60 hipGraph replays (tools/unet_bench.py only=$pass)" > /dev/null 2>&1 || true
done
fi
if want models; then
echo "[4] sd21 768^2 bench"; python bench.py --model sd21 --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_sd21_768.json 2> $out/bench_sd21.err || true
echo "[5] sdxl bench"; python bench.py --model sdxl --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_sdxl.json 2> $out/bench_sdxl.err || true
python - <<'PY'
import json
for m in ("sd21_768", "sdxl"):
    try:
        d = json.load(open(f"gpurun_out/ev/bench_{m}.json")); print("   ", m, d["value"], d["ms_per_step"], d["dtype"])
    except Exception as e:
        print("   ", m, "failed", e)
PY
fi
if want variants; then
echo "[5b] W-every-step and the truncated guidance forward"
python bench.py --workload every-step --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_every_step.json 2> $out/bench_every_step.err || true
python bench.py --guidance-forward truncated --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_truncated_guidance_forward.json 2> $out/bench_truncated.err || true
python - <<'PY'
import json
for m in ("every_step", "truncated_guidance_forward"):
    try:
        d = json.load(open(f"gpurun_out/ev/bench_{m}.json")); print("   ", m, d["value"], d["ms_per_step"], d["unet_calls_per_image"])
    except Exception as e:
        print("   ", m, "failed", e)
PY
fi
if want pmc; then
echo "[6] PMC traffic"; for m in sd15 sd21 sdxl; do python tools/pmc_traffic.py $m > $out/pmc_traffic_$m.log 2>&1 && cp gpurun_out/r4_pmc_traffic_$m.json $out/ || tail -3 $out/pmc_traffic_$m.log; echo "   $m done"; done
fi
echo done
