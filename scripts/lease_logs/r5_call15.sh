#!/bin/bash
# round 5, GPU call 15: FINAL TREE -- full GPU suite (default precision f16x3), rocprofv3 kernel stats of the default bench command,
# the default bench line (with its cpu_baseline leg), then the secondary lines on the same box
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r5c15; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "gpu suite rc=$rc"; tail -3 $O/pytest_gpu.log | cut -c1-200
if [ $rc -ne 0 ]; then grep -E "Error|assert|error|FAILED" $O/pytest_gpu.log | head -20 | cut -c1-250; fi
REPO=/root/repo
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_stats
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o s -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> /tmp/prof_stats.err
find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
cd $REPO
timeout 700 python bench.py > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench_final.json")); r = d["roofline"]
print(round(d["value"], 1), round(d["ms_per_step"], 2), d["config"].get("sigma_I"), r["bound"], round(r["frac"], 3), r.get("traffic"), d["cpu_baseline"]["value"], d["cpu_baseline"].get("step_s"), r["end_to_end"]["time_in_mfma_kernels"])
PY
run() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name rc=$? $(python -c "
import json
try:
    d = json.load(open('$O/bench_$name.json')); print(round(d['value'], 1), round(d['ms_per_step'], 2), d['config'].get('sigma_I'), d['config'].get('pose_refine', {}).get('iters_per_s') if isinstance(d['config'].get('pose_refine'), dict) else '')
except Exception as e: print('no line', e)
")"; }
run f32x6 --precision f32x6 --steps 10 --warmup 3
run fp32_mfma --fp32-mfma --steps 5 --warmup 2
run render --mode render --steps 10 --warmup 3
run twohands --two-hands --chunk 16384 --steps 5 --warmup 2
run c3 --mode c3 --steps 30 --warmup 5
run c3_f32x6 --mode c3 --steps 30 --warmup 5 --precision f32x6 --no-refine
run c5 --mode c5 --steps 1 --warmup 1
run beta005 --beta 0.005 --steps 5 --warmup 2
run beta005_nocompact --beta 0.005 --no-compact --steps 5 --warmup 2
