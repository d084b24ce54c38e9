#!/bin/bash
# Round-2 GPU call 12: where does the reference-shaped training step (C3: 10 frames x 128 rays) spend its time?
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
REPO=$(pwd); cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_c3
timeout 300 python $REPO/bench.py --mode c3 --steps 20 --warmup 5 --no-refine 2>/dev/null | tee $REPO/gpurun_out/bench_c3_r12.json | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("c3", d["ms_per_step"], d["value"], d["config"].get("c_abi_calls_per_step"), d.get("roofline",{}).get("end_to_end"))'
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -o s -- python $REPO/bench.py --mode c3 --steps 20 --warmup 5 --no-refine --no-profile > $REPO/gpurun_out/bench_c3_under_rocprof.json 2> /tmp/prof_c3.err
find /tmp/prof_c3 -name "*kernel_stats.csv" -exec cp {} $REPO/gpurun_out/kernel_stats_c3.csv \;
cd $REPO; python - <<'PY'
import csv, json
rows = list(csv.DictReader(open("gpurun_out/kernel_stats_c3.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
d = json.loads(open("gpurun_out/bench_c3_under_rocprof.json").read().strip().splitlines()[-1])
print("kernel time total ms", tot / 1e6, "launches", calls, "-> per step (25 steps)", tot / 1e6 / 25, calls / 25, "wall ms/step", d["ms_per_step"])
for r in rows[:25]:
    print(f"{float(r['TotalDurationNs'])/1e6:8.1f} {float(r['Percentage']):6.2f} {int(r['Calls']):7d} {float(r['AverageNs'])/1e3:9.1f}us {r['Name'][:100]}")
PY
