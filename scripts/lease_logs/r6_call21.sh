#!/bin/bash
# round 6, GPU call 21: why is the 10-frame step at 2 048 pixels per frame SLOWER with compaction (358 vs 284 ms, call 20)?  kernel statistics
# of both runs
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r6c21; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in compact nocompact; do
  fl=""; [ $v = nocompact ] && fl="--no-compact"
  rm -rf /tmp/prof_$v
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o s -- python /root/repo/bench.py --mode c3 --no-refine --beta 0.005 --c3-pixels 2048 --steps 4 --warmup 2 --no-cpu-baseline --no-profile $fl > $O/bench_$v.json 2> $O/bench_$v.err
  find /tmp/prof_$v -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$v.csv \;
  python - <<PY
import csv, json
try:
    d = json.load(open("$O/bench_$v.json")); print("$v", round(d["ms_per_step"], 1), "ms/step", d["config"]["sample_compaction"].get("live_samples_last_call"))
except Exception as e: print("$v no line", e)
rows = list(csv.DictReader(open("$O/kernel_stats_$v.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", round(tot / 1e6, 1))
for r in rows[:22]:
    print("  ", r["Name"].replace("(anonymous namespace)::", "")[:90], r["Calls"], round(float(r["TotalDurationNs"]) / 1e6, 1), round(float(r["AverageNs"]) / 1e3, 1))
PY
done
