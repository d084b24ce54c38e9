#!/bin/bash
# round 5, GPU call 17: where the reference's own 1 280-ray step (C3) stands on the final tree: rocprofv3 kernel stats + torch operator sites
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r5c17; mkdir -p $O
REPO=/root/repo
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_c3
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -o s -- python $REPO/bench.py --mode c3 --steps 20 --warmup 5 --no-cpu-baseline --no-refine > $O/bench_c3_under_rocprof.json 2> /tmp/prof_c3.err
find /tmp/prof_c3 -name "*kernel_stats.csv" -exec cp {} $O/c3_kernel_stats.csv \;
cd $REPO
timeout 300 python bench.py --mode c3 --steps 20 --warmup 5 --no-cpu-baseline --no-refine --op-sites $O/c3_op_sites.txt --sync-debug $O/c3_sync_sites.txt > $O/bench_c3_diag.json 2> $O/bench_c3_diag.err; echo "c3 diag rc=$?"
python - <<PY
import csv, json
rows = list(csv.DictReader(open("$O/c3_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows); n = sum(int(r["Calls"]) for r in rows)
d = json.load(open("$O/bench_c3_under_rocprof.json"))
print("c3 under rocprof", round(d["ms_per_step"], 2), "ms/step; kernel time per step", round(tot / 1e6 / 25, 2), "ms; launches per step", n / 25)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print(f'{r["Name"][:70]:70s} calls/step {int(r["Calls"])/25:7.1f} ms/step {float(r["TotalDurationNs"])/1e6/25:6.3f}')
PY
head -5 $O/c3_op_sites.txt | cut -c1-200
