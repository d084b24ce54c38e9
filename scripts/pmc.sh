#!/bin/bash
# HBM traffic counters for the bench (separate passes per counter, kernel-trace only), summaries -> gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
mkdir -p /root/repo/gpurun_out/pmc
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile > /tmp/pmc_$c.json 2> /tmp/pmc_$c.err
  ls /tmp/pmc_$c | head
  python - <<PY
import csv, collections, glob, json
f = glob.glob("/tmp/pmc_$c/*counter_collection.csv")
agg = collections.defaultdict(lambda: [0, 0.0])
if f:
    for r in csv.DictReader(open(f[0])):
        import re
        m = re.search(r"(?:anonymous namespace\)::)?(\w+)(?:<|\()", r["Kernel_Name"].replace("void ", ""))
        k = m.group(1) if m else r["Kernel_Name"][:40]
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
out = {k: {"launches": v[0], "sum": v[1], "avg_per_launch": v[1] / v[0]} for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]}
json.dump(out, open("/root/repo/gpurun_out/pmc/$c.json", "w"), indent=1)
print("$c", json.dumps(list(out.items())[:3]))
PY
done
