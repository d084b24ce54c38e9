#!/bin/bash
# Round-4 evidence for profiles/: rocprofv3 kernel stats of the DEFAULT bench command + separate --pmc passes
# (kernel-trace only, one counter per pass, as MI355X_MICROARCH.md prescribes).  Outputs under gpurun_out/prof_r04/.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_r04
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_stats
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o s -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> /tmp/prof_stats.err
find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} $OUT/bench_kernel_stats.csv \;
head -14 $OUT/bench_kernel_stats.csv | cut -c1-200
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 500 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile > /tmp/pmc_$c.json 2> /tmp/pmc_$c.err
  python - <<PY
import csv, collections, glob, json, re
f = glob.glob("/tmp/pmc_$c/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: [0, 0.0])
if f:
    for r in csv.DictReader(open(f[0])):
        nm = r["Kernel_Name"].replace("void ", "")
        m = re.search(r"(?:anonymous namespace\)::)?(\w+)(<[^>]*>)?(?:\()", nm)
        k = m.group(1) if m else nm[:40]
        if m and m.group(2) and k in ("rmlp_kernel", "rsweep_kernel", "chain_x6_kernel", "rgemm_kernel", "wgrad_r6_kernel"):
            k += m.group(2).replace(" ", "")  # instantiations of the trunk kernels are different sweeps
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
out = {k: {"launches": v[0], "sum": v[1], "avg_per_launch": v[1] / v[0]} for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]}
json.dump(out, open("$OUT/$c.json", "w"), indent=1)
print("$c", json.dumps(list(out.items())[:4]))
PY
done
ls -la $OUT
