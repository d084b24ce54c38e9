#!/bin/bash
# round 4, GPU call 32: PMC traffic passes re-taken on the final tree (the narrow layers left the gemm_nt family), then the
# default bench line with that traffic file
cd /root/repo; O=/root/repo/gpurun_out/r4c32; mkdir -p $O
REPO=/root/repo; OUT=$REPO/gpurun_out/prof_r04; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-refine > /tmp/pmc_$c.json 2> /tmp/pmc_$c.err
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
            k += m.group(2).replace(" ", "")
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
out = {k: {"launches": v[0], "sum": v[1], "avg_per_launch": v[1] / v[0]} for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]}
json.dump(out, open("$OUT/$c.json", "w"), indent=1)
print("$c", len(out))
PY
done
cd $REPO
python scripts/make_pmc_json.py > $O/pmc.log 2>&1; tail -1 $O/pmc.log | cut -c1-400
cp profiles/r04_pmc_traffic.json $O/
timeout 600 python bench.py > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench_final.json")); r = d["roofline"]
print(round(d["value"], 1), round(d["ms_per_step"], 2), d["config"].get("sigma_I"), r["bound"], round(r["frac"], 3), r.get("traffic"), d["cpu_baseline"]["value"])
for k in ("gemm_nt_kernel", "rnarrow_kernel"):
    v = r["kernels"][k]; print(k, v["time_share"], v["hbm_frac"], v["hbm_frac_measured_bytes"], v["traffic"])
PY
