#!/bin/bash
# round 5, GPU call 14: FINAL TREE -- full GPU suite (default precision f16x3; the arith-parametrised tests also run f32x6 and
# f32), build() + smoke(), rocprofv3 kernel stats of the default bench command, the two PMC passes, the default bench line
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r5c14; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "gpu suite rc=$rc"; tail -3 $O/pytest_gpu.log | cut -c1-200
if [ $rc -ne 0 ]; then grep -E "Error|assert|error|FAILED" $O/pytest_gpu.log | head -20 | cut -c1-250; fi
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log | cut -c1-200
REPO=/root/repo; OUT=$REPO/gpurun_out/prof_r05; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_stats
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o s -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> /tmp/prof_stats.err
find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
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
        if m and m.group(2) and k in ("rmlp_kernel", "rmlp_h3_kernel", "rsweep_kernel", "chain_x6_kernel", "rgemm_kernel", "wgrad_r6_kernel",
                                      "wgrad_h3_kernel", "wgrad_r6_group_kernel"):
            k += m.group(2).replace(" ", "")
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
out = {k: {"launches": v[0], "sum": v[1], "avg_per_launch": v[1] / v[0]} for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]}
json.dump(out, open("$OUT/$c.json", "w"), indent=1)
print("$c", len(out))
PY
done
cd $REPO
python scripts/make_pmc_json.py > $O/pmc.log 2>&1; tail -1 $O/pmc.log | cut -c1-500
cp profiles/r05_pmc_traffic.json $O/ 2>/dev/null; cp $OUT/FETCH_SIZE.json $O/pmc_FETCH_SIZE.json; cp $OUT/WRITE_SIZE.json $O/pmc_WRITE_SIZE.json
timeout 700 python bench.py > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench_final.json")); r = d["roofline"]
print(round(d["value"], 1), round(d["ms_per_step"], 2), d["config"].get("sigma_I"), r["bound"], round(r["frac"], 3), r.get("traffic"), r.get("hbm_frac_measured_bytes"), d["cpu_baseline"]["value"], d["cpu_baseline"].get("step_s"), r["end_to_end"]["time_in_mfma_kernels"])
PY
