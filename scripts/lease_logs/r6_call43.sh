#!/bin/bash
# round 6, GPU call 43: FINAL TREE evidence run (call 35 + whole-line result stores of hold_gemm_h3 and hold_trunk_h3) -- full GPU suite, build() + smoke(), rocprofv3 kernel
# stats of the default bench command, the two PMC passes, the default bench line (with its cpu_baseline leg), secondary lines
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r6c43; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "gpu suite rc=$rc"; tail -3 $O/pytest_gpu.log | cut -c1-200
if [ $rc -ne 0 ]; then grep -E "^E  |FAILED" $O/pytest_gpu.log | head -30 | cut -c1-300; fi
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log | cut -c1-200
REPO=/root/repo; OUT=$REPO/gpurun_out/prof_r06; mkdir -p $OUT
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
        if m and m.group(2) and k in ("rmlp_kernel", "rmlp_h3_kernel", "rsweep_kernel", "rsweep_h3_kernel", "chain_x6_kernel", "rgemm_kernel", "rgemm_h3_kernel",
                                      "wgrad_r6_kernel", "wgrad_h3_kernel", "wgrad_r6_group_kernel"):
            k += m.group(2).replace(" ", "")
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
out = {k: {"launches": v[0], "sum": v[1], "avg_per_launch": v[1] / v[0]} for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]}
json.dump(out, open("$OUT/$c.json", "w"), indent=1)
print("$c", len(out))
PY
done
cd $REPO
python scripts/make_pmc_json.py > $O/pmc.log 2>&1; tail -1 $O/pmc.log | cut -c1-600
cp profiles/r06_pmc_traffic.json $O/ 2>/dev/null; cp $OUT/FETCH_SIZE.json $O/pmc_FETCH_SIZE.json; cp $OUT/WRITE_SIZE.json $O/pmc_WRITE_SIZE.json
timeout 700 python bench.py > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench_final.json")); r = d["roofline"]
print(round(d["value"], 1), round(d["ms_per_step"], 2), d["config"].get("sigma_I"), r["bound"], round(r["frac"], 3), r.get("traffic"), r.get("hbm_frac_measured_bytes"), d["cpu_baseline"]["value"], d["cpu_baseline"].get("step_s"), r["end_to_end"]["time_in_mfma_kernels"])
PY
run() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name rc=$? $(python -c "
import json
try:
    d = json.load(open('$O/bench_$name.json')); print(round(d['value'], 1), round(d['ms_per_step'], 2), d['config'].get('sigma_I'), d['config'].get('f16x3_launches_recomputed_in_f32x6_per_step'), d['config'].get('pose_refine', {}).get('iters_per_s') if isinstance(d['config'].get('pose_refine'), dict) else '')
except Exception as e: print('no line', e)
")"; }
run c2_alt --c2-alt --steps 5 --warmup 2
run f32x6 --precision f32x6 --steps 5 --warmup 2
run render --mode render --steps 10 --warmup 3
run twohands --two-hands --chunk 16384 --steps 5 --warmup 2
run c3 --mode c3 --steps 30 --warmup 5
run c3_again --mode c3 --steps 30 --warmup 5 --no-refine
run fp32_mfma --fp32-mfma --steps 3 --warmup 1
run beta005_nocompact --beta 0.005 --no-compact --steps 5 --warmup 2
run c5 --mode c5 --steps 1 --warmup 1
run beta005 --beta 0.005 --steps 5 --warmup 2
