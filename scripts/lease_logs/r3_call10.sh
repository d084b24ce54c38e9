#!/bin/bash
# round 3, GPU call 10: explicit per-gap schedule + round-major micro-operation epilogue in the r6 kernels
cd /root/repo; O=/root/repo/gpurun_out/r3c10; mkdir -p $O
timeout 200 python scripts/bench_rmlp.py 1605632 2>&1 | grep -v Warning | tee $O/bench_rmlp.log
HOLD_X6=1 timeout 200 python scripts/bench_chain.py 2>&1 | grep -v Warning | grep "r6\|DSP" | tee $O/bench_chain.log
timeout 600 python -m pytest tests/test_rmlp_gpu.py tests/test_chain_gpu.py -q > $O/pytest_a.log 2>&1; echo "a rc=$?"; tail -4 $O/pytest_a.log
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/sq$i
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/sq$i -o p -- python /root/repo/scripts/bench_rmlp.py 1605632 > /tmp/sq$i.log 2>&1
  f=$(find /tmp/sq$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $O/sq$i.csv
done
python - <<PY
import csv, collections, glob, json
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in sorted(glob.glob("$O/sq*.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "rmlp" not in k: continue
        a = agg[k[:100]][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
out = {k: {c: v[1] / v[0] for c, v in d.items()} for k, d in agg.items()}
json.dump(out, open("$O/sq_counters.json", "w"), indent=1)
for k, d in out.items():
    g = d.get("GRBM_GUI_ACTIVE", 0) / 8
    busy = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * g) if g else 0
    w = d.get("SQ_WAVE_CYCLES", 1)
    print(k[40:100], f"mfma_busy {busy:.3f} active {d.get('SQ_ACTIVE_INST_ANY',0)/w:.3f} wait_any {d.get('SQ_WAIT_ANY',0)/w:.3f} wait_inst {d.get('SQ_WAIT_INST_ANY',0)/w:.3f} valu/mfma {d.get('SQ_INSTS_VALU',0)/max(1,d.get('SQ_INSTS_MFMA',1)):.1f}")
PY
