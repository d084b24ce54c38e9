#!/bin/bash
# round 3, GPU call 2: trunk_r6 store diagnostics + SQ counters of the register-resident kernels
cd /root/repo; O=/root/repo/gpurun_out/r3c2; mkdir -p $O
timeout 200 python scripts/bench_rmlp.py 1605632 > $O/bench_rmlp.log 2>&1; grep -v Warning $O/bench_rmlp.log
timeout 300 python -m pytest tests/test_rmlp_gpu.py -q > $O/pytest_rmlp.log 2>&1; echo "rmlp rc=$?"; tail -12 $O/pytest_rmlp.log
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/sq$i
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/sq$i -o p -- python /root/repo/scripts/bench_rmlp.py 1605632 > /tmp/sq$i.log 2>&1
  f=$(find /tmp/sq$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $O/sq$i.csv
done
python - <<PY
import csv, collections, glob, json, re
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in sorted(glob.glob("$O/sq*.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "rmlp" not in k and "fused_sdf" not in k and "chain" not in k: continue
        a = agg[k[:90]][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
out = {k: {c: v[1] / v[0] for c, v in d.items()} for k, d in agg.items()}
json.dump(out, open("$O/sq_counters.json", "w"), indent=1)
for k, d in out.items():
    b, w = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), d.get("SQ_BUSY_CYCLES", 0)
    g = d.get("GRBM_GUI_ACTIVE", 0)
    print(k[:70], {c: f"{v:.3e}" for c, v in d.items()})
PY
