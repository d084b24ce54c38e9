#!/bin/bash
# round 4, GPU call 5: reference-golden tests with K = 15 tie identification, the ray-tile loss test, weight-norm / optimiser
# fixes; then SQ counters of the register-resident sweeps (three --pmc passes over the chain micro-benchmark)
cd /root/repo; O=/root/repo/gpurun_out/r4c5; mkdir -p $O
timeout 600 python -m pytest tests/test_path_gpu.py tests/test_train_targets_gpu.py tests/test_wnorm_gpu.py tests/test_parallel_gpu.py -q -k "golden or ray_tile or flat_adam or chunk or wnorm or weight_norm or parallel or trace" > $O/pytest_sel.log 2>&1; echo "selected tests rc=$?"; tail -6 $O/pytest_sel.log | cut -c1-220
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/sq$i
  HOLD_X6=1 timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/sq$i -o p -- python /root/repo/scripts/bench_chain.py > /tmp/sq$i.log 2>&1
  f=$(find /tmp/sq$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $O/sq$i.csv
done
python - <<PY
import csv, collections, glob, json
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in sorted(glob.glob("$O/sq*.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "rsweep" not in k and "chain_x6" not in k: continue
        a = agg[k[:110]][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
out = {k: {c: v[1] / v[0] for c, v in d.items()} for k, d in agg.items()}
json.dump(out, open("$O/sq_counters.json", "w"), indent=1)
for k, d in out.items():
    g = d.get("GRBM_GUI_ACTIVE", 0) / 8
    busy = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * g) if g else 0
    w = d.get("SQ_WAVE_CYCLES", 1)
    print(k[40:110], f"mfma_busy {busy:.3f} active {d.get('SQ_ACTIVE_INST_ANY',0)/w:.3f} wait_any {d.get('SQ_WAIT_ANY',0)/w:.3f} wait_inst {d.get('SQ_WAIT_INST_ANY',0)/w:.3f} valu/mfma {d.get('SQ_INSTS_VALU',0)/max(1,d.get('SQ_INSTS_MFMA',1)):.1f} vmem_act {d.get('SQ_ACTIVE_INST_VMEM',0)/w:.3f} lds_act {d.get('SQ_ACTIVE_INST_LDS',0)/w:.3f} bank_conf/lds_active {d.get('SQ_LDS_BANK_CONFLICT',0)/max(1,d.get('SQ_LDS_IDX_ACTIVE',1)):.4f}")
PY
rm -f $O/sq*.csv
