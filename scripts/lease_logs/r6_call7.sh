#!/bin/bash
# round 6, GPU call 7: SQ counters of the f16x3 sweeps (three --pmc passes over the chain micro-benchmark, kernel-trace only)
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r6c7; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM"; do
  i=$((i+1)); rm -rf /tmp/sq$i
  HOLD_X6=1 timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/sq$i -o p -- python /root/repo/scripts/bench_chain.py > /tmp/sq$i.log 2>&1
  f=$(find /tmp/sq$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $O/sq$i.csv
  tail -2 /tmp/sq$i.log | cut -c1-200
done
python - <<PY
import csv, collections, glob, json
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in sorted(glob.glob("$O/sq*.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "rsweep" not in k: continue
        a = agg[k[:110]][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
out = {k: {c: v[1] / v[0] for c, v in d.items()} for k, d in agg.items()}
json.dump(out, open("$O/sq_counters.json", "w"), indent=1)
for k, d in out.items():
    g = d.get("GRBM_GUI_ACTIVE", 0) / 8
    busy = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * g) if g else 0
    w = d.get("SQ_WAVE_CYCLES", 1)
    print(k[40:110], f"mfma_busy {busy:.3f} active {d.get('SQ_ACTIVE_INST_ANY',0)/w:.3f} wait_any {d.get('SQ_WAIT_ANY',0)/w:.3f} wait_inst {d.get('SQ_WAIT_INST_ANY',0)/w:.3f} wait_lds {d.get('SQ_WAIT_INST_LDS',0)/w:.3f} valu/mfma {d.get('SQ_INSTS_VALU',0)/max(1,d.get('SQ_INSTS_MFMA',1)):.1f} valu_act {d.get('SQ_ACTIVE_INST_VALU',0)/w:.3f} vmem_act {d.get('SQ_ACTIVE_INST_VMEM',0)/w:.3f} lds_act {d.get('SQ_ACTIVE_INST_LDS',0)/w:.3f} sca_act {d.get('SQ_ACTIVE_INST_SCA',0)/w:.3f} bank_conf/lds_active {d.get('SQ_LDS_BANK_CONFLICT',0)/max(1,d.get('SQ_LDS_IDX_ACTIVE',1)):.4f}")
PY
rm -f $O/sq*.csv
