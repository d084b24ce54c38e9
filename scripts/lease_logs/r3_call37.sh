#!/bin/bash
# round 3, GPU call 37: SQ counters of the register-resident single-layer GEMM
cd /root/repo; O=/root/repo/gpurun_out/r3c37; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/sq$i
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/sq$i -o p -- python /root/repo/scripts/bench_rgemm.py > /tmp/sq$i.log 2>&1
  f=$(find /tmp/sq$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $O/sq$i.csv
done
python - <<PY
import csv, collections, glob, json
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in sorted(glob.glob("$O/sq*.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "rgemm" not in k and "gemm_nt" not in k: continue
        a = agg[k[:80]][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
out = {k: {c: v[1] / v[0] for c, v in d.items()} for k, d in agg.items()}
json.dump(out, open("$O/sq_counters.json", "w"), indent=1)
for k, d in out.items():
    g = d.get("GRBM_GUI_ACTIVE", 0) / 8
    busy = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * g) if g else 0
    w = d.get("SQ_WAVE_CYCLES", 1)
    print(k[30:80], f"mfma_busy {busy:.3f} active {d.get('SQ_ACTIVE_INST_ANY',0)/w:.3f} wait_any {d.get('SQ_WAIT_ANY',0)/w:.3f} valu/mfma {d.get('SQ_INSTS_VALU',0)/max(1,d.get('SQ_INSTS_MFMA',1)):.2f} salu/mfma {d.get('SQ_INSTS_SALU',0)/max(1,d.get('SQ_INSTS_MFMA',1)):.2f} lds_conf {d.get('SQ_LDS_BANK_CONFLICT',0)/max(1,d.get('SQ_LDS_IDX_ACTIVE',1)):.3f}")
PY
