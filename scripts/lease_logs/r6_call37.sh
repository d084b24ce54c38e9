#!/bin/bash
# round 6, GPU call 37: SQ counters of hold_gemm_h3 (the four epilogue variants of the final tree), hold_wgrad_h3 and the trunk kernels (three --pmc
# passes over the micro-benchmarks, kernel-trace only): what these kernels' waves do with their cycles
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r6c37; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  for b in rgemm wgrad_h3; do
    rm -rf /tmp/sq${i}_$b
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/sq${i}_$b -o p -- python /root/repo/scripts/bench_$b.py > /tmp/sq${i}_$b.log 2>&1
    f=$(find /tmp/sq${i}_$b -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $O/sq${i}_$b.csv
    tail -1 /tmp/sq${i}_$b.log | cut -c1-160
  done
done
python - <<PY
import csv, collections, glob, json
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in sorted(glob.glob("$O/sq*.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        if not any(s in k for s in ("rgemm_h3_kernel", "rgemm_kernel", "wgrad_h3_kernel", "wgrad_r6_kernel")): continue
        a = agg[k[:60]][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
out = {k: {c: v[1] / v[0] for c, v in d.items()} for k, d in agg.items()}
json.dump(out, open("$O/sq_counters.json", "w"), indent=1)
for k, d in sorted(out.items()):
    g = d.get("GRBM_GUI_ACTIVE", 0) / 8
    busy = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * g) if g else 0
    w = d.get("SQ_WAVE_CYCLES", 1)
    print(k[:44].ljust(44), f"mfma_busy {busy:.3f} active {d.get('SQ_ACTIVE_INST_ANY',0)/w:.3f} wait_any {d.get('SQ_WAIT_ANY',0)/w:.3f} wait_inst {d.get('SQ_WAIT_INST_ANY',0)/w:.3f} wait_lds {d.get('SQ_WAIT_INST_LDS',0)/w:.3f} vmem_active {d.get('SQ_ACTIVE_INST_VMEM',0)/w:.3f} valu/mfma {d.get('SQ_INSTS_VALU',0)/max(d.get('SQ_INSTS_MFMA',1),1):.2f} vmem_rd+wr/mfma {(d.get('SQ_INSTS_VMEM_RD',0)+d.get('SQ_INSTS_VMEM_WR',0))/max(d.get('SQ_INSTS_MFMA',1),1):.3f}")
PY
rm -f $O/sq*.csv
