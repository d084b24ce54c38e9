#!/bin/bash
# Round-2 GPU call 15: SQ busy / wait counters of the other three MFMA kernel families (+ MFMA/VALU co-execution of the chains)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/pmc_kernels
REPO=$(pwd); cd /tmp && export TMPDIR=/tmp
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE"
run() {  # $1 tag, $2 kernel substring, rest = command
  tag=$1; pat=$2; shift; shift
  rm -rf /tmp/pmc_$tag
  HOLD_X6=1 timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_$tag -o p -- "$@" > /tmp/pmc_$tag.log 2>&1
  python - <<PY
import csv, glob, collections, re, json
f = glob.glob("/tmp/pmc_$tag/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
if f:
    for r in csv.DictReader(open(f[0])):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "").split("(")[0]
        if "$pat" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
out = {k: {c: round(v / max(1, n[(k, c)])) for c, v in agg[k].items()} | {"launches": max(n[(k, c)] for c in agg[k])} for k in sorted(agg)}
json.dump(out, open("$REPO/gpurun_out/pmc_kernels/$tag.json", "w"), indent=1)
for k, v in out.items():
    busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / max(1, v.get("GRBM_GUI_ACTIVE", 1) / 8)
    w = max(1, v.get("SQ_WAVE_CYCLES", 1))
    print("$tag", k, "launches", v["launches"], "mfma_busy %.2f" % busy, "issuing %.2f parked %.2f issue_stalled %.2f" % (v.get("SQ_ACTIVE_INST_ANY", 0) / w, v.get("SQ_WAIT_ANY", 0) / w, v.get("SQ_WAIT_INST_ANY", 0) / w), "coexec/busy %.2f" % (v.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0) / max(1, v.get("SQ_VALU_MFMA_BUSY_CYCLES", 1))))
PY
}
run fused fused_sdf python $REPO/scripts/bench_fused.py
run gemm gemm_nt python $REPO/scripts/bench_gemm.py 1605632
run wgrad wgrad_lds python $REPO/scripts/bench_gemm.py 1605632
run chain chain_x6 python $REPO/scripts/bench_chain.py
