#!/bin/bash
# Round-2 GPU call 14: SQ stall / busy counters of the layer-chain kernels (what is the matrix pipe waiting for?)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/pmc_chain
REPO=$(pwd); cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_WAIT[A-Z_]*\|SQ_ACTIVE_INST[A-Z_]*\|SQ_INSTS_[A-Z_]*\|SQ_VALU_MFMA[A-Z_]*\|SQ_BUSY[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*\|GRBM_GUI_ACTIVE" | sort -u | tr '\n' ' ' > $REPO/gpurun_out/pmc_chain/available.txt
cat $REPO/gpurun_out/pmc_chain/available.txt; echo
run() {  # $1 = tag, rest = counters
  tag=$1; shift
  rm -rf /tmp/pmc_$tag
  HOLD_X6=1 timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$tag -o p -- python $REPO/scripts/bench_chain.py > /tmp/pmc_$tag.log 2>&1
  python - <<PY
import csv, glob, collections, re
f = glob.glob("/tmp/pmc_$tag/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
if f:
    for r in csv.DictReader(open(f[0])):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "").split("(")[0]
        if "chain" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in sorted(agg):
    print("$tag", k, {c: round(v / max(1, n[(k, c)])) for c, v in agg[k].items()})
PY
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU
tail -3 /tmp/pmc_a.log; tail -3 /tmp/pmc_b.log
