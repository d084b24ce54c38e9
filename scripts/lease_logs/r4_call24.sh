#!/bin/bash
# round 4, GPU call 24: above 512 k points, the two operand pairs of one dW in one grouped launch (128 workgroups each) instead of
# two single launches (HOLD_WGRAD_PAIRS=1): same-box A/B of the bench line
cd /root/repo; O=/root/repo/gpurun_out/r4c24; mkdir -p $O
for g in 1 0 1 0; do
  HOLD_WGRAD_PAIRS=$g timeout 400 python bench.py --no-cpu-baseline --no-refine > $O/bench_p$g.json 2> $O/bench_p$g.err; echo "bench pairs=$g rc=$?"
  python - <<PY
import json
d = json.load(open("$O/bench_p$g.json")); v = d["roofline"]["kernels"]["wgrad_kernel"]
print("pairs", $g, "rays/s", round(d["value"], 1), "wgrad share", round(v["time_share"], 4), "TF-eq", round(v["fp32_equivalent_tflops"], 1), "launches", v["launches"])
PY
done
