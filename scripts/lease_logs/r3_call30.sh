#!/bin/bash
# round 3, GPU call 30: same-box A/B of the register-resident single-layer GEMM in the bench line (HOLD_R6_GEMM=0 / 1, twice)
cd /root/repo; O=/root/repo/gpurun_out/r3c30; mkdir -p $O
for i in 1 2; do for v in 0 1; do
  HOLD_R6_GEMM=$v timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_$v_$i.json 2> /dev/null
  python - <<PY
import json
d = json.load(open("$O/bench_$v_$i.json")); k = d["roofline"]["kernels"]
print("HOLD_R6_GEMM=$v run $i:", round(d["value"], 1), round(d["ms_per_step"], 1), d["config"]["sigma_I"], "fused_sdf ms", round(k["fused_sdf_kernel"]["avg_launch_ms"], 3), "gemm share", round(k["gemm_nt_kernel"]["time_share"] + k.get("rgemm_kernel", {}).get("time_share", 0), 4))
PY
done; done
