#!/bin/bash
# round 5, GPU call 16: the whole GPU suite once more with HOLD_PRECISION=f32x6 as the package default (the arithmetic of rounds 2-4
# stays selectable and green), then the default bench line of the final tree (cpu_baseline leg with the sub-sampled geometry)
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r5c16; mkdir -p $O
HOLD_PRECISION=f32x6 timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_f32x6.log 2>&1; rc=$?; echo "gpu suite (f32x6 default) rc=$rc"; tail -3 $O/pytest_gpu_f32x6.log | cut -c1-200
if [ $rc -ne 0 ]; then grep -E "Error|assert|error|FAILED" $O/pytest_gpu_f32x6.log | head -20 | cut -c1-250; fi
timeout 700 python bench.py > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench_final.json")); r = d["roofline"]
print(round(d["value"], 1), round(d["ms_per_step"], 2), d["config"].get("sigma_I"), r["bound"], round(r["frac"], 3), r.get("traffic"), d["cpu_baseline"]["value"], d["cpu_baseline"].get("step_s"), d["cpu_baseline"].get("parts_s"), r["end_to_end"]["time_in_mfma_kernels"])
PY
