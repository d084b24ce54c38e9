#!/bin/bash
# round 5, GPU call 2: the whole GPU suite with f16x3 as the package default (arith-parametrised tests also run f32x6 and f32),
# same-box A/B of the bench line f16x3 vs f32x6, sustained clock / power re-taken with rocm-smi polled inside each loop
cd /root/repo; O=/root/repo/gpurun_out/r5c2; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "gpu suite rc=$rc"; tail -3 $O/pytest_gpu.log | cut -c1-200
if [ $rc -ne 0 ]; then grep -E "Error|assert|error|FAILED" $O/pytest_gpu.log | head -30 | cut -c1-250; fi
for prec in f16x3 f32x6; do
  timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --precision $prec > $O/bench_$prec.json 2> $O/bench_$prec.err; echo "bench $prec rc=$?"
  python - <<PY
import json
d = json.load(open("$O/bench_$prec.json")); r = d["roofline"]
print("$prec", round(d["value"], 1), round(d["ms_per_step"], 2), d["config"].get("sigma_I"), r["bound"], round(r["frac"], 3), r["end_to_end"]["time_in_mfma_kernels"])
for k, v in sorted(r["kernels"].items(), key=lambda kv: -kv[1]["time_share"])[:12]:
    print("   ", k, round(v["time_share"], 4), round(v.get("fp32_equivalent_tflops", 0), 1), round(v["avg_launch_ms"], 3))
PY
done
timeout 600 python scripts/sustained_clock.py 8 > $O/sustained_clock.log 2>&1; echo "clock rc=$?"; tail -8 $O/sustained_clock.log | cut -c1-1200
cp gpurun_out/r05_sustained_clock.json $O/ 2>/dev/null
