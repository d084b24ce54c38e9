#!/bin/bash
# round 5, GPU call 3: first run of the two-limb fp16 weight gradient (wgrad_h3_body): gemm tests in both split arithmetics,
# A/B timing, the end-to-end gradient parity test, bench A/B with and without it
cd /root/repo; O=/root/repo/gpurun_out/r5c3; mkdir -p $O
timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q -k "wgrad" > $O/pytest_wgrad.log 2>&1; echo "wgrad tests rc=$?"; tail -12 $O/pytest_wgrad.log | cut -c1-250
timeout 300 python scripts/bench_wgrad_h3.py > $O/bench_wgrad_h3.log 2>&1; echo "bench_wgrad rc=$?"; cat $O/bench_wgrad_h3.log | cut -c1-250
timeout 900 python -m pytest tests/test_path_gpu.py -x -q -k "train_step_gradients or eval_forward_matches_oracle" > $O/pytest_path.log 2>&1; echo "path tests rc=$?"; tail -5 $O/pytest_path.log | cut -c1-250
for v in 1 0; do
  HOLD_H3_WGRAD=$v timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_h3wgrad_$v.json 2> $O/bench_h3wgrad_$v.err; echo "bench HOLD_H3_WGRAD=$v rc=$?"
  python - <<PY
import json
d = json.load(open("$O/bench_h3wgrad_$v.json")); r = d["roofline"]
print("H3_WGRAD=$v", round(d["value"], 1), round(d["ms_per_step"], 2), d["config"].get("sigma_I"), r["bound"], round(r["frac"], 3), r["end_to_end"]["time_in_mfma_kernels"], d["config"]["loss"])
for k, v in sorted(r["kernels"].items(), key=lambda kv: -kv[1]["time_share"])[:4]:
    print("   ", k, round(v["time_share"], 4), round(v.get("fp32_equivalent_tflops", 0), 1), round(v["avg_launch_ms"], 3))
PY
done
