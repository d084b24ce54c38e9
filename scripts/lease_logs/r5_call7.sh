#!/bin/bash
# round 5, GPU call 7: the padded-k-step fix of rmlp_h3 v2 (stale LDS test), bench with the A/B switches, trajectory test
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r5c7; mkdir -p $O
timeout 600 python -m pytest tests/test_rmlp_gpu.py -x -q > $O/pytest_rmlp.log 2>&1; echo "rmlp tests rc=$?"; tail -3 $O/pytest_rmlp.log | cut -c1-250
for cfg in "1 1" "0 1"; do
  set -- $cfg
  HOLD_H3_TRUNK=$1 HOLD_H3_WGRAD=$2 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_t$1_w$2.json 2> $O/bench_t$1_w$2.err
  echo "H3_TRUNK=$1 H3_WGRAD=$2 rc=$? $(grep -c 'Memory access fault' $O/bench_t$1_w$2.err)"
  python - <<PY
import json
try:
    d = json.load(open("$O/bench_t$1_w$2.json")); r = d["roofline"]
    print(round(d["value"], 1), round(d["ms_per_step"], 2), d["config"].get("sigma_I"), r["bound"], round(r["frac"], 3), r["end_to_end"]["time_in_mfma_kernels"], d["config"]["loss"])
    for k, v in sorted(r["kernels"].items(), key=lambda kv: -kv[1]["time_share"])[:9]:
        print("   ", k, round(v["time_share"], 4), round(v.get("fp32_equivalent_tflops", 0), 1), round(v["avg_launch_ms"], 3))
except Exception as e:
    print("no bench line:", e)
PY
done
timeout 600 python -m pytest tests/test_train_targets_gpu.py -x -q -s -k "five_step" > $O/pytest_traj.log 2>&1; echo "trajectory rc=$?"; grep -E "five-step|^E  .*assert|Error" $O/pytest_traj.log | head -8 | cut -c1-900
timeout 600 python -m pytest tests/test_gemm_gpu.py -x -q -k "wgrad" > $O/pytest_wgrad.log 2>&1; echo "wgrad tests rc=$?"; tail -3 $O/pytest_wgrad.log | cut -c1-250
