#!/bin/bash
# round 5, GPU call 9: h3 trunk kernels with the fragment reads issued 5-9 MFMAs ahead (tests, A/B timing), trajectory test, bench
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r5c9; mkdir -p $O
timeout 600 python -m pytest tests/test_rmlp_gpu.py -x -q > $O/pytest_rmlp.log 2>&1; echo "rmlp tests rc=$?"; tail -3 $O/pytest_rmlp.log | cut -c1-250
timeout 600 python scripts/bench_rmlp.py > $O/bench_rmlp.log 2>&1; echo "bench_rmlp rc=$?"; grep -E "h3|trunk_h3" $O/bench_rmlp.log | cut -c1-250
timeout 600 python -m pytest tests/test_train_targets_gpu.py -x -q -s -k "five_step" > $O/pytest_traj.log 2>&1; echo "trajectory rc=$?"; grep -E "five-step|^E  .*assert|Error" $O/pytest_traj.log | head -8 | cut -c1-1500
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench.json")); r = d["roofline"]
print(round(d["value"], 1), round(d["ms_per_step"], 2), d["config"].get("sigma_I"), r["bound"], round(r["frac"], 3), r["end_to_end"]["time_in_mfma_kernels"], d["config"]["loss"])
for k, v in sorted(r["kernels"].items(), key=lambda kv: -kv[1]["time_share"])[:9]:
    print("   ", k, round(v["time_share"], 4), round(v.get("fp32_equivalent_tflops", 0), 1), round(v["avg_launch_ms"], 3))
PY
