#!/bin/bash
# round 5, GPU call 11: compaction with padded row counts and the live-fraction gate: tests, bench --beta 0.005 A/B
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r5c11; mkdir -p $O
timeout 600 python -m pytest tests/test_compact_gpu.py -x -q -s > $O/pytest_compact.log 2>&1; echo "compact tests rc=$?"; grep -E "compaction:|passed|failed|^E  " $O/pytest_compact.log | head -12 | cut -c1-400
for c in "" "--no-compact"; do
  timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --beta 0.005 $c > $O/bench_beta$c.json 2> $O/bench_beta$c.err; echo "bench --beta 0.005 $c rc=$?"
  python - <<PY
import json
try:
    d = json.load(open("$O/bench_beta$c.json")); r = d["roofline"]
    print(round(d["value"], 1), round(d["ms_per_step"], 2), d["config"].get("sigma_I"), d["config"]["loss"], d["config"]["sample_compaction"]["live_samples_last_call"])
    for k, v in sorted(r["kernels"].items(), key=lambda kv: -kv[1]["time_share"])[:6]:
        print("   ", k, round(v["time_share"] * d["ms_per_step"], 1), "ms", v["launches"], round(v["avg_launch_ms"], 3))
except Exception as e:
    print("no line", e)
PY
done
