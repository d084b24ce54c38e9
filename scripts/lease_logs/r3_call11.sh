#!/bin/bash
# round 3, GPU call 11: full GPU suite + bench lines on the explicit-schedule r6 kernels
cd /root/repo; O=/root/repo/gpurun_out/r3c11; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "gpu suite rc=$?"; tail -6 $O/pytest_gpu.log
timeout 400 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench.json"))
print(d["value"], d["ms_per_step"], d["config"]["sampler_rounds_mean_over_timed_calls"], d["config"]["rays_per_s_at_sigmaI_4"])
for k, v in d["roofline"]["kernels"].items():
    print("  ", k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a in ("frac", "achieved", "unit", "fp32_equivalent_tflops", "launches", "avg_launch_ms", "time_share")})
print("  ", d["roofline"]["end_to_end"])
PY
