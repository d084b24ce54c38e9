#!/bin/bash
# round 3, GPU call 3: store-hazard fix of trunk_r6, full GPU suite with the register-resident trunk wired into the path, bench line
cd /root/repo; O=/root/repo/gpurun_out/r3c3; mkdir -p $O
timeout 300 python -m pytest tests/test_rmlp_gpu.py -q > $O/pytest_rmlp.log 2>&1; echo "rmlp rc=$?"; tail -4 $O/pytest_rmlp.log
timeout 100 python scripts/bench_rmlp.py 1605632 2>&1 | grep -v Warning | tee $O/bench_rmlp.log
timeout 900 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "gpu suite rc=$?"; tail -6 $O/pytest_gpu.log
timeout 400 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-300 $O/bench.json; tail -3 $O/bench.err
python - <<PY
import json
d = json.load(open("$O/bench.json"))
print(d["value"], d["ms_per_step"], d["config"]["sampler_rounds_mean_over_timed_calls"], d["config"]["flop_per_ray"], d["config"]["rays_per_s_at_sigmaI_4"])
for k, v in d["roofline"]["kernels"].items():
    print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a in ("achieved", "frac", "fp32_equivalent_tflops", "launches", "avg_launch_ms", "time_share")})
print(d["roofline"]["end_to_end"])
PY
