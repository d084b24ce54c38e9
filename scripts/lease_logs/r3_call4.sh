#!/bin/bash
# round 3, GPU call 4: first run of the register-resident backward sweeps (csrc/rchain.hip)
cd /root/repo; O=/root/repo/gpurun_out/r3c4; mkdir -p $O
timeout 120 python -m pytest tests/test_chain_gpu.py -x -q -k "r6 and dsp and 130" > $O/pytest_first.log 2>&1; rc=$?; echo "first rc=$rc"; tail -5 $O/pytest_first.log
timeout 300 python -m pytest tests/test_chain_gpu.py tests/test_rmlp_gpu.py -q > $O/pytest_chain.log 2>&1; echo "chain rc=$?"; tail -12 $O/pytest_chain.log
HOLD_X6=1 timeout 200 python scripts/bench_chain.py 2>&1 | grep -v Warning | tee $O/bench_chain.log
timeout 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "gpu suite rc=$?"; tail -8 $O/pytest_gpu.log
timeout 400 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench.json"))
print(d["value"], d["ms_per_step"], d["config"]["sampler_rounds_mean_over_timed_calls"], d["config"]["rays_per_s_at_sigmaI_4"])
for k, v in d["roofline"]["kernels"].items():
    print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a in ("frac", "fp32_equivalent_tflops", "launches", "avg_launch_ms", "time_share")})
print(d["roofline"]["end_to_end"])
PY
