#!/bin/bash
# round 3, GPU call 8: MFMA order pinned (accumulators alternate); r6 kernels re-timed; the re-toleranced tests
cd /root/repo; O=/root/repo/gpurun_out/r3c8; mkdir -p $O
timeout 200 python scripts/bench_rmlp.py 1605632 2>&1 | grep -v Warning | tee $O/bench_rmlp.log
HOLD_X6=1 timeout 200 python scripts/bench_chain.py 2>&1 | grep -v Warning | grep "r6\|DSP" | tee $O/bench_chain.log
timeout 900 python -m pytest tests/test_rmlp_gpu.py tests/test_chain_gpu.py tests/test_parallel_gpu.py -q > $O/pytest_a.log 2>&1; echo "a rc=$?"; tail -5 $O/pytest_a.log
timeout 900 python -m pytest tests/test_scale_gpu.py -q -k "bench_chunk" > $O/pytest_scale.log 2>&1; echo "scale rc=$?"; tail -5 $O/pytest_scale.log
timeout 400 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --two-hands --chunk 16384 > $O/bench_twohands.json 2> $O/bench_twohands.err; echo "twohands rc=$?"; tail -2 $O/bench_twohands.err
python - <<PY
import json
for f in ("bench.json", "bench_twohands.json"):
    try:
        d = json.load(open("$O/" + f))
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, d["value"], d["ms_per_step"], d["config"]["sampler_rounds_mean_over_timed_calls"], d["config"]["rays_per_s_at_sigmaI_4"])
    for k, v in d["roofline"]["kernels"].items():
        print("  ", k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a in ("frac", "achieved", "unit", "fp32_equivalent_tflops", "launches", "avg_launch_ms", "time_share")})
    print("  ", d["roofline"]["end_to_end"])
PY
