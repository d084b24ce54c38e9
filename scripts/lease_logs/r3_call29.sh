#!/bin/bash
# round 3, GPU call 29: path / route / scale parity and the bench lines with the register-resident single-layer GEMM wired in
cd /root/repo; O=/root/repo/gpurun_out/r3c29; mkdir -p $O
timeout 1200 python -m pytest tests/test_path_gpu.py tests/test_chain_gpu.py tests/test_train_targets_gpu.py tests/test_scale_gpu.py -q -x > $O/pytest_b.log 2>&1; echo "b rc=$?"; tail -4 $O/pytest_b.log | cut -c1-200
timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 300 python bench.py --mode c3 --steps 30 --warmup 8 --no-cpu-baseline --no-refine > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"
python - <<PY
import json
for f in ("bench", "bench_c3"):
    d = json.load(open("$O/" + f + ".json"))
    print(f, round(d["value"], 1), round(d["ms_per_step"], 2), d["config"]["sampler_rounds_mean_over_timed_calls"], d["roofline"]["end_to_end"])
    for k, v in d["roofline"]["kernels"].items():
        print("  ", k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a in ("frac", "fp32_equivalent_tflops", "launches", "avg_launch_ms", "time_share")})
PY
