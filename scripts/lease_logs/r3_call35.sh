#!/bin/bash
# round 3, GPU call 35: evidence for profiles/ on the final tree -- rocprofv3 kernel stats + PMC traffic of the default bench command,
# the bench lines of every mode on this tree
cd /root/repo; O=/root/repo/gpurun_out/r3c35; mkdir -p $O
bash scripts/prof_r03.sh > $O/prof.log 2>&1; echo "prof rc=$?"; tail -12 $O/prof.log | cut -c1-240
timeout 600 python bench.py > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc=$?"
timeout 300 python bench.py --mode render --no-cpu-baseline > $O/bench_render.json 2> $O/bench_render.err; echo "render rc=$?"
timeout 400 python bench.py --two-hands --chunk 16384 --no-cpu-baseline > $O/bench_twohands.json 2> $O/bench_twohands.err; echo "twohands rc=$?"
timeout 300 python bench.py --mode c3 --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"
timeout 300 python bench.py --mode c5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?"
timeout 400 python bench.py --fp32-mfma --no-cpu-baseline > $O/bench_fp32_mfma.json 2> $O/bench_fp32.err; echo "fp32 rc=$?"
python - <<PY
import json
for f in ("bench_final", "bench_render", "bench_twohands", "bench_c3", "bench_c5", "bench_fp32_mfma"):
    try:
        d = json.load(open("$O/" + f + ".json"))
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, round(d["value"], 1), round(d["ms_per_step"], 2), d["config"].get("sampler_rounds_mean_over_timed_calls"), d["roofline"]["frac"], d.get("cpu_baseline"))
PY
timeout 900 python -m pytest tests -q -m gpu > $O/pytest_gpu_x6.log 2>&1; echo "x6 suite rc=$?"; tail -3 $O/pytest_gpu_x6.log
