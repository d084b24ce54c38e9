#!/bin/bash
# round 4, GPU call 18: d sdf / d embedding kept in place in t_3's skip columns (no 39-column copies: forward ge, backward
# gebar via hold_embed_bwd2's second output, ebar as a view of r_3): kernel test, end-to-end gradient tests, bench
cd /root/repo; O=/root/repo/gpurun_out/r4c18; mkdir -p $O
timeout 300 python -m pytest tests/test_points_gpu.py -q -x > $O/pytest_points.log 2>&1; rc=$?; echo "points tests rc=$rc"; tail -4 $O/pytest_points.log | cut -c1-220
if [ $rc -ne 0 ]; then grep -E "Error|assert|error" $O/pytest_points.log | head -20 | cut -c1-220; exit 0; fi
timeout 900 python -m pytest tests/test_path_gpu.py tests/test_train_targets_gpu.py tests/test_chain_gpu.py tests/test_scale_gpu.py -q -x > $O/pytest_sel.log 2>&1; rc=$?; echo "path tests rc=$rc"; tail -4 $O/pytest_sel.log | cut -c1-220
if [ $rc -ne 0 ]; then grep -E "Error|assert|error" $O/pytest_sel.log | head -20 | cut -c1-220; exit 0; fi
timeout 400 python bench.py --no-cpu-baseline --no-refine > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 200 python bench.py --mode c3 --steps 40 --warmup 10 --no-cpu-baseline --no-refine > $O/c3.json 2> $O/c3.err; echo "c3 rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench.json")); c = json.load(open("$O/c3.json"))
print("rays/s", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 1), "| c3 ms/step", round(c["ms_per_step"], 2))
for k, v in d["roofline"]["kernels"].items():
    print(f"  {k:22s} share {v['time_share']:.3f} TF-eq {v.get('fp32_equivalent_tflops', 0):.1f} avg_ms {v['avg_launch_ms']:.3f} launches {v['launches']}")
print(d["roofline"]["end_to_end"])
PY
