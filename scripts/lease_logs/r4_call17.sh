#!/bin/bash
# round 4, GPU call 17: the 256 x 256 weight gradients of a backward as ONE grouped launch (hold_wgrad_group_x6) and the
# frame column sums with a block-level reduction: kernel tests, end-to-end gradient tests, then headline and C3 bench lines
# with the grouped launch on and off
cd /root/repo; O=/root/repo/gpurun_out/r4c17; mkdir -p $O
timeout 400 python -m pytest tests/test_gemm_gpu.py -q -x -k "wgrad" > $O/pytest_wgrad.log 2>&1; rc=$?; echo "wgrad tests rc=$rc"; tail -4 $O/pytest_wgrad.log | cut -c1-220
if [ $rc -ne 0 ]; then grep -E "Error|assert|error" $O/pytest_wgrad.log | head -20 | cut -c1-220; exit 0; fi
timeout 700 python -m pytest tests/test_path_gpu.py tests/test_train_targets_gpu.py tests/test_points_gpu.py -q -x > $O/pytest_sel.log 2>&1; rc=$?; echo "path tests rc=$rc"; tail -4 $O/pytest_sel.log | cut -c1-220
if [ $rc -ne 0 ]; then grep -E "Error|assert|error" $O/pytest_sel.log | head -20 | cut -c1-220; exit 0; fi
for g in 1 0; do
  HOLD_WGRAD_GROUP=$g timeout 400 python bench.py --no-cpu-baseline --no-refine > $O/bench_g$g.json 2> $O/bench_g$g.err; echo "bench group=$g rc=$?"
  HOLD_WGRAD_GROUP=$g timeout 200 python bench.py --mode c3 --steps 40 --warmup 10 --no-cpu-baseline --no-refine > $O/c3_g$g.json 2> $O/c3_g$g.err; echo "c3 group=$g rc=$?"
done
python - <<PY
import json
for g in (1, 0):
    d = json.load(open("$O/bench_g%d.json" % g)); c = json.load(open("$O/c3_g%d.json" % g))
    print("group", g, "rays/s", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 1), "| c3 ms/step", round(c["ms_per_step"], 2))
    for k, v in d["roofline"]["kernels"].items():
        if k.startswith("wgrad") or k.startswith("frame"):
            print(f"  {k:22s} share {v['time_share']:.3f} TF-eq {v.get('fp32_equivalent_tflops', 0):.1f} avg_ms {v['avg_launch_ms']:.3f} launches {v['launches']}")
PY
