#!/bin/bash
# round 4, GPU call 23: 16-byte seed_dsp path: kernel test, end-to-end tests, bench line
cd /root/repo; O=/root/repo/gpurun_out/r4c23; mkdir -p $O
timeout 900 python -m pytest tests/test_points_gpu.py tests/test_path_gpu.py tests/test_scale_gpu.py -q -x > $O/pytest_sel.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/pytest_sel.log | cut -c1-220
if [ $rc -ne 0 ]; then grep -E "Error|assert|error" $O/pytest_sel.log | head -20 | cut -c1-220; exit 0; fi
timeout 400 python bench.py --no-cpu-baseline --no-refine > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench.json"))
print("rays/s", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 1), d["roofline"]["end_to_end"]["time_in_mfma_kernels"])
PY
