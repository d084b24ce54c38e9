#!/bin/bash
# round 4, GPU call 33: the background's lin8 as N = 256 GEMM + row dot for the sdf column: end-to-end tests, bench line
cd /root/repo; O=/root/repo/gpurun_out/r4c33; mkdir -p $O
timeout 600 python -m pytest tests/test_path_gpu.py -q -x > $O/pytest_path.log 2>&1; rc=$?; echo "path tests rc=$rc"; tail -3 $O/pytest_path.log | cut -c1-220
if [ $rc -ne 0 ]; then grep -E "Error|assert|error" $O/pytest_path.log | head -20 | cut -c1-220; exit 0; fi
timeout 300 python bench.py --no-cpu-baseline --no-refine > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench.json")); k = d["roofline"]["kernels"]["gemm_nt_kernel"]
print("rays/s", round(d["value"], 1), "gemm_nt share", round(k["time_share"], 4), "avg ms", round(k["avg_launch_ms"], 3), "launches", k["launches"])
PY
