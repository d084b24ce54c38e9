#!/bin/bash
# round 4, GPU call 30: hold_gemm_narrow_x6 (csrc/rnarrow.hip) -- kernel tests, end-to-end tests, same-box A/B of the bench line
cd /root/repo; O=/root/repo/gpurun_out/r4c30; mkdir -p $O
timeout 300 python -m pytest tests/test_gemm_gpu.py -q -x -k "narrow" > $O/pytest_narrow.log 2>&1; rc=$?; echo "narrow tests rc=$rc"; tail -3 $O/pytest_narrow.log | cut -c1-220
if [ $rc -ne 0 ]; then grep -E "Error|assert|error" $O/pytest_narrow.log | head -20 | cut -c1-220; exit 0; fi
timeout 600 python -m pytest tests/test_path_gpu.py -q -x > $O/pytest_path.log 2>&1; rc=$?; echo "path tests rc=$rc"; tail -3 $O/pytest_path.log | cut -c1-220
if [ $rc -ne 0 ]; then grep -E "Error|assert|error" $O/pytest_path.log | head -20 | cut -c1-220; exit 0; fi
for g in 1 0; do
  HOLD_NARROW=$g timeout 400 python bench.py --no-cpu-baseline --no-refine > $O/bench_n$g.json 2> $O/bench_n$g.err; echo "bench narrow=$g rc=$?"
  python - <<PY
import json
d = json.load(open("$O/bench_n$g.json")); k = d["roofline"]["kernels"]
print("narrow", $g, "rays/s", round(d["value"], 1), {n: (round(v["time_share"], 4), round(v["avg_launch_ms"], 3), v["launches"], round(v.get("fp32_equivalent_tflops", 0), 1)) for n, v in k.items() if n in ("gemm_nt_kernel", "rnarrow_kernel")})
PY
done
