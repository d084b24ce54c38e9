#!/bin/bash
# round 6, GPU call 2: the overflow guard (rmlp_h3 + conditional f32x6 launches), the wgrad_h3 under-scaling retry, the per-ray
# beta-search bound and the composite keys against the reference fixtures; then the headline line (cost of the guard launches)
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r6c2; mkdir -p $O
timeout 1500 python -m pytest tests/test_rmlp_gpu.py tests/test_path_gpu.py tests/test_gemm_gpu.py -m gpu -q -x -k "h3 or path or wgrad" > $O/pytest_gpu.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/pytest_gpu.log | cut -c1-200
if [ $rc -ne 0 ]; then grep -E "Error|assert|error|FAILED" $O/pytest_gpu.log | head -30 | cut -c1-300; fi
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(round(d['value'],1), round(d['ms_per_step'],2))"
HOLD_DEV= timeout 200 python scripts/bench_rmlp.py > $O/bench_rmlp.log 2>&1; tail -8 $O/bench_rmlp.log
