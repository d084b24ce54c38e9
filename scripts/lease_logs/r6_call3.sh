#!/bin/bash
# round 6, GPU call 3: the reference module's recorded steps replayed on the HIP path (tests/test_dropin_gpu.py), the overflow
# guard, the wgrad_h3 under-scaling retry, the per-ray beta-search bound and the composite keys against the reference fixtures
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r6c3; mkdir -p $O
timeout 1500 python -m pytest tests/test_dropin_gpu.py tests/test_rmlp_gpu.py tests/test_path_gpu.py tests/test_gemm_gpu.py -m gpu -q -s -k "dropin or h3 or path or wgrad" > $O/pytest_gpu.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/pytest_gpu.log | cut -c1-200
grep -E "three reference steps" $O/pytest_gpu.log | cut -c1-400
if [ $rc -ne 0 ]; then grep -E "^E  |FAILED" $O/pytest_gpu.log | head -60 | cut -c1-300; fi
