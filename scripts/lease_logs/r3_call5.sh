#!/bin/bash
# round 3, GPU call 5: rchain with the side request behind the weight DMA (4-slot side ring); at-scale parity with tie-ray identification
cd /root/repo; O=/root/repo/gpurun_out/r3c5; mkdir -p $O
timeout 300 python -m pytest tests/test_chain_gpu.py tests/test_rmlp_gpu.py -q > $O/pytest_chain.log 2>&1; echo "chain rc=$?"; tail -6 $O/pytest_chain.log
HOLD_X6=1 timeout 200 python scripts/bench_chain.py 2>&1 | grep -v Warning | tee $O/bench_chain.log
timeout 900 python -m pytest tests/test_scale_gpu.py -q -x > $O/pytest_scale.log 2>&1; echo "scale rc=$?"; tail -25 $O/pytest_scale.log
