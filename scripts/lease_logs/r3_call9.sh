#!/bin/bash
# round 3, GPU call 9: DMA pieces spread between the MFMAs (r6 kernels)
cd /root/repo; O=/root/repo/gpurun_out/r3c9; mkdir -p $O
timeout 200 python scripts/bench_rmlp.py 1605632 2>&1 | grep -v Warning | tee $O/bench_rmlp.log
HOLD_X6=1 timeout 200 python scripts/bench_chain.py 2>&1 | grep -v Warning | grep "r6\|DSP" | tee $O/bench_chain.log
timeout 600 python -m pytest tests/test_rmlp_gpu.py tests/test_chain_gpu.py -q > $O/pytest_a.log 2>&1; echo "a rc=$?"; tail -4 $O/pytest_a.log
