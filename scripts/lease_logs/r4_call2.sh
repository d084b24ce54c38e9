#!/bin/bash
# round 4, GPU call 2: unified register-resident sweep kernel (pair-granular weight ring; DSP / DSP+a2 at DIST 2, new DBWD):
# parity of every chain test, then the micro-benchmark against hold_chain_x6
cd /root/repo; O=/root/repo/gpurun_out/r4c2; mkdir -p $O
timeout 400 python -m pytest tests/test_chain_gpu.py -q -x > $O/pytest_chain.log 2>&1; echo "chain tests rc=$?"; tail -8 $O/pytest_chain.log | cut -c1-200
for v in 1 2; do
  HOLD_X6=1 timeout 200 python scripts/bench_chain.py 2>&1 | grep -E "DSP|DBWD" | tee -a $O/ab.log
done
