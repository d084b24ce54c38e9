#!/bin/bash
# round 4, GPU call 1: full-line side I/O for the register-resident descending sweeps (rtile_kernel): parity of the
# r6 chain tests on the product build, then same-process-family A/B against the round-3 row-fragment kernel (dev build)
cd /root/repo; O=/root/repo/gpurun_out/r4c1; mkdir -p $O
timeout 300 python -m pytest tests/test_chain_gpu.py -q -x -k "descending" > $O/pytest_chain.log 2>&1; echo "chain tests rc=$?"; tail -5 $O/pytest_chain.log | cut -c1-200
export HOLD_LIB=/root/repo/hold_amd/libholdhip_dev.so
for v in frag tile frag tile; do
  echo "== HOLD_R6_IO=$v"
  HOLD_R6_IO=$v HOLD_X6=1 timeout 200 python scripts/bench_chain.py 2>&1 | grep -E "DSP|DBWD" | tee -a $O/ab.log
done
