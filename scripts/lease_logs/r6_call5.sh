#!/bin/bash
# round 6, GPU call 5: first hardware run of the f16x3 sweeps (csrc/rchain_h3.hip): chain tests, the gate, the fallback, micro-benchmark
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r6c5; mkdir -p $O
timeout 900 python -m pytest tests/test_chain_gpu.py -m gpu -q -s -x > $O/pytest_chain.log 2>&1; rc=$?; echo "chain tests rc=$rc"; tail -3 $O/pytest_chain.log | cut -c1-200
grep -E "family" $O/pytest_chain.log | cut -c1-200
if [ $rc -ne 0 ]; then grep -E "^E  |FAILED|Error" $O/pytest_chain.log | head -40 | cut -c1-300; fi
HOLD_X6=1 timeout 300 python scripts/bench_chain.py > $O/bench_chain.log 2>&1; tail -15 $O/bench_chain.log | cut -c1-250
