#!/bin/bash
# round 3, GPU call 34: rgemm with compile-time ring slots and un-hoisted DMA addresses (331 -> 89 spilled scalars)
cd /root/repo; O=/root/repo/gpurun_out/r3c34; mkdir -p $O
timeout 300 python -m pytest tests/test_gemm_gpu.py -q -k "gemm_r6" > $O/pytest.log 2>&1; echo "rc=$?"; tail -15 $O/pytest.log | cut -c1-200
timeout 200 python scripts/bench_rgemm.py 2>&1 | grep -v Warn | tee $O/bench_rgemm.log
timeout 200 python scripts/bench_rgemm.py 125440 2>&1 | grep -v Warn | tee -a $O/bench_rgemm.log
