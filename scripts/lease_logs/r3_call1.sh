#!/bin/bash
# round 3, GPU call 1: first run of the register-resident trunk (csrc/rmlp.hip): parity, timing, baseline bench line
cd /root/repo; O=gpurun_out/r3c1; mkdir -p $O
timeout 120 python -m pytest tests/test_rmlp_gpu.py -x -q -k "fused_sdf_r6 and 33-False" > $O/pytest_first.log 2>&1; rc=$?; echo "first rc=$rc"; tail -5 $O/pytest_first.log
if [ $rc -ne 0 ]; then
  echo "--- product DMA form failed; trying the per-piece form of the developer build"
  HOLD_LIB=hold_amd/libholdhip_dev.so HOLD_R6_DMA=1 timeout 120 python -m pytest tests/test_rmlp_gpu.py -x -q -k "fused_sdf_r6" > $O/pytest_dma1.log 2>&1; echo "dma1 rc=$?"; tail -15 $O/pytest_dma1.log
else
  timeout 300 python -m pytest tests/test_rmlp_gpu.py -x -q > $O/pytest_rmlp.log 2>&1; echo "rmlp rc=$?"; tail -15 $O/pytest_rmlp.log
  timeout 200 python scripts/bench_rmlp.py > $O/bench_rmlp.log 2>&1; cat $O/bench_rmlp.log | grep -v Warning
fi
timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_base.json 2> $O/bench_base.err; echo "bench rc=$?"; cut -c1-400 $O/bench_base.json
