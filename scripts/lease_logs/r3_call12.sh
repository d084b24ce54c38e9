#!/bin/bash
# round 3, GPU call 12: developer-build timing ablations of the register-resident kernels' global-memory access pattern
# (32-byte row fragments vs lane-linear 1 KiB pieces; results of the ablated launches are garbage, only the time counts)
cd /root/repo; O=/root/repo/gpurun_out/r3c12; mkdir -p $O
export HOLD_LIB=/root/repo/hold_amd/libholdhip_dev.so
for v in 0 2 3; do
  echo "== trunk_r6 store ablation HOLD_R6_DMA=$v"
  HOLD_R6_DMA=$v timeout 200 python scripts/bench_rmlp.py 1605632 2>&1 | grep "fwd trunk" | tee -a $O/abl.log
done
for v in 0 2 3 4 5; do
  echo "== rchain ablation HOLD_R6_ABL=$v"
  HOLD_R6_ABL=$v HOLD_X6=1 timeout 200 python scripts/bench_chain.py 2>&1 | grep "r6" | tee -a $O/abl.log
done
