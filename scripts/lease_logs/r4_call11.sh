#!/bin/bash
# round 4, GPU call 11: what HBM latency / stores cost the register-resident sweeps (developer-build ablations; results garbage)
cd /root/repo; O=/root/repo/gpurun_out/r4c11; mkdir -p $O
export HOLD_LIB=/root/repo/hold_amd/libholdhip_dev.so
for v in 0 1 2; do
  echo "== HOLD_R6_ABL=$v (1: side tiles from L2-resident rows, 2: + no result stores)"
  HOLD_R6_ABL=$v HOLD_X6=1 timeout 200 python scripts/bench_chain.py 2>&1 | grep -E "r6" | tee -a $O/abl.log
done
