#!/bin/bash
# round 4, GPU call 20: wgrad_r6 -- is the cost of the LDS-DMA stream HBM latency?  Developer-build variants: 4 = rows from
# two L2-resident steps, 5 = requests four steps ahead instead of three (results CORRECT), 6 = both
cd /root/repo; O=/root/repo/gpurun_out/r4c20; mkdir -p $O
export HOLD_LIB=/root/repo/hold_amd/libholdhip_dev.so
for P in 1638400 131072; do
  for v in 0 5 4 6 0 5; do
    HOLD_WGRAD_ABL=$v timeout 120 python scripts/bench_wgrad_abl.py $P 2>&1 | grep "wgrad P" | tee -a $O/abl.log
  done
done
