#!/bin/bash
# round 4, GPU call 19: which of its three streams bounds wgrad_r6 (developer-build ablations: no limb preparation / no
# LDS-DMA / no MFMAs; results garbage, durations only), at the headline chunk size and at the C3 chunk size
cd /root/repo; O=/root/repo/gpurun_out/r4c19; mkdir -p $O
export HOLD_LIB=/root/repo/hold_amd/libholdhip_dev.so
for P in 1638400 131072; do
  for v in 0 1 2 3; do
    HOLD_WGRAD_ABL=$v timeout 120 python scripts/bench_wgrad_abl.py $P 2>&1 | grep "wgrad P" | tee -a $O/abl.log
  done
done
