#!/bin/bash
# round 6, GPU call 8: the f16x3 sweeps with the weight groups requested 3 rendezvous ahead (DIST 3, 64 KiB ring) against DIST 1, and the
# two ablations (side tiles from L2-resident rows; additionally no result stores) -- developer build, chain micro-benchmark
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r6c8; mkdir -p $O
export HOLD_LIB=/root/repo/hold_amd/libholdhip_dev.so HOLD_X6=1
for cfg in "DIST=1" "DIST=3" "DIST=1 ABL=1" "DIST=1 ABL=2" "DIST=3 ABL=2"; do
  d=$(echo $cfg | sed -n 's/.*DIST=\([0-9]\).*/\1/p'); a=$(echo $cfg | sed -n 's/.*ABL=\([0-9]\).*/\1/p')
  echo "== $cfg"
  HOLD_H3C_DIST=$d HOLD_H3C_ABL=${a:-0} timeout 200 python scripts/bench_chain.py 2>&1 | grep -E "^h3|fallbacks" | cut -c1-200
done | tee $O/dist_abl.log
HOLD_H3C_DIST=3 timeout 600 python -m pytest tests/test_chain_gpu.py -m gpu -q -x -k "h3" > $O/pytest_dist3.log 2>&1; echo "chain tests (DIST 3) rc=$?"; tail -2 $O/pytest_dist3.log | cut -c1-200
