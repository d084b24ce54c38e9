#!/bin/bash
# round 6, GPU call 39 (ABL 8 = whole-line result stores with the product's loads; ABL 5 = L2-hot weights, 6 = L2-hot input rows, 7 = no result stores; 30 = the same WITH the conditional f32x6 relaunch, which the garbage results of an ablation trigger): what the access SHAPE of hold_gemm_h3 costs -- developer build, timing only (results garbage): HOLD_RG_ABL=1 requests the
# input / mask pieces as whole lines (8 rows x 128 B per instruction) instead of 32-byte row fragments (32 rows x 32 B), =2 the result stores too;
# same bytes, same instruction counts, same queue
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r6c39; mkdir -p $O
for rep in 1 2; do
for v in 0 8 7; do
  echo "--- HOLD_RG_ABL=$v (run $rep)"; HOLD_LIB=/root/repo/hold_amd/libholdhip_dev.so HOLD_RG_ABL=$v timeout 300 python scripts/bench_rgemm.py > $O/rgemm_abl${v}_$rep.log 2>&1; grep gemm_h3 $O/rgemm_abl${v}_$rep.log | cut -c1-75
done
done
