#!/bin/bash
# round 3, GPU call 19: XCD-aware (tile, split) numbering of the wgrad workgroups, A/B in the developer build
cd /root/repo; O=/root/repo/gpurun_out/r3c19; mkdir -p $O
for P in 1605632 125440; do for v in 0 1; do
  echo "== P=$P HOLD_WGRAD_REMAP=$v"
  HOLD_LIB=/root/repo/hold_amd/libholdhip_dev.so HOLD_WGRAD_REMAP=$v timeout 200 python scripts/bench_gemm.py $P 2>&1 | grep "^wgrad" | tee -a $O/wgrad.log
done; done
timeout 600 python -m pytest tests/test_gemm_gpu.py -q -k "wgrad" > $O/pytest.log 2>&1; echo "rc=$?"; tail -3 $O/pytest.log
