#!/bin/bash
# round 3, GPU call 21: first run of the register-resident 256 x 256 weight gradient (csrc/wgrad_r6.hip)
cd /root/repo; O=/root/repo/gpurun_out/r3c21; mkdir -p $O
timeout 300 python -m pytest tests/test_gemm_gpu.py -q -k "wgrad" > $O/pytest.log 2>&1; echo "rc=$?"; tail -12 $O/pytest.log
for P in 1605632 125440; do for v in 0 1; do
  echo "== P=$P HOLD_WGRAD_R6=$v"
  HOLD_LIB=/root/repo/hold_amd/libholdhip_dev.so HOLD_WGRAD_R6=$v timeout 200 python scripts/bench_gemm.py $P 2>&1 | grep "^wgrad" | tee -a $O/wgrad.log
done; done
