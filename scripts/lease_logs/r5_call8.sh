#!/bin/bash
# round 5, GPU call 8: what bounds the h3 sampler query (developer-build ablations: no LDS-DMA / no fragment reads / no barrier),
# trajectory test, the sigma = 1e-6 silhouette test under the conditioning-aware bound
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r5c8; mkdir -p $O
for a in 0 4 8 12 16 28 31; do
  HOLD_LIB=/root/repo/hold_amd/libholdhip_dev.so HOLD_H3_ABL=$a timeout 120 python scripts/bench_h3_abl.py 2>&1 | tail -1
done
timeout 600 python -m pytest tests/test_train_targets_gpu.py -x -q -s -k "five_step" > $O/pytest_traj.log 2>&1; echo "trajectory rc=$?"; grep -E "five-step|^E  .*assert|Error" $O/pytest_traj.log | head -8 | cut -c1-1200
timeout 600 python -m pytest tests/test_fitting_gpu.py -x -q -s > $O/pytest_fit.log 2>&1; echo "fitting rc=$?"; grep -E "sigma 1e-6|^E  |passed|failed" $O/pytest_fit.log | head -8 | cut -c1-600
