#!/bin/bash
# round 3, GPU call 32: full GPU suite in both arithmetics + smoke on the final tree (wgrad_r6, rgemm)
cd /root/repo; O=/root/repo/gpurun_out/r3c32; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu > $O/pytest_gpu_x6.log 2>&1; echo "x6 suite rc=$?"; tail -4 $O/pytest_gpu_x6.log
HOLD_PRECISION=f32 timeout 1200 python -m pytest tests -q -m gpu > $O/pytest_gpu_f32.log 2>&1; echo "f32 suite rc=$?"; tail -4 $O/pytest_gpu_f32.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
