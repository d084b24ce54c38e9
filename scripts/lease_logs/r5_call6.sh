#!/bin/bash
# round 5, GPU call 6: the memory access fault of calls 4 / 5 under HIP_LAUNCH_BLOCKING + faulthandler (which launch?), no core
# dumps (they filled the box's disk in call 5); then the transcendental ablation and the new tests that call 5 did not reach
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r5c6; mkdir -p $O
HOLD_H3_TRUNK=0 HOLD_H3_WGRAD=0 HIP_LAUNCH_BLOCKING=1 AMD_SERIALIZE_KERNEL=3 timeout 200 python -X faulthandler bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile > $O/bench_blocking.json 2> $O/bench_blocking.err
echo "blocking bench rc=$?"; grep -v "Warning\|warn\|amdgpu.ids" $O/bench_blocking.err | tail -40 | cut -c1-200
for a in 0 1 2 3; do
  HOLD_LIB=/root/repo/hold_amd/libholdhip_dev.so HOLD_H3_ABL=$a timeout 120 python scripts/bench_h3_abl.py 2>&1 | tail -1
done
timeout 600 python -m pytest tests/test_train_targets_gpu.py -x -q -s -k "five_step" > $O/pytest_traj.log 2>&1; echo "trajectory rc=$?"; grep -E "five-step|assert|Error" $O/pytest_traj.log | head -8 | cut -c1-700
timeout 300 python -m pytest tests/test_parallel_gpu.py -x -q > $O/pytest_par.log 2>&1; echo "parallel rc=$?"; tail -3 $O/pytest_par.log | cut -c1-300
