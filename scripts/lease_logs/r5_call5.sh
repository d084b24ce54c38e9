#!/bin/bash
# round 5, GPU call 5: (1) localise the memory access fault of call 4's bench run with the A/B switches; (2) developer-build
# ablation: what the transcendentals of the sampler query's k step cost
cd /root/repo; O=/root/repo/gpurun_out/r5c5; mkdir -p $O
for cfg in "0 0" "1 0" "0 1" "1 1"; do
  set -- $cfg
  HOLD_H3_TRUNK=$1 HOLD_H3_WGRAD=$2 timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile > $O/bench_t$1_w$2.json 2> $O/bench_t$1_w$2.err
  echo "H3_TRUNK=$1 H3_WGRAD=$2 rc=$? $(cut -c1-60 $O/bench_t$1_w$2.json) $(grep -c 'Memory access fault' $O/bench_t$1_w$2.err)"
done
for a in 0 1 2 3; do
  HOLD_LIB=/root/repo/hold_amd/libholdhip_dev.so HOLD_H3_ABL=$a timeout 120 python scripts/bench_h3_abl.py 2>&1 | tail -1
done
timeout 600 python -m pytest tests/test_train_targets_gpu.py -x -q -s -k "five_step" > $O/pytest_traj.log 2>&1; echo "trajectory rc=$?"; grep -E "five-step|assert|Error" $O/pytest_traj.log | head -8 | cut -c1-600
timeout 300 python -m pytest tests/test_parallel_gpu.py -x -q > $O/pytest_par.log 2>&1; echo "parallel rc=$?"; tail -3 $O/pytest_par.log | cut -c1-300
