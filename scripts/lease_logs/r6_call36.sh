#!/bin/bash
# round 6, GPU call 36: the full GPU suite with the OTHER arithmetic as the package default (HOLD_PRECISION=f32x6: every f16x3 route off), and the
# path / drop-in / compaction tests with the round's A/B switches off one by one -- the routes a fallback or an A/B takes must stay green
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r6c36; mkdir -p $O
HOLD_PRECISION=f32x6 timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_f32x6.log 2>&1; echo "f32x6 suite rc=$?"; tail -2 $O/pytest_f32x6.log | cut -c1-200
for sw in HOLD_RELU_BITS HOLD_H3_GEMM HOLD_H3_BWD HOLD_H3_TRUNK HOLD_H3_WGRAD; do
  env $sw=0 timeout 900 python -m pytest tests/test_path_gpu.py tests/test_dropin_gpu.py tests/test_compact_gpu.py -x -q > $O/pytest_$sw.log 2>&1; echo "$sw=0 rc=$?"; tail -1 $O/pytest_$sw.log | cut -c1-200
done
