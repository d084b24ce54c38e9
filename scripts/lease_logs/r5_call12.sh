#!/bin/bash
# round 5, GPU call 12: where does the NaN gradient of the sharp-beta compaction test come from (which run, which arithmetic)
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r5c12; mkdir -p $O
for prec in f16x3 f32x6; do
  HOLD_PRECISION=$prec timeout 600 python -m pytest tests/test_compact_gpu.py -x -q -s -k sharp > $O/pytest_compact_$prec.log 2>&1; echo "$prec rc=$?"; grep -E "compaction:|passed|failed|^E  " $O/pytest_compact_$prec.log | head -6 | cut -c1-600
done
