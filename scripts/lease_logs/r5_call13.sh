#!/bin/bash
# round 5, GPU call 13: wgrad_h3 with exact running maxima + one retry: gemm tests, timing, compaction test in both arithmetics
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r5c13; mkdir -p $O
timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q -k "wgrad" > $O/pytest_wgrad.log 2>&1; echo "wgrad tests rc=$?"; tail -4 $O/pytest_wgrad.log | cut -c1-300; grep -E "^E  " $O/pytest_wgrad.log | head -5 | cut -c1-300
timeout 300 python scripts/bench_wgrad_h3.py > $O/bench_wgrad_h3.log 2>&1; echo "bench_wgrad rc=$?"; cat $O/bench_wgrad_h3.log | cut -c1-250
for prec in f16x3 f32x6; do
  HOLD_PRECISION=$prec timeout 600 python -m pytest tests/test_compact_gpu.py -x -q -s > $O/pytest_compact_$prec.log 2>&1; echo "compact $prec rc=$?"; grep -E "compaction:|passed|failed|^E  " $O/pytest_compact_$prec.log | head -6 | cut -c1-600
done
timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --beta 0.005 > $O/bench_beta.json 2> $O/bench_beta.err; echo "bench --beta 0.005 rc=$?"
python -c "
import json; d = json.load(open('$O/bench_beta.json')); print(round(d['value'],1), d['config']['loss'], d['config']['sample_compaction']['live_samples_last_call'])"
