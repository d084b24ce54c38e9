#!/bin/bash
# round 6, GPU call 22: final form of the batch-of-frames compaction (mask kernel, rank arithmetic, attempt back-off, size gate): tests; C3 at the
# default beta and at 0.005 (gated off at 125 k samples: must equal --no-compact); the 10-frame step at 2 048 pixels per frame with / without;
# the headline at beta = 0.005 with / without (single-frame path, unchanged); chunk-size check of the headline
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r6c22; mkdir -p $O
timeout 900 python -m pytest tests/test_compact_gpu.py tests/test_scale_gpu.py -x -q > $O/pytest.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/pytest.log | cut -c1-200
grep -E "compaction:" $O/pytest.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -E "^E  |FAILED|Error" $O/pytest.log | head -30 | cut -c1-300; fi
run() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python -c "
import json
try:
    d = json.load(open('$O/bench_$name.json')); print('$name', round(d['ms_per_step'], 2), 'ms/step', round(d['value'], 1), 'rays/s', d['config'].get('sigma_I'), d['config']['sample_compaction'].get('live_samples_last_call'))
except Exception as e: print('$name no line', e)
"; }
run c3_default --mode c3 --no-refine --steps 30 --warmup 5
run c3_b005 --mode c3 --no-refine --steps 30 --warmup 5 --beta 0.005
run c3_b005_nocompact --mode c3 --no-refine --steps 30 --warmup 5 --beta 0.005 --no-compact
run c3_default_2 --mode c3 --no-refine --steps 30 --warmup 5
for i in 1 2; do
  run px2048_b005_compact_$i --mode c3 --no-refine --beta 0.005 --c3-pixels 2048 --steps 12 --warmup 6
  run px2048_b005_nocompact_$i --mode c3 --no-refine --beta 0.005 --c3-pixels 2048 --steps 12 --warmup 6 --no-compact
done
run b005 --beta 0.005 --steps 4 --warmup 2
run b005_nocompact --beta 0.005 --no-compact --steps 4 --warmup 2
run headline --steps 4 --warmup 2
run headline_chunk32k --steps 4 --warmup 2 --chunk 32768
