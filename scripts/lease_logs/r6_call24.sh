#!/bin/bash
# round 6, GPU call 24: pool buffers sized by the uncompacted sample count (call 23: a pool growing with the compacted count re-allocated ~19 buffers
# = 30 GB whenever a call had more live samples than any before): compaction tests, the 10-frame step at 2 048 pixels per frame with / without
# compaction (three processes each, alternating, per-step times of the first), the headline at beta = 0.005 with / without
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r6c24; mkdir -p $O
timeout 900 python -m pytest tests/test_compact_gpu.py -x -q > $O/pytest.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/pytest.log | cut -c1-200
grep -E "compaction:" $O/pytest.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -E "^E  |FAILED|Error" $O/pytest.log | head -30 | cut -c1-300; fi
run() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python -c "
import json
try:
    d = json.load(open('$O/bench_$name.json')); print('$name', round(d['ms_per_step'], 2), 'ms/step', round(d['value'], 1), 'rays/s', d['config'].get('sigma_I'), d['config']['sample_compaction'].get('live_samples_last_call'))
except Exception as e: print('$name no line', e)
"; }
for i in 1 2 3; do
  st=""; [ $i = 1 ] && st="--step-times $O/steps_compact.txt"
  run px2048_b005_compact_$i --mode c3 --no-refine --beta 0.005 --c3-pixels 2048 --steps 12 --warmup 6 $st
  run px2048_b005_nocompact_$i --mode c3 --no-refine --beta 0.005 --c3-pixels 2048 --steps 12 --warmup 6 --no-compact
done
awk '{print $1, $2, $3, $4, $6, $7, $8, $18, $19, $20}' $O/steps_compact.txt
run b005 --beta 0.005 --steps 4 --warmup 2 --step-times $O/steps_b005.txt
run b005_nocompact --beta 0.005 --no-compact --steps 4 --warmup 2
awk '{print $1, $2, $3, $4, $6, $7, $8, $18, $19, $20}' $O/steps_b005.txt | head -6
