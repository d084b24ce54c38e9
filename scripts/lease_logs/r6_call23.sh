#!/bin/bash
# round 6, GPU call 23: the 10-frame step at 2 048 pixels per frame with compaction is bimodal across processes (275 / 359 ms, calls 20-22):
# per-step times and allocator counters of four processes
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r6c23; mkdir -p $O
for i in 1 2 3 4; do
  timeout 400 python bench.py --no-cpu-baseline --mode c3 --no-refine --beta 0.005 --c3-pixels 2048 --steps 8 --warmup 6 --step-times $O/steps_$i.txt > $O/bench_$i.json 2> $O/bench_$i.err
  python -c "
import json
d = json.load(open('$O/bench_$i.json')); print('run $i', round(d['ms_per_step'], 1), d['config']['sample_compaction'].get('live_samples_last_call'))"
  cut -c1-260 $O/steps_$i.txt
done
