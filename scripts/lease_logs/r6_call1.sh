#!/bin/bash
# round 6, GPU call 1: starting tree (= round 5's final) -- full GPU suite, then the same-box A/B of the headline line with and
# without the per-launch HIP event pairs, twice each, interleaved
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r6c1; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "gpu suite rc=$rc"; tail -3 $O/pytest_gpu.log | cut -c1-200
line() { python -c "
import json
try:
    d = json.load(open('$1')); print('$1'.split('/')[-1], round(d['value'], 1), round(d['ms_per_step'], 2))
except Exception as e: print('no line', e)
"; }
for i in 1 2; do
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_profile_$i.json 2> $O/bench_profile_$i.err; line $O/bench_profile_$i.json
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile > $O/bench_noprofile_$i.json 2> $O/bench_noprofile_$i.err; line $O/bench_noprofile_$i.json
done
