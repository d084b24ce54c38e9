#!/bin/bash
# round 6, GPU call 25: full GPU suite on the tree with the batch-of-frames compaction + smoke + the default bench line (no CPU leg) + C3
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r6c25; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "gpu suite rc=$rc"; tail -3 $O/pytest_gpu.log | cut -c1-200
if [ $rc -ne 0 ]; then grep -E "^E  |FAILED" $O/pytest_gpu.log | head -30 | cut -c1-300; fi
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log | cut -c1-200
run() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python -c "
import json
try:
    d = json.load(open('$O/bench_$name.json')); print('$name', round(d['ms_per_step'], 2), 'ms/step', round(d['value'], 1), 'rays/s', d['config'].get('sigma_I'), d['config']['sample_compaction'].get('live_samples_last_call'), d['config'].get('f16x3_launches_recomputed_in_f32x6_per_step'))
except Exception as e: print('$name no line', e)
"; }
run headline --steps 4 --warmup 2
run c3 --mode c3 --steps 30 --warmup 5
run render --mode render --steps 10 --warmup 3
run twohands --two-hands --chunk 16384 --steps 4 --warmup 2
