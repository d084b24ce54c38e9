#!/bin/bash
# round 5, GPU call 26: smoke() and the default bench line (GPU part) on the tree with the sampler's round predictor
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r5c26; mkdir -p $O
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log | cut -c1-200
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$O/bench.json')); r=d['roofline']; print(round(d['value'],1), round(d['ms_per_step'],2), r['bound'], round(r['frac'],3), d['config'].get('sampler_rounds_mean_over_timed_calls'))"
