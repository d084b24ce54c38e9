#!/bin/bash
# round 5, GPU call 25: the sampler predicts the smallest round count of its last 4 calls: the sampler tests, the synchronisation sites of
# a C3 step (were 4.8 per step), and the C3 line three times
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r5c25; mkdir -p $O
timeout 600 python -m pytest tests/test_path_gpu.py -m gpu -x -q -k "sampler or speculative or forward" > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -2 $O/pytest.log | cut -c1-200
timeout 300 python bench.py --mode c3 --steps 20 --warmup 8 --no-cpu-baseline --no-refine --sync-debug $O/c3_sync_sites.txt > $O/c3_sync.json 2> $O/c3.err; cat $O/c3_sync_sites.txt
for i in 1 2 3; do timeout 300 python bench.py --mode c3 --steps 40 --warmup 8 --no-cpu-baseline --no-refine 2>> $O/c3.err | python -c "import json,sys; d=json.load(sys.stdin); print('c3', round(d['ms_per_step'],2), 'ms/step', d['config']['sampler_rounds_mean_over_timed_calls'])"; done
