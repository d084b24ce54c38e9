#!/bin/bash
# round 6, GPU call 20 (19 = the same with a first version whose mask scatter and frame sort cost a synchronisation and 440 ms at 2 M points): compaction for batches of frames (field._compaction) and the betas' host copies queued behind the optimiser step
# (LaplaceDensity.prefetch_host): tests; then the reference's 1 280-ray step (C3) A/B with / without the prefetch, alternating, same box;
# C3 at a trained model's beta = 0.005 with / without compaction at 128 and 2 048 pixels per frame; synchronisation sites of a C3 step
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r6c20; mkdir -p $O
timeout 900 python -m pytest tests/test_compact_gpu.py tests/test_dropin_gpu.py tests/test_train_targets_gpu.py -x -q > $O/pytest.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/pytest.log | cut -c1-200
grep -E "compaction:" $O/pytest.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -E "^E  |FAILED|Error" $O/pytest.log | head -30 | cut -c1-300; fi
c3() { name=$1; shift; envs=(); while [[ "$1" == *=* ]]; do envs+=("$1"); shift; done; env "${envs[@]}" timeout 300 python bench.py --mode c3 --steps 30 --warmup 5 --no-cpu-baseline --no-refine "$@" > $O/c3_$name.json 2> $O/c3_$name.err; python -c "
import json
try:
    d = json.load(open('$O/c3_$name.json')); print('$name', round(d['ms_per_step'], 2), 'ms/step', round(d['value'], 1), 'rays/s', d['config'].get('sigma_I'), d['config']['sample_compaction'].get('live_samples_last_call'))
except Exception as e: print('$name no line', e)
"; }
for i in 1 2 3; do
  c3 prefetch_$i X=1
  c3 direct_$i HOLD_BETA_PREFETCH=0
done
c3 b005_compact X=1 --beta 0.005
c3 b005_nocompact X=1 --beta 0.005 --no-compact
c3 b005_compact_2 X=1 --beta 0.005
c3 b005_nocompact_2 X=1 --beta 0.005 --no-compact
c3 b005_px2048_compact X=1 --beta 0.005 --c3-pixels 2048 --steps 10 --warmup 3
c3 b005_px2048_nocompact X=1 --beta 0.005 --c3-pixels 2048 --no-compact --steps 10 --warmup 3
c3 px2048 X=1 --c3-pixels 2048 --steps 10 --warmup 3
timeout 300 python bench.py --mode c3 --steps 8 --warmup 4 --no-cpu-baseline --no-refine --sync-debug $O/c3_sync_sites.txt > $O/c3_sync.json 2> $O/c3_sync.err; tail -25 $O/c3_sync_sites.txt | cut -c1-200
