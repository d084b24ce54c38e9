#!/bin/bash
# round 5, GPU call 23 (experiment, not a bench line; HOLD_EXPERIMENT_NOSYNC was a LOCAL two-line patch of sampler.py / hold_net.py, reverted):
# predicted round count and the betas are never re-read (valid only because the bench restores the weights every step)
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r5c23; mkdir -p $O
for v in sync nosync sync nosync sync nosync; do
  if [ $v = nosync ]; then export HOLD_EXPERIMENT_NOSYNC=1; else unset HOLD_EXPERIMENT_NOSYNC; fi
  timeout 300 python bench.py --mode c3 --steps 40 --warmup 8 --no-cpu-baseline --no-refine --no-profile 2> $O/err_$v.txt | python -c "import json,sys; d=json.load(sys.stdin); print('$v', round(d['ms_per_step'],2), 'ms/step', d['config'].get('sampler_rounds_mean'), d['config'].get('host_syncs_per_step'))"
done
