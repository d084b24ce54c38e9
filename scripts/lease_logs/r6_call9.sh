#!/bin/bash
# round 6, GPU call 9: the headline line with the f16x3 sweeps (A/B against HOLD_H3_BWD=0 on the same box), the path / training
# tests that run through them, and how often the overflow guard fired
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r6c9; mkdir -p $O
line() { python -c "
import json
try:
    d = json.load(open('$1')); e = d['roofline']['kernels']
    print('$1'.split('/')[-1], d['config'].get('f16x3_launches_recomputed_in_f32x6_per_step'), round(d['value'], 1), round(d['ms_per_step'], 2), d['config'].get('sigma_I'), {k: (round(v['avg_launch_ms'], 2), round(v['time_share'] * 100, 1)) for k, v in e.items() if 'rchain' in k})
except Exception as ex: print('no line', ex)
"; }
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_h3bwd_on.json 2> $O/bench_on.err; line $O/bench_h3bwd_on.json
HOLD_H3_BWD=0 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_h3bwd_off.json 2> $O/bench_off.err; line $O/bench_h3bwd_off.json
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --beta 0.005 > $O/bench_beta005.json 2> $O/bench_b.err; line $O/bench_beta005.json
grep -h "overflow" $O/*.err | head
timeout 1500 python -m pytest tests/test_path_gpu.py tests/test_train_targets_gpu.py tests/test_compact_gpu.py tests/test_scale_gpu.py tests/test_dropin_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -3 $O/pytest.log | cut -c1-200
grep -E "^E  |FAILED" $O/pytest.log | head -20 | cut -c1-300
