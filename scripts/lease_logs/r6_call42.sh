#!/bin/bash
# round 6, GPU call 42: hold_trunk_h3 (STORE variant): the h rows leave as whole lines through wave-private LDS tiles (as hold_gemm_h3 since call 40)
# instead of 32-byte row fragments.  Trunk tests, bit-reproducibility probe,
# micro-benchmark against the previous build, path tests, headline A/B previous / new build, alternating, same box
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r6c42; mkdir -p $O
timeout 900 python -m pytest tests/test_rmlp_gpu.py -x -q > $O/pytest_rmlp.log 2>&1; rc=$?; echo "rmlp tests rc=$rc"; tail -2 $O/pytest_rmlp.log | cut -c1-200
if [ $rc -ne 0 ]; then grep -E "^E  |FAILED|Error" $O/pytest_rmlp.log | head -30 | cut -c1-300; fi
timeout 900 python scripts/probes/trunk_h3_flake.py 600 > $O/flake.log 2>&1; tail -3 $O/flake.log | cut -c1-300
timeout 1500 python -m pytest tests/test_path_gpu.py tests/test_scale_gpu.py tests/test_dropin_gpu.py -x -q > $O/pytest_path.log 2>&1; rc=$?; echo "path tests rc=$rc"; tail -2 $O/pytest_path.log | cut -c1-200
if [ $rc -ne 0 ]; then grep -E "^E  |FAILED|Error" $O/pytest_path.log | head -30 | cut -c1-300; fi
run() { name=$1; shift; envs=(); while [[ "$1" == *=* ]]; do envs+=("$1"); shift; done; env "${envs[@]}" timeout 400 python bench.py --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python -c "
import json
try:
    d = json.load(open('$O/bench_$name.json')); k = d['roofline']['kernels']; print('$name', round(d['ms_per_step'], 2), 'ms/step', round(d['value'], 1), 'rays/s', d['config'].get('sigma_I'), 'trunk avg ms', round(k['trunk_r6_kernel']['avg_launch_ms'], 4), 'rgemm', round(k['rgemm_h3_kernel']['avg_launch_ms'], 4), d['config'].get('f16x3_launches_recomputed_in_f32x6_per_step'))
except Exception as e: print('$name no line', e)
"; }
for i in 1 2; do
  run new_$i X=1 --steps 4 --warmup 2
  run prev_$i HOLD_LIB=/root/repo/hold_amd/libholdhip_prev.so --steps 4 --warmup 2
done
