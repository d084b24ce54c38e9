#!/bin/bash
# round 6, GPU call 40: hold_gemm_h3's result stores as whole lines through wave-private LDS tiles (the sweeps' result tiles) instead of 32-byte
# row fragments: GEMM tests, bit-reproducibility probe, micro-benchmark per shape against the previous build (hold_amd/libholdhip_prev.so),
# path / invariants tests, headline A/B previous / new build, alternating, same box
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r6c40; mkdir -p $O
timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q > $O/pytest_gemm.log 2>&1; rc=$?; echo "gemm tests rc=$rc"; tail -3 $O/pytest_gemm.log | cut -c1-200
if [ $rc -ne 0 ]; then grep -E "^E  |FAILED|Error" $O/pytest_gemm.log | head -30 | cut -c1-300; fi
timeout 600 python scripts/probes/gemm_h3_flake.py 2000 > $O/flake.log 2>&1; tail -2 $O/flake.log | cut -c1-300
echo "--- micro-benchmark, new build"; timeout 300 python scripts/bench_rgemm.py > $O/rgemm_new.log 2>&1; grep gemm_h3 $O/rgemm_new.log | cut -c1-110
echo "--- micro-benchmark, previous build"; HOLD_LIB=/root/repo/hold_amd/libholdhip_prev.so timeout 300 python scripts/bench_rgemm.py > $O/rgemm_prev.log 2>&1; grep gemm_h3 $O/rgemm_prev.log | cut -c1-110
timeout 1500 python -m pytest tests/test_path_gpu.py tests/test_scale_gpu.py tests/test_dropin_gpu.py tests/test_compact_gpu.py -x -q > $O/pytest_path.log 2>&1; rc=$?; echo "path tests rc=$rc"; tail -2 $O/pytest_path.log | cut -c1-200
if [ $rc -ne 0 ]; then grep -E "^E  |FAILED|Error" $O/pytest_path.log | head -30 | cut -c1-300; fi
run() { name=$1; shift; envs=(); while [[ "$1" == *=* ]]; do envs+=("$1"); shift; done; env "${envs[@]}" timeout 400 python bench.py --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python -c "
import json
try:
    d = json.load(open('$O/bench_$name.json')); k = d['roofline']['kernels'].get('rgemm_h3_kernel', {}); print('$name', round(d['ms_per_step'], 2), 'ms/step', round(d['value'], 1), 'rays/s', d['config'].get('sigma_I'), 'rgemm_h3 avg ms', round(k.get('avg_launch_ms', 0), 4), d['config'].get('f16x3_launches_recomputed_in_f32x6_per_step'))
except Exception as e: print('$name no line', e)
"; }
for i in 1 2; do
  run new_$i X=1 --steps 4 --warmup 2
  run prev_$i HOLD_LIB=/root/repo/hold_amd/libholdhip_prev.so --steps 4 --warmup 2
done
