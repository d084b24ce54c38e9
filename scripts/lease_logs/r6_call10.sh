#!/bin/bash
# round 6, GPU call 10: first hardware run of hold_gemm_h3 (csrc/rgemm_h3.hip): kernel tests, micro-benchmark, then the path tests
# that run through it and the headline A/B against HOLD_H3_GEMM=0
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r6c10; mkdir -p $O
timeout 900 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x -k "gemm_h3 or gemm_r6" > $O/pytest_gemm.log 2>&1; rc=$?; echo "gemm tests rc=$rc"; tail -3 $O/pytest_gemm.log | cut -c1-200
if [ $rc -ne 0 ]; then grep -E "^E  |FAILED|Error" $O/pytest_gemm.log | head -30 | cut -c1-300; fi
timeout 300 python scripts/bench_rgemm.py > $O/bench_rgemm.log 2>&1; grep -E "gemm_h3" $O/bench_rgemm.log | cut -c1-200
timeout 900 python -m pytest tests/test_path_gpu.py tests/test_compact_gpu.py -m gpu -q -x > $O/pytest_path.log 2>&1; echo "path tests rc=$?"; tail -2 $O/pytest_path.log | cut -c1-200
grep -E "^E  |FAILED" $O/pytest_path.log | head -20 | cut -c1-300
line() { python -c "
import json
try:
    d = json.load(open('$1')); e = d['roofline']['kernels']
    print('$1'.split('/')[-1], d['config'].get('f16x3_launches_recomputed_in_f32x6_per_step'), round(d['value'], 1), round(d['ms_per_step'], 2), {k: (round(v['avg_launch_ms'], 2), round(v['time_share'] * 100, 1)) for k, v in e.items() if 'rgemm' in k})
except Exception as ex: print('no line', ex)
"; }
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_h3gemm_on.json 2> $O/bench_on.err; line $O/bench_h3gemm_on.json
HOLD_H3_GEMM=0 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_h3gemm_off.json 2> $O/bench_off.err; line $O/bench_h3gemm_off.json
