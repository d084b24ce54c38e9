#!/bin/bash
# round 6, GPU call 34: the backward's ReLU masks as BITS (hold_gemm_h3_bits: the forward ReLU launches write 32 bytes per point, the masked
# input-gradient launches read them instead of streaming the fp32 activation again): GEMM tests, end-to-end gradient / drop-in / invariants
# tests, the bit-reproducibility probe, headline A/B with HOLD_RELU_BITS=0, alternating, same box
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r6c34; mkdir -p $O
timeout 1500 python -m pytest tests/test_gemm_gpu.py tests/test_path_gpu.py tests/test_dropin_gpu.py tests/test_scale_gpu.py tests/test_compact_gpu.py -x -q > $O/pytest.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/pytest.log | cut -c1-200
if [ $rc -ne 0 ]; then grep -E "^E  |FAILED|Error" $O/pytest.log | head -30 | cut -c1-300; fi
timeout 600 python scripts/probes/gemm_h3_flake.py 1500 > $O/flake.log 2>&1; tail -2 $O/flake.log | cut -c1-300
run() { name=$1; shift; envs=(); while [[ "$1" == *=* ]]; do envs+=("$1"); shift; done; env "${envs[@]}" timeout 400 python bench.py --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python -c "
import json
try:
    d = json.load(open('$O/bench_$name.json')); k = d['roofline']['kernels'].get('rgemm_h3_kernel', {}); print('$name', round(d['ms_per_step'], 2), 'ms/step', round(d['value'], 1), 'rays/s', d['config'].get('sigma_I'), 'rgemm_h3 avg ms', round(k.get('avg_launch_ms', 0), 4), 'TB/step', round(d['roofline']['end_to_end']['hbm_bytes_per_step'] / 1e12, 3), d['config'].get('f16x3_launches_recomputed_in_f32x6_per_step'))
except Exception as e: print('$name no line', e)
"; }
for i in 1 2; do
  run bits_$i X=1 --steps 4 --warmup 2
  run nobits_$i HOLD_RELU_BITS=0 --steps 4 --warmup 2
done
run c3 X=1 --mode c3 --no-refine --steps 30 --warmup 5
