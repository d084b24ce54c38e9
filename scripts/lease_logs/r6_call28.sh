#!/bin/bash
# round 6, GPU call 28: non-temporal hint on the LOADS that stream whole lines exactly once -- the sweeps' side tiles (variant swnt), and the
# weight gradients' operand tiles too (variant swwgnt); stores and every re-touched line stay as they are (call 27: nt result stores double the
# kernels that store 32-byte row fragments -- they rely on L2 merging four partial writes per line -- and nt on rgemm's quarter-line input loads
# costs 20 %).  Headline, three builds alternating; parity of the sweeps / weight gradients under the hint
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r6c28; mkdir -p $O
run() { name=$1; shift; envs=(); while [[ "$1" == *=* ]]; do envs+=("$1"); shift; done; env "${envs[@]}" timeout 400 python bench.py --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python -c "
import json
try:
    d = json.load(open('$O/bench_$name.json')); k = d['roofline']['kernels']; print('$name', round(d['ms_per_step'], 2), 'ms/step', round(d['value'], 1), 'rays/s', d['config'].get('sigma_I'), {n: round(k[n]['avg_launch_ms'], 3) for n in ('rgemm_h3_kernel', 'rchain_h3_kernel', 'rchain_a2_h3_kernel', 'rchain_dbwd_h3_kernel', 'rchain_bg_h3_kernel', 'trunk_r6_kernel', 'wgrad_h3_kernel', 'wgrad_kernel', 'fused_sdf_kernel') if n in k})
except Exception as e: print('$name no line', e)
"; }
for i in 1 2; do
  run base_$i X=1 --steps 4 --warmup 2
  run swnt_$i HOLD_LIB=/root/repo/hold_amd/libholdhip_swnt.so --steps 4 --warmup 2
  run swwgnt_$i HOLD_LIB=/root/repo/hold_amd/libholdhip_swwgnt.so --steps 4 --warmup 2
done
HOLD_LIB=/root/repo/hold_amd/libholdhip_swwgnt.so timeout 900 python -m pytest tests/test_chain_gpu.py tests/test_gemm_gpu.py -x -q > $O/pytest.log 2>&1; echo "tests (swwgnt) rc=$?"; tail -2 $O/pytest.log | cut -c1-200
