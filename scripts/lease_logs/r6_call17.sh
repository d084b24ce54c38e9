#!/bin/bash
# round 6, GPU call 17: hold_gemm_h3 / hold_gemm_r6 with both stores in every k step (the vmcnt race of the store-less steps) and the
# next block's row maximum requested by an assembler load: stress probe, kernel tests, micro-benchmark, the bg sweep in f16x3
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r6c17; mkdir -p $O
timeout 600 python scripts/probes/gemm_h3_flake.py 6000 > $O/flake.log 2>&1; tail -6 $O/flake.log | cut -c1-300
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_chain_gpu.py tests/test_scale_gpu.py -m gpu -q -x -k "gemm_h3 or gemm_r6 or background or full_frame" > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -2 $O/pytest.log | cut -c1-200
grep -E "^E  |FAILED" $O/pytest.log | head -10 | cut -c1-400
timeout 300 python scripts/bench_rgemm.py > $O/bench_rgemm.log 2>&1; grep -E "gemm_h3" $O/bench_rgemm.log | cut -c1-200
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); e=d['roofline']['kernels']; print('headline', round(d['value'],1), round(d['ms_per_step'],2), {k: (round(v['avg_launch_ms'],3), round(v['time_share']*100,1)) for k,v in e.items() if 'rgemm' in k or 'bg' in k})"
