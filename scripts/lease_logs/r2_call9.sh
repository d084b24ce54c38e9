#!/bin/bash
# Round-2 GPU call 9: colour-head streaming kernels, invalid-tile skips, vectorised bias reduce
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
F='Warning\|warnings.warn\|WeightNorm\|kaiming'
echo "== tests, default precision"
timeout 800 python -m pytest tests/test_gemm_gpu.py tests/test_points_gpu.py tests/test_chain_gpu.py tests/test_path_gpu.py tests/test_train_targets_gpu.py tests/test_scale_gpu.py -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -v "$F" > gpurun_out/r9_tests_x6.log; tail -3 gpurun_out/r9_tests_x6.log
echo "== fp32 subset"
HOLD_PRECISION=f32 timeout 400 python -m pytest tests/test_path_gpu.py tests/test_train_targets_gpu.py -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -v "$F" > gpurun_out/r9_tests_f32.log; tail -2 gpurun_out/r9_tests_f32.log
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get("roofline",{}); print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],1), d["config"].get("sampler_rounds_last_call"), {k:(round(v["achieved"],1),round(v["time_share"],3),v["launches"]) for k,v in r.get("kernels",{}).items()}, r.get("end_to_end"))'
B="python bench.py --no-cpu-baseline --steps 2 --warmup 1"
echo "== bench"
timeout 300 $B --shape-report gpurun_out/shapes_r9.json 2>gpurun_out/b1.err | tee gpurun_out/bench_r9.json | python -c "$P" full_x6
echo "== rocprofv3 kernel stats of the default bench"
REPO=$(pwd); cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_stats
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o s -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $REPO/gpurun_out/bench_under_rocprof_r9.json 2> /tmp/prof_stats.err
find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} $REPO/gpurun_out/kernel_stats_r9.csv \;
cd $REPO; head -30 gpurun_out/kernel_stats_r9.csv | cut -c1-150
for f in gpurun_out/b?.err; do echo "-- $f"; grep -v "$F\|amdgpu.ids" $f | tail -4; done
echo "== failures"
for f in gpurun_out/r9_tests_x6.log gpurun_out/r9_tests_f32.log; do grep -n "^FAILED\|^ERROR\|passed\|failed" $f | tail -12; done
