#!/bin/bash
# Round-2 GPU call 6: re-validation on the final synthetic scene + a rocprofv3 kernel-stats pass for the non-MFMA share
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
F='Warning\|warnings.warn\|WeightNorm\|kaiming'
echo "== gpu suite, default precision (f32x6)"
timeout 900 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -v "$F" > gpurun_out/r6_tests_x6.log; tail -3 gpurun_out/r6_tests_x6.log
echo "== fp32 MFMA everywhere (everything but the scale tests)"
HOLD_PRECISION=f32 timeout 600 python -m pytest tests/test_path_gpu.py tests/test_chain_gpu.py tests/test_gemm_gpu.py tests/test_train_targets_gpu.py tests/test_fitting_gpu.py -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -v "$F" > gpurun_out/r6_tests_f32.log; tail -3 gpurun_out/r6_tests_f32.log
echo "== fixture"
timeout 300 python scripts/record_hip_outputs.py gpurun_out/hip_train_output.npz 2>&1 | tail -1
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get("roofline",{}); print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],1), d["config"].get("sampler_rounds_last_call"), {k:(round(v["achieved"],1),round(v["time_share"],3),v["launches"]) for k,v in r.get("kernels",{}).items()}, r.get("end_to_end"))'
B="python bench.py --no-cpu-baseline --steps 2 --warmup 1"
timeout 300 $B --shape-report gpurun_out/shapes_r2c.json 2>gpurun_out/b1.err | tee gpurun_out/bench_full.json | python -c "$P" full_x6
timeout 300 $B --fp32-mfma 2>gpurun_out/b2.err | tee gpurun_out/bench_full_f32.json | python -c "$P" full_f32
echo "== rocprofv3 kernel stats of the default bench"
REPO=$(pwd); cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_stats
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o s -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $REPO/gpurun_out/bench_under_rocprof_r6.json 2> /tmp/prof_stats.err
find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} $REPO/gpurun_out/kernel_stats_r6.csv \;
cd $REPO; head -45 gpurun_out/kernel_stats_r6.csv | cut -c1-160
for f in gpurun_out/b?.err; do echo "-- $f"; grep -v "$F\|amdgpu.ids" $f | tail -4; done
echo "== failures"
for f in gpurun_out/r6_tests_x6.log gpurun_out/r6_tests_f32.log; do grep -n "^FAILED\|^ERROR\|passed\|failed" $f | tail -12; done
