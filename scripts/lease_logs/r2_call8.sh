#!/bin/bash
# Round-2 GPU call 8: KNN three-pass kernel, lin8 backward without the cotangent copy (rank-1 epilogue + wcolsum),
# aligned render-net input, and the A/B of three operand stages in the x6 GEMM / wgrad (developer library)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
F='Warning\|warnings.warn\|WeightNorm\|kaiming'
echo "== tests, default precision"
timeout 700 python -m pytest tests/test_points_gpu.py tests/test_gemm_gpu.py tests/test_chain_gpu.py tests/test_path_gpu.py tests/test_train_targets_gpu.py -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -v "$F" > gpurun_out/r8_tests_x6.log; tail -3 gpurun_out/r8_tests_x6.log
echo "== gemm/wgrad tests with three operand stages (developer library)"
HOLD_LIB=hold_amd/libholdhip_dev.so HOLD_GEMM_NBUF=3 HOLD_WGRAD_NBUF=3 timeout 300 python -m pytest tests/test_gemm_gpu.py -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -v "$F" > gpurun_out/r8_tests_nbuf3.log; tail -2 gpurun_out/r8_tests_nbuf3.log
echo "== micro-benchmarks: NBUF 2 | NBUF 3 | ablations of the NBUF-2 x6 GEMM (no prefetch, no wait, no epilogue)"
D="HOLD_LIB=hold_amd/libholdhip_dev.so"
G='^gemm_nt none\|^gemm_nt mul_dsp\|^wgrad'
env $D timeout 120 python scripts/bench_gemm.py 1605632 2>&1 | grep "$G"
env $D HOLD_GEMM_NBUF=3 HOLD_WGRAD_NBUF=3 timeout 120 python scripts/bench_gemm.py 1605632 2>&1 | grep "$G"
for dbg in 256 512 1024; do echo "-- HOLD_GEMM_DEBUG=$dbg"; env $D HOLD_GEMM_DEBUG=$dbg timeout 120 python scripts/bench_gemm.py 1605632 2>&1 | grep '^gemm_nt none'; done
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get("roofline",{}); print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],1), d["config"].get("sampler_rounds_last_call"), {k:(round(v["achieved"],1),round(v["time_share"],3),v["launches"]) for k,v in r.get("kernels",{}).items()}, r.get("end_to_end"))'
B="python bench.py --no-cpu-baseline --steps 2 --warmup 1"
echo "== bench"
timeout 300 $B --shape-report gpurun_out/shapes_r8.json 2>gpurun_out/b1.err | tee gpurun_out/bench_r8.json | python -c "$P" full_x6
echo "== bench, three stages"
env $D HOLD_GEMM_NBUF=3 HOLD_WGRAD_NBUF=3 timeout 300 $B 2>gpurun_out/b2.err | tee gpurun_out/bench_r8_nbuf3.json | python -c "$P" full_x6_nbuf3
echo "== rocprofv3 kernel stats of the default bench"
REPO=$(pwd); cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_stats
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o s -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $REPO/gpurun_out/bench_under_rocprof_r8.json 2> /tmp/prof_stats.err
find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} $REPO/gpurun_out/kernel_stats_r8.csv \;
cd $REPO; head -28 gpurun_out/kernel_stats_r8.csv | cut -c1-150
echo "== scale test"
timeout 600 python -m pytest tests/test_scale_gpu.py -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -v "$F" > gpurun_out/r8_tests_scale.log; tail -3 gpurun_out/r8_tests_scale.log
for f in gpurun_out/b?.err; do echo "-- $f"; grep -v "$F\|amdgpu.ids" $f | tail -4; done
echo "== failures"
for f in gpurun_out/r8_tests_x6.log gpurun_out/r8_tests_nbuf3.log gpurun_out/r8_tests_scale.log; do grep -n "^FAILED\|^ERROR\|passed\|failed" $f | tail -12; done
