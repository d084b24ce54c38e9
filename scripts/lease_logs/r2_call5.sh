#!/bin/bash
# Round-2 GPU call 5: split-precision layer chains (hold_chain_x6), lin8 split, fixed tests
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
F='Warning\|warnings.warn\|WeightNorm\|kaiming'
echo "== chain kernels, both arithmetics"
timeout 600 python -m pytest tests/test_chain_gpu.py -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -v "$F" > gpurun_out/r5_tests_chain.log; tail -3 gpurun_out/r5_tests_chain.log
echo "== chain micro-bench fp32 / x6"
timeout 200 python scripts/bench_chain.py 2>&1 | grep -v "$F" | head -5
HOLD_X6=1 timeout 200 python scripts/bench_chain.py 2>&1 | grep -v "$F" | head -5
echo "== gpu suite, default precision (f32x6)"
timeout 900 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -v "$F" > gpurun_out/r5_tests_x6.log; tail -3 gpurun_out/r5_tests_x6.log
echo "== fp32 MFMA everywhere (path / gemm / training targets / fitting)"
HOLD_PRECISION=f32 timeout 600 python -m pytest tests/test_path_gpu.py tests/test_gemm_gpu.py tests/test_train_targets_gpu.py tests/test_fitting_gpu.py -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -v "$F" > gpurun_out/r5_tests_f32.log; tail -3 gpurun_out/r5_tests_f32.log
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get("roofline",{}); print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],1), d["config"].get("sampler_rounds_last_call"), {k:(round(v["achieved"],1),round(v["time_share"],3),v["launches"]) for k,v in r.get("kernels",{}).items()}, r.get("end_to_end"))'
B="python bench.py --no-cpu-baseline --steps 2 --warmup 1"
timeout 300 $B --loss full --shape-report gpurun_out/shapes_r2b.json 2>gpurun_out/b1.err | tee gpurun_out/bench_full.json | python -c "$P" full_x6
timeout 300 $B --loss full --fp32-mfma 2>gpurun_out/b2.err | tee gpurun_out/bench_full_f32.json | python -c "$P" full_f32
timeout 300 python bench.py --no-cpu-baseline --mode c3 --steps 20 --warmup 3 2>gpurun_out/b3.err | tee gpurun_out/bench_c3.json | python -c "$P" c3_full
for f in gpurun_out/b?.err; do echo "-- $f"; grep -v "$F\|amdgpu.ids" $f | tail -4; done
echo "== failures"
for f in gpurun_out/r5_tests_chain.log gpurun_out/r5_tests_x6.log gpurun_out/r5_tests_f32.log; do grep -n "^FAILED\|^ERROR\|passed\|failed" $f | tail -12; done
