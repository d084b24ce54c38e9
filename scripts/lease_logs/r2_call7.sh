#!/bin/bash
# Round-2 GPU call 7: split-precision gemm_nt, wide-tile x6 wgrad, vectorised wgrad reduce, cached weight packs
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
F='Warning\|warnings.warn\|WeightNorm\|kaiming'
echo "== gemm / chain / path / targets tests, default precision"
timeout 600 python -m pytest tests/test_gemm_gpu.py tests/test_chain_gpu.py tests/test_path_gpu.py tests/test_train_targets_gpu.py -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -v "$F" > gpurun_out/r7_tests_x6.log; tail -3 gpurun_out/r7_tests_x6.log
echo "== micro-benchmarks (product lib): x6 then fp32"
timeout 120 python scripts/bench_gemm.py 1605632 2>&1 | grep -v "$F\|amdgpu.ids"
HOLD_X6=0 timeout 120 python scripts/bench_gemm.py 1605632 2>&1 | grep -v "$F\|amdgpu.ids"
echo "== wgrad x6 tile A/B (developer lib)"
for t in 128 256; do HOLD_LIB=hold_amd/libholdhip_dev.so HOLD_WGRAD_X6_TILE=$t timeout 120 python scripts/bench_gemm.py 1605632 2>&1 | grep "^wgrad"; done
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get("roofline",{}); print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],1), d["config"].get("sampler_rounds_last_call"), {k:(round(v["achieved"],1),round(v["time_share"],3),v["launches"]) for k,v in r.get("kernels",{}).items()}, r.get("end_to_end"))'
B="python bench.py --no-cpu-baseline --steps 2 --warmup 1"
timeout 300 $B --shape-report gpurun_out/shapes_r7.json 2>gpurun_out/b1.err | tee gpurun_out/bench_r7.json | python -c "$P" full_x6
echo "== scale test + remaining gpu tests"
timeout 600 python -m pytest tests/test_scale_gpu.py tests/test_fitting_gpu.py -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -v "$F" > gpurun_out/r7_tests_scale.log; tail -3 gpurun_out/r7_tests_scale.log
for f in gpurun_out/b1.err; do echo "-- $f"; grep -v "$F\|amdgpu.ids" $f | tail -4; done
echo "== failures"
for f in gpurun_out/r7_tests_x6.log gpurun_out/r7_tests_scale.log; do grep -n "^FAILED\|^ERROR\|passed\|failed" $f | tail -12; done
