#!/bin/bash
# Round-2 GPU call 10: hold_chain_x6 with the limb planes in LDS (chain_x6p_kernel) against the split-on-fetch kernel
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
F='Warning\|warnings.warn\|WeightNorm\|kaiming'
echo "== chain tests (product library: limb planes)"
timeout 300 python -m pytest tests/test_chain_gpu.py -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -v "$F" > gpurun_out/r10_tests_chain.log; tail -3 gpurun_out/r10_tests_chain.log
echo "== chain micro-benchmark: limb planes | split on fetch (developer library)"
HOLD_X6=1 timeout 200 python scripts/bench_chain.py 2>&1 | grep "^chain"
HOLD_X6=1 HOLD_LIB=hold_amd/libholdhip_dev.so HOLD_CHAIN_X6_VARIANT=0 timeout 200 python scripts/bench_chain.py 2>&1 | grep "^chain"
echo "== path / targets / scale tests"
timeout 800 python -m pytest tests/test_path_gpu.py tests/test_train_targets_gpu.py tests/test_scale_gpu.py -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -v "$F" > gpurun_out/r10_tests_x6.log; tail -3 gpurun_out/r10_tests_x6.log
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get("roofline",{}); print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],1), d["config"].get("sampler_rounds_last_call"), {k:(round(v["achieved"],1),round(v["time_share"],3),v["launches"]) for k,v in r.get("kernels",{}).items()}, r.get("end_to_end"))'
B="python bench.py --no-cpu-baseline --steps 2 --warmup 1"
echo "== bench"
timeout 300 $B --shape-report gpurun_out/shapes_r10.json 2>gpurun_out/b1.err | tee gpurun_out/bench_r10.json | python -c "$P" full_x6
for f in gpurun_out/b?.err; do echo "-- $f"; grep -v "$F\|amdgpu.ids" $f | tail -4; done
echo "== failures"
for f in gpurun_out/r10_tests_chain.log gpurun_out/r10_tests_x6.log; do grep -n "^FAILED\|^ERROR\|passed\|failed" $f | tail -12; done
