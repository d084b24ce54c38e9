#!/bin/bash
# Round-2 GPU call 11: packed-fp32 limb split (chain / gemm / wgrad), x6p sampler trunk with weights two steps ahead (A/B)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
F='Warning\|warnings.warn\|WeightNorm\|kaiming'
echo "== chain + gemm tests"
timeout 400 python -m pytest tests/test_chain_gpu.py tests/test_gemm_gpu.py -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -v "$F" > gpurun_out/r11_tests_a.log; tail -2 gpurun_out/r11_tests_a.log
echo "== sampler trunk: product | PF2 (developer library) + its parity test"
HOLD_X6=1 timeout 100 python scripts/bench_fused.py 2>&1 | grep "^fused"
HOLD_X6=1 HOLD_LIB=hold_amd/libholdhip_dev.so HOLD_X6P_PF2=1 timeout 100 python scripts/bench_fused.py 2>&1 | grep "^fused"
HOLD_LIB=hold_amd/libholdhip_dev.so HOLD_X6P_PF2=1 timeout 200 python -m pytest tests/test_gemm_gpu.py -q -m gpu -k fused --tb=short -p no:cacheprovider 2>&1 | tail -2
echo "== chain / gemm micro-benchmarks"
HOLD_X6=1 timeout 200 python scripts/bench_chain.py 2>&1 | grep "^chain"
timeout 120 python scripts/bench_gemm.py 1605632 2>&1 | grep "^gemm_nt none\|^gemm_nt mul\|^wgrad"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get("roofline",{}); print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],1), d["config"].get("sampler_rounds_last_call"), {k:(round(v["achieved"],1),round(v["time_share"],3),v["launches"]) for k,v in r.get("kernels",{}).items()}, r.get("end_to_end"))'
B="python bench.py --no-cpu-baseline --steps 2 --warmup 1"
echo "== bench: product | PF2"
timeout 300 $B 2>gpurun_out/b1.err | tee gpurun_out/bench_r11.json | python -c "$P" full_x6
HOLD_LIB=hold_amd/libholdhip_dev.so HOLD_X6P_PF2=1 timeout 300 $B 2>gpurun_out/b2.err | tee gpurun_out/bench_r11_pf2.json | python -c "$P" full_x6_pf2
echo "== path tests"
timeout 600 python -m pytest tests/test_path_gpu.py tests/test_train_targets_gpu.py -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -v "$F" > gpurun_out/r11_tests_b.log; tail -2 gpurun_out/r11_tests_b.log
for f in gpurun_out/b?.err; do echo "-- $f"; grep -v "$F\|amdgpu.ids" $f | tail -4; done
echo "== failures"
for f in gpurun_out/r11_tests_a.log gpurun_out/r11_tests_b.log; do grep -n "^FAILED\|^ERROR\|passed\|failed" $f | tail -12; done
