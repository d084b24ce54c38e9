#!/bin/bash
# Round-2 GPU call 4: surface-bearing synthetic scene, x6q trunk, L2 weight-stream ablation, full-loss benches
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
F='Warning\|warnings.warn\|WeightNorm\|kaiming'
echo "== gpu suite, default precision (f32x6, trunk p)"
timeout 900 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -v "$F" > gpurun_out/r4_tests_x6.log; tail -3 gpurun_out/r4_tests_x6.log
echo "== x6q trunk in the loop"
HOLD_X6_TRUNK=q timeout 600 python -m pytest tests/test_gemm_gpu.py tests/test_path_gpu.py -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -v "$F" > gpurun_out/r4_tests_q.log; tail -3 gpurun_out/r4_tests_q.log
echo "== fp32 MFMA everywhere (path / chain / gemm / training targets)"
HOLD_PRECISION=f32 timeout 600 python -m pytest tests/test_path_gpu.py tests/test_chain_gpu.py tests/test_gemm_gpu.py tests/test_train_targets_gpu.py -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -v "$F" > gpurun_out/r4_tests_f32.log; tail -3 gpurun_out/r4_tests_f32.log
echo "== sampler trunk micro-bench: fp32 / x6p / x6q / x6p without the weight stream (dev build)"
timeout 120 python scripts/bench_fused.py 2>&1 | tail -1
HOLD_X6_TRUNK=p timeout 120 python scripts/bench_fused.py 2>&1 | tail -1
HOLD_X6_TRUNK=q timeout 120 python scripts/bench_fused.py 2>&1 | tail -1
HOLD_LIB=hold_amd/libholdhip_dev.so HOLD_X6_TRUNK=p HOLD_X6P_NOSTREAM=1 timeout 120 python scripts/bench_fused.py 2>&1 | tail -2
echo "== fixture"
timeout 300 python scripts/record_hip_outputs.py gpurun_out/hip_train_output.npz 2>&1 | tail -1
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get("roofline",{}); print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],1), d["config"].get("sampler_rounds_last_call"), d["config"].get("c_abi_calls_per_step"), d["config"].get("pose_refine"), {k:(round(v["achieved"],1),round(v["time_share"],3),v["launches"]) for k,v in r.get("kernels",{}).items()}, r.get("end_to_end"))'
B="python bench.py --no-cpu-baseline --steps 2 --warmup 1"
timeout 300 $B --loss pixel --shape-report gpurun_out/shapes_r2.json 2>gpurun_out/b1.err | tee gpurun_out/bench_pixel.json | python -c "$P" pixel_x6p
HOLD_X6_TRUNK=q timeout 300 $B --loss pixel 2>gpurun_out/b2.err | tee gpurun_out/bench_pixel_q.json | python -c "$P" pixel_x6q
timeout 300 $B --loss full 2>gpurun_out/b3.err | tee gpurun_out/bench_full.json | python -c "$P" full_x6p
timeout 300 $B --loss pixel --fp32-mfma 2>gpurun_out/b4.err | tee gpurun_out/bench_pixel_f32.json | python -c "$P" pixel_f32
timeout 300 python bench.py --no-cpu-baseline --mode c3 --steps 20 --warmup 3 2>gpurun_out/b5.err | tee gpurun_out/bench_c3.json | python -c "$P" c3_full
timeout 300 python bench.py --no-cpu-baseline --mode c3 --loss pixel --no-refine --steps 20 --warmup 3 2>gpurun_out/b6.err | tee gpurun_out/bench_c3_pixel.json | python -c "$P" c3_pixel
timeout 300 python bench.py --no-cpu-baseline --mode render --steps 2 --warmup 1 2>gpurun_out/b7.err | tee gpurun_out/bench_render.json | python -c "$P" render
for f in gpurun_out/b?.err; do echo "-- $f"; grep -v "$F\|amdgpu.ids" $f | tail -5; done
echo "== failures"
for f in gpurun_out/r4_tests_x6.log gpurun_out/r4_tests_q.log gpurun_out/r4_tests_f32.log; do grep -n "^FAILED\|^ERROR\|passed\|failed" $f | tail -12; done
