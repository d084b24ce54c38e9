#!/bin/bash
# Round-2 GPU call 3: re-validation after the fixes of call 2 + full-loss benches
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
echo "== gpu suite, default precision (f32x6)"
timeout 900 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -v "Warning\|warnings.warn\|WeightNorm\|kaiming" > gpurun_out/r3_tests_x6.log
tail -5 gpurun_out/r3_tests_x6.log
echo "== gpu suite, fp32 MFMA everywhere"
HOLD_PRECISION=f32 timeout 900 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -v "Warning\|warnings.warn\|WeightNorm\|kaiming" > gpurun_out/r3_tests_f32.log
tail -5 gpurun_out/r3_tests_f32.log
echo "== smoke"
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
echo "== fixture"
timeout 300 python scripts/record_hip_outputs.py gpurun_out/hip_train_output.npz 2>&1 | tail -2
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get("roofline",{}); print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],1), d["config"].get("c_abi_calls_per_step"), d["config"].get("pose_refine"), {k:(round(v["achieved"],1),round(v["time_share"],3)) for k,v in r.get("kernels",{}).items()}, r.get("end_to_end"))'
B="python bench.py --no-cpu-baseline --steps 2 --warmup 1"
timeout 300 $B --loss pixel --shape-report gpurun_out/shapes_r2.json 2>gpurun_out/b1.err | tee gpurun_out/bench_pixel.json | python -c "$P" pixel_x6
timeout 300 $B --loss full 2>gpurun_out/b2.err | tee gpurun_out/bench_full.json | python -c "$P" full_x6
timeout 300 $B --loss pixel --fp32-mfma 2>gpurun_out/b3.err | tee gpurun_out/bench_pixel_f32.json | python -c "$P" pixel_f32
timeout 300 python bench.py --no-cpu-baseline --mode c3 --steps 20 --warmup 3 2>gpurun_out/b4.err | tee gpurun_out/bench_c3.json | python -c "$P" c3_full
timeout 300 python bench.py --no-cpu-baseline --mode c3 --loss pixel --no-refine --steps 20 --warmup 3 2>gpurun_out/b5.err | tee gpurun_out/bench_c3_pixel.json | python -c "$P" c3_pixel
timeout 300 python bench.py --no-cpu-baseline --mode c5 --steps 1 --warmup 1 2>gpurun_out/b6.err | tee gpurun_out/bench_c5.json | python -c "$P" c5
for f in gpurun_out/b?.err; do echo "-- $f"; grep -v "Warning\|warnings.warn\|WeightNorm\|kaiming\|amdgpu.ids" $f | tail -6; done
echo "== failures (x6 suite)"
grep -n "^FAILED\|^ERROR\|passed\|failed" gpurun_out/r3_tests_x6.log | tail -30
grep -n "^FAILED\|^ERROR\|passed\|failed" gpurun_out/r3_tests_f32.log | tail -30
