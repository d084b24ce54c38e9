#!/bin/bash
# Round-2 final GPU call: the whole GPU suite in both arithmetics, smoke(), every bench line quoted in DESIGN.md, and the
# rocprofv3 evidence for profiles/ (kernel stats of the default command + the two --pmc passes).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/final
O=gpurun_out/final
F='Warning\|warnings.warn\|WeightNorm\|kaiming'
echo "== gpu suite, default precision (f32x6)"
timeout 1000 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -v "$F" > $O/tests_x6.log; tail -3 $O/tests_x6.log
echo "== gpu suite, fp32 MFMA (everything but the 170 GB scale tests)"
HOLD_PRECISION=f32 timeout 800 python -m pytest tests/test_path_gpu.py tests/test_chain_gpu.py tests/test_gemm_gpu.py tests/test_points_gpu.py tests/test_train_targets_gpu.py tests/test_fitting_gpu.py -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -v "$F" > $O/tests_f32.log; tail -3 $O/tests_f32.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "$F" | tail -2
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get("roofline",{}); print(sys.argv[1], round(d["value"],1), d["unit"], round(d["ms_per_step"],1), "ms/step", {k:(round(v["achieved"],1),round(v["time_share"],3)) for k,v in r.get("kernels",{}).items()}, r.get("end_to_end"), d.get("cpu_baseline"), d["config"].get("pose_refine"))'
echo "== bench lines"
timeout 600 python bench.py 2>$O/b_default.err | tee $O/bench_default.json | python -c "$P" default
timeout 300 python bench.py --fp32-mfma --no-cpu-baseline 2>$O/b_f32.err | tee $O/bench_fp32_mfma.json | python -c "$P" fp32_mfma
timeout 300 python bench.py --mode render --no-cpu-baseline 2>$O/b_render.err | tee $O/bench_render.json | python -c "$P" render
timeout 300 python bench.py --mode c3 --steps 20 --warmup 5 2>$O/b_c3.err | tee $O/bench_c3.json | python -c "$P" c3
timeout 400 python bench.py --mode c5 --no-cpu-baseline --steps 1 --warmup 1 2>$O/b_c5.err | tee $O/bench_c5.json | python -c "$P" c5
timeout 300 python bench.py --two-hands --no-cpu-baseline 2>$O/b_two.err | tee $O/bench_twohands.json | python -c "$P" two_hands
timeout 300 python bench.py --loss pixel --no-cpu-baseline 2>$O/b_pixel.err | tee $O/bench_pixel_loss.json | python -c "$P" pixel_loss
echo "== rocprofv3: kernel stats + FETCH_SIZE / WRITE_SIZE passes"
bash scripts/prof_r02.sh 2>&1 | tail -12
cp gpurun_out/prof_r02/* $O/ 2>/dev/null
for f in $O/b_*.err; do echo "-- $f"; grep -v "$F\|amdgpu.ids" $f | tail -3; done
echo "== failures"
for f in $O/tests_x6.log $O/tests_f32.log; do grep -n "^FAILED\|^ERROR\|passed\|failed\|skipped" $f | tail -12; done
