#!/bin/bash
# First GPU call of round 2: everything that was written without hardware at the end of round 1, in one gpurun
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash scripts/round2_first_calls.sh > gpurun_out/r2_first.log 2>&1; tail -60 gpurun_out/r2_first.log'
cd "$(dirname "$0")/.."
echo "== gated tests (x6 variants, mesh sdf, full 512x512 frame)"
HOLD_TEST_EXPERIMENTAL=1 timeout 300 python -m pytest tests -q -m gpu -k "x6 or mesh_sdf or full_frame"   # incl. test_wgrad_x6_matches_fp64 2>&1 | tail -8
echo "== split-precision sampler trunk, kernel level (variant 0 = on-the-fly split, 1 = limb planes)"
timeout 100 python scripts/dbg_x6.py 2>&1 | grep -v Warn | tail -8
echo "== sampler / end-to-end parity tests with the x6 trunk in the loop"
for v in 0 1; do
  echo "-- variant $v"
  HOLD_FUSED_SDF_X6=1 HOLD_FUSED_X6_VARIANT=$v timeout 300 python -m pytest tests/test_path_gpu.py -q -m gpu 2>&1 | tail -4
done
echo "== bench, fp32 sampler trunk vs x6 variants"
timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32', d['value'], d['ms_per_step'])"
for v in 0 1; do
  HOLD_FUSED_SDF_X6=1 HOLD_FUSED_X6_VARIANT=$v timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('x6 variant $v', d['value'], d['ms_per_step'], d['roofline']['kernels'].get('fused_sdf_kernel'))"
done
echo "== split-precision wgrad (HOLD_WGRAD_X6=1): micro + end to end"
timeout 100 python scripts/bench_gemm.py 2>&1 | grep wgrad
HOLD_WGRAD_X6=1 timeout 100 python scripts/bench_gemm.py 2>&1 | grep wgrad
HOLD_WGRAD_X6=1 HOLD_X6_SPLIT=trunc timeout 100 python scripts/bench_gemm.py 2>&1 | grep wgrad
HOLD_TEST_EXPERIMENTAL=1 HOLD_X6_SPLIT=trunc timeout 100 python -m pytest tests/test_gemm_gpu.py -q -m gpu -k wgrad_x6 2>&1 | tail -2
HOLD_WGRAD_X6=1 timeout 300 python -m pytest tests/test_path_gpu.py tests/test_chain_gpu.py -q -m gpu 2>&1 | tail -3
HOLD_WGRAD_X6=1 timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wgrad x6', d['value'], d['ms_per_step'], d['roofline']['kernels'].get('wgrad_kernel'))"
