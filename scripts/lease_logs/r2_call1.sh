#!/bin/bash
# Round-2 GPU call 1: run everything written blind at the end of round 1 (trimmed: 2 timed steps per bench)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
echo "== full gpu suite with gated tests on"
HOLD_TEST_EXPERIMENTAL=1 timeout 600 python -m pytest tests -q -m gpu 2>&1 | grep -v Warning | tail -40
echo "== dbg_x6"
timeout 100 python scripts/dbg_x6.py 2>&1 | grep -v Warn | tail -8
for v in 0 1; do
  echo "-- path tests, x6 variant $v"
  HOLD_FUSED_SDF_X6=1 HOLD_FUSED_X6_VARIANT=$v timeout 300 python -m pytest tests/test_path_gpu.py -q -m gpu 2>&1 | tail -4
done
B="python bench.py --no-cpu-baseline --steps 2 --warmup 1"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], {k:(round(v["achieved"],1),round(v["time_share"],3)) for k,v in d["roofline"]["kernels"].items()})'
timeout 200 $B 2>/dev/null | python -c "$P" fp32
for v in 0 1; do
  HOLD_FUSED_SDF_X6=1 HOLD_FUSED_X6_VARIANT=$v timeout 200 $B 2>/dev/null | python -c "$P" x6v$v
done
echo "== wgrad x6"
timeout 100 python scripts/bench_gemm.py 2>&1 | grep wgrad
HOLD_WGRAD_X6=1 timeout 100 python scripts/bench_gemm.py 2>&1 | grep wgrad
HOLD_WGRAD_X6=1 HOLD_X6_SPLIT=trunc timeout 100 python scripts/bench_gemm.py 2>&1 | grep wgrad
HOLD_WGRAD_X6=1 timeout 300 python -m pytest tests/test_path_gpu.py tests/test_chain_gpu.py -q -m gpu 2>&1 | tail -3
HOLD_WGRAD_X6=1 timeout 200 $B 2>/dev/null | python -c "$P" wgradx6
