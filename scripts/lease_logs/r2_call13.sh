#!/bin/bash
# Round-2 GPU call 13 (provenance): A/B harness used for the wide / ping-pong / full-height layer-chain variants against the
# product kernel (developer-library switch HOLD_CHAIN_X6_PP, variants since removed -- DESIGN.md section 4 has the numbers;
# with the current library both legs time the product kernel)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
F='Warning\|warnings.warn\|WeightNorm\|kaiming'
D="HOLD_LIB=hold_amd/libholdhip_dev.so"
echo "== chain tests with the ping-pong kernel"
env $D HOLD_CHAIN_X6_PP=2 timeout 300 python -m pytest tests/test_chain_gpu.py -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -v "$F" > gpurun_out/r13_tests_chain.log; tail -3 gpurun_out/r13_tests_chain.log
echo "== micro-benchmark: eight waves | ping-pong"
HOLD_X6=1 env $D timeout 200 python scripts/bench_chain.py 2>&1 | grep "^chain"
HOLD_X6=1 env $D HOLD_CHAIN_X6_PP=2 timeout 200 python scripts/bench_chain.py 2>&1 | grep "^chain"
echo "== failures"
grep -n "^FAILED\|^ERROR\|passed\|failed" gpurun_out/r13_tests_chain.log | tail
