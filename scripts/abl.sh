for m in 0 256 512 768 1024; do echo "== HOLD_FUSED_DEBUG=$m"; HOLD_FUSED_DEBUG=$m timeout 100 python scripts/bench_gemm.py 2>&1 | grep "trunk fused"; done
