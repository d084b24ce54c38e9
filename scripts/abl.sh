for m in 0 256 768 1024 1792; do echo "== HOLD_GEMM_DEBUG=$m"; HOLD_GEMM_DEBUG=$m timeout 100 python scripts/bench_gemm.py 2>&1 | grep "none\|softplus "; done
