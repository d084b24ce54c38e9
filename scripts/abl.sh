for m in 0 512 1024 1536; do echo "== HOLD_GEMM_DEBUG=$m"; HOLD_GEMM_DEBUG=$m timeout 100 python scripts/bench_gemm.py 2>&1 | grep "none\|softplus "; done
