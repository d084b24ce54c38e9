for v in 129 128; do echo "== variant $v"; HOLD_FUSED_VARIANT=$v timeout 100 python scripts/bench_fused.py 2>&1 | grep fused; done
HOLD_FUSED_VARIANT=129 timeout 300 python -m pytest tests/test_gemm_gpu.py tests/test_path_gpu.py -x -q -m gpu -k "fused or sampler or eval" 2>&1 | tail -3
