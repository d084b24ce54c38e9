echo "== 128"; timeout 100 python scripts/bench_fused.py 2>&1 | grep fused
for s in 0 1 2 3 4 6 2049 2051; do echo "== 64 stagger=$s"; HOLD_FUSED_VARIANT=64 HOLD_FUSED_STAGGER=$s timeout 100 python scripts/bench_fused.py 2>&1 | grep fused; done
