#!/bin/bash
# round 5, GPU call 4: rmlp_h3 v2 (ring of 4 compile-time slots, DMA pairs, packed fp32 epilogue, one LDS wait per group) and the
# reworked wgrad_h3 prologue: kernel tests, A/B timings, structured-operand error of both weight-gradient arithmetics, bench
cd /root/repo; O=/root/repo/gpurun_out/r5c4; mkdir -p $O
timeout 900 python -m pytest tests/test_rmlp_gpu.py -x -q -s > $O/pytest_rmlp.log 2>&1; echo "rmlp tests rc=$?"; tail -6 $O/pytest_rmlp.log | cut -c1-250
timeout 600 python scripts/bench_rmlp.py > $O/bench_rmlp.log 2>&1; echo "bench_rmlp rc=$?"; grep -E "h3|trunk_h3" $O/bench_rmlp.log | cut -c1-250
timeout 900 python -m pytest tests/test_gemm_gpu.py -q -k "wgrad" > $O/pytest_wgrad.log 2>&1; echo "wgrad tests rc=$?"; tail -6 $O/pytest_wgrad.log | cut -c1-250; grep -E "^E  .*assert|AssertionError: assert" $O/pytest_wgrad.log | head -5 | cut -c1-200
timeout 300 python scripts/bench_wgrad_h3.py > $O/bench_wgrad_h3.log 2>&1; echo "bench_wgrad rc=$?"; cat $O/bench_wgrad_h3.log | cut -c1-250
python - <<'PY' 2>&1 | tail -5
import torch, hold_amd
from hold_amd import gemm
dev = "cuda:0"
for P in (200000, 65536 + 16, 4096):
    R = torch.zeros(P, 256, device=dev); X = torch.zeros(P, 256, device=dev)
    R[:, 37] = 1.0
    X[:, 201] = torch.arange(P, device=dev, dtype=torch.float32) % 7 + 0.123456789
    ref = float(X[:, 201].double().sum())
    for mode in ("f32x6", "f16x3", "f32"):
        hold_amd.set_precision(mode)
        dW = torch.empty(256, 256, device=dev)
        gemm.wgrad(R, X, dW, None)
        print(f"structured P={P} {mode}: rel err {abs(float(dW[37, 201].double()) - ref) / ref:.3e}")
PY
timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench.json")); r = d["roofline"]
print(round(d["value"], 1), round(d["ms_per_step"], 2), d["config"].get("sigma_I"), r["bound"], round(r["frac"], 3), r["end_to_end"]["time_in_mfma_kernels"], d["config"]["loss"])
for k, v in sorted(r["kernels"].items(), key=lambda kv: -kv[1]["time_share"])[:9]:
    print("   ", k, round(v["time_share"], 4), round(v.get("fp32_equivalent_tflops", 0), 1), round(v["avg_launch_ms"], 3))
PY
