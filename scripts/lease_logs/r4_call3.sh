#!/bin/bash
# round 4, GPU call 3: power-of-two weight ring (compile-time slot addresses): chain parity, micro-benchmark incl. the ring
# depth A/B of the one-side-input sweep (developer build), and the headline bench line on this tree
cd /root/repo; O=/root/repo/gpurun_out/r4c3; mkdir -p $O
timeout 400 python -m pytest tests/test_chain_gpu.py -q -x > $O/pytest_chain.log 2>&1; echo "chain tests rc=$?"; tail -4 $O/pytest_chain.log | cut -c1-200
for v in 1 3 1 3; do
  echo "== HOLD_R6_DIST=$v (one-side-input DSP only)"
  HOLD_LIB=/root/repo/hold_amd/libholdhip_dev.so HOLD_R6_DIST=$v HOLD_X6=1 timeout 200 python scripts/bench_chain.py 2>&1 | grep -E "r6" | tee -a $O/ab.log
done
timeout 500 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench.json"))
print("rays/s", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 1), "sigmaI", d["config"]["sigma_I"])
r = d["roofline"]
print("dominant:", r["kernel"][:40], r["bound"], round(r["frac"], 3), "mfma_frac", round(r["mfma_frac"], 3), "hbm_frac", round(r["hbm_frac"], 3))
for k, v in r["kernels"].items():
    if "mfma_frac" in v:
        print(f"  {k:22s} share {v['time_share']:.3f} TF-eq {v['fp32_equivalent_tflops']:.1f} mfma {v['mfma_frac']:.3f} hbm {v['hbm_frac']:.3f} bound {v['bound']} avg_ms {v['avg_launch_ms']:.3f}")
print(r["end_to_end"])
PY
