#!/bin/bash
# round 4, GPU call 7: whole-dW weight gradient for N = 217 and K = 272 / 304 (select-based bias sums, poisoned-column test):
# full GPU suite, then the headline bench line
cd /root/repo; O=/root/repo/gpurun_out/r4c7; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "suite rc=$?"; tail -8 $O/pytest_gpu.log | cut -c1-250
timeout 500 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench.json"))
print("rays/s", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 1), "sigmaI", d["config"]["sigma_I"])
r = d["roofline"]
print("dominant:", r["kernel"][:40], r["bound"], round(r["frac"], 3), "mfma_frac", round(r["mfma_frac"], 3), "hbm_frac", round(r["hbm_frac"], 3))
for k, v in r["kernels"].items():
    if "mfma_frac" in v:
        print(f"  {k:22s} share {v['time_share']:.3f} TF-eq {v['fp32_equivalent_tflops']:.1f} mfma {v['mfma_frac']:.3f} hbm {v['hbm_frac']:.3f} meas {v['hbm_frac_measured_bytes']} bound {v['bound']} avg_ms {v['avg_launch_ms']:.3f}")
print(r["end_to_end"])
PY
