#!/bin/bash
# round 4, GPU call 14: the background net's backward as one register-resident descending sweep (skip width 172): chain +
# end-to-end gradient tests, then the headline bench line
cd /root/repo; O=/root/repo/gpurun_out/r4c14; mkdir -p $O
timeout 600 python -m pytest tests/test_chain_gpu.py tests/test_path_gpu.py tests/test_train_targets_gpu.py -q -x > $O/pytest_sel.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/pytest_sel.log | cut -c1-220
if [ $rc -ne 0 ]; then exit 0; fi
timeout 500 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench.json"))
print("rays/s", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 1), "sigmaI", d["config"]["sigma_I"])
r = d["roofline"]
for k, v in r["kernels"].items():
    if "mfma_frac" in v:
        print(f"  {k:22s} share {v['time_share']:.3f} TF-eq {v['fp32_equivalent_tflops']:.1f} avg_ms {v['avg_launch_ms']:.3f} launches {v['launches']}")
print(r["end_to_end"])
PY
