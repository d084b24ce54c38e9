#!/bin/bash
# round 4, GPU call 13: the reference's 1 280-ray step (C3) without instrumentation: speculative sampler rounds on / off
cd /root/repo; O=/root/repo/gpurun_out/r4c13; mkdir -p $O
for v in 1 0 1 0; do
  HOLD_SAMPLER_SPECULATE=$v timeout 200 python bench.py --mode c3 --steps 40 --warmup 10 --no-cpu-baseline --no-refine > $O/c3_spec$v.json 2> $O/c3_spec$v.err; echo "c3 spec=$v rc=$?"
  python - <<PY
import json
d = json.load(open("$O/c3_spec$v.json"))
print("  spec=$v", round(d["ms_per_step"], 2), "ms/step", round(d["value"], 1), "rays/s; in MFMA kernels", round(d["roofline"]["end_to_end"]["time_in_mfma_kernels"], 3), "calls/step", d["config"]["c_abi_calls_per_step"])
PY
done
