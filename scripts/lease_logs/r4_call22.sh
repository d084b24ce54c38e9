#!/bin/bash
# round 4, GPU call 22: the default bench line again, now that profiles/r04_pmc_traffic.json counts the whole-dW weight-gradient
# kernels under their present names (call 21's line priced the wgrad family's measured bytes from the grouped / tile kernels only)
cd /root/repo; O=/root/repo/gpurun_out/r4c22; mkdir -p $O
timeout 700 python bench.py > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench_final.json")); r = d["roofline"]
print(round(d["value"], 1), round(d["ms_per_step"], 2), d["config"].get("sigma_I"), r["bound"], round(r["frac"], 3), r.get("traffic"), r.get("hbm_frac_measured_bytes"), d["cpu_baseline"]["value"])
PY
