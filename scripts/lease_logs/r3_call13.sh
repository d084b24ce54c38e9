#!/bin/bash
# round 3, GPU call 13: where the reference's own 1 280-ray training step (configs[2]) spends its launches
cd /root/repo; O=/root/repo/gpurun_out/r3c13; mkdir -p $O
timeout 300 python bench.py --mode c3 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench_c3.json")); print(d["value"], d["ms_per_step"], d["roofline"]["end_to_end"], d["config"].get("c_abi_calls_per_step"))
PY
timeout 300 python bench.py --mode c3 --steps 3 --warmup 5 --no-cpu-baseline --no-refine --torch-profile $O/c3_ops.txt > /dev/null 2> $O/prof.err; echo "prof rc=$?"
head -80 $O/c3_ops.txt | cut -c1-200
