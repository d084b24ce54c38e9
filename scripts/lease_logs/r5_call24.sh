#!/bin/bash
# round 5, GPU call 24 (closing check of the final tree -- what the driver runs at round end): the whole GPU suite with the default
# arithmetic, smoke(), the default bench line, and the C3 line (the launch-stream getter of _lib.stream_ptr changed since call 16)
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r5c24; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "gpu suite rc=$rc"; tail -3 $O/pytest_gpu.log | cut -c1-200
if [ $rc -ne 0 ]; then grep -E "Error|assert|error|FAILED" $O/pytest_gpu.log | head -20 | cut -c1-250; fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log | cut -c1-200
timeout 700 python bench.py > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench_final.json")); r = d["roofline"]
print(round(d["value"], 1), round(d["ms_per_step"], 2), r["bound"], round(r["frac"], 3), r.get("traffic"), d["cpu_baseline"]["value"], d["cpu_baseline"].get("step_s"), r["end_to_end"]["time_in_mfma_kernels"])
PY
for i in 1 2 3; do timeout 300 python bench.py --mode c3 --steps 40 --warmup 8 --no-cpu-baseline --no-refine 2> $O/c3.err | python -c "import json,sys; d=json.load(sys.stdin); print('c3', round(d['ms_per_step'],2), 'ms/step')"; done
