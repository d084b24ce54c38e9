#!/bin/bash
# round 3, GPU call 18: trunk pack shared between the node and ImplicitNet, gather-based axis-angle, pinned BARF upload
cd /root/repo; O=/root/repo/gpurun_out/r3c18; mkdir -p $O
timeout 900 python -m pytest tests/test_path_gpu.py tests/test_fitting_gpu.py -x -q > $O/pytest_a.log 2>&1; echo "a rc=$?"; tail -8 $O/pytest_a.log
timeout 300 python bench.py --mode c3 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"; tail -3 $O/bench_c3.err
python - <<PY
import json
d = json.load(open("$O/bench_c3.json")); print(d["value"], d["ms_per_step"], d["roofline"]["end_to_end"])
PY
timeout 300 python bench.py --mode c3 --steps 2 --warmup 3 --no-cpu-baseline --no-refine --sync-debug $O/c3_syncs.txt --op-sites $O/c3_sites.txt > /dev/null 2> $O/err.log; echo "rc=$?"
cat $O/c3_syncs.txt; head -60 $O/c3_sites.txt | cut -c1-200
