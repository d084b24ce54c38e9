#!/bin/bash
# round 5, GPU call 10: exact sample compaction: kernel test, compacted == uncompacted step, bench --beta 0.005 with / without it
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r5c10; mkdir -p $O
timeout 600 python -m pytest tests/test_compact_gpu.py -x -q -s > $O/pytest_compact.log 2>&1; echo "compact tests rc=$?"; grep -E "compaction:|passed|failed|^E  " $O/pytest_compact.log | head -12 | cut -c1-400
timeout 600 python -m pytest tests/test_path_gpu.py -x -q > $O/pytest_path.log 2>&1; echo "path tests rc=$?"; tail -3 $O/pytest_path.log | cut -c1-300
for c in "" "--no-compact"; do
  timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --beta 0.005 $c > $O/bench_beta$c.json 2> $O/bench_beta$c.err; echo "bench --beta 0.005 $c rc=$?"
  python - <<PY
import json
try:
    d = json.load(open("$O/bench_beta$c.json")); r = d["roofline"]
    print(round(d["value"], 1), round(d["ms_per_step"], 2), d["config"].get("sigma_I"), d["config"]["loss"], d["config"]["sample_compaction"]["live_samples_last_call"])
except Exception as e:
    print("no line", e)
PY
done
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json; d = json.load(open('$O/bench.json')); print(round(d['value'],1), d['config']['loss'], d['config']['sample_compaction']['live_samples_last_call'])"
timeout 300 python bench.py --mode c3 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"; python -c "
import json; d = json.load(open('$O/bench_c3.json')); print('c3', round(d['value'],1), round(d['ms_per_step'],2), d['config'].get('pose_refine'))"
