#!/bin/bash
# round 4, GPU call 26: KNN threshold pass with one median-of-three per slot (same tau): kernel tests (778 / 61 / 800 vertices),
# end-to-end golden tests (the K = 15 selections of the reference), bench line
cd /root/repo; O=/root/repo/gpurun_out/r4c26; mkdir -p $O
timeout 900 python -m pytest tests/test_points_gpu.py tests/test_path_gpu.py -q -x > $O/pytest_sel.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/pytest_sel.log | cut -c1-220
if [ $rc -ne 0 ]; then grep -E "Error|assert|error" $O/pytest_sel.log | head -20 | cut -c1-220; exit 0; fi
timeout 400 python bench.py --no-cpu-baseline --no-refine > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -o s -- python /root/repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-refine > /dev/null 2> /tmp/kp.err
find /tmp/kp -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
cd /root/repo
python - <<PY
import json, csv
d = json.load(open("$O/bench.json"))
print("rays/s", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 1), d["roofline"]["end_to_end"]["time_in_mfma_kernels"])
for r in csv.DictReader(open("$O/kernel_stats.csv")):
    if any(k in r["Name"] for k in ("knn_invlbs", "seed_dsp", "embed_bwd2", "frame_colsum")):
        print(r["Name"][:60], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
