#!/bin/bash
# round 4, GPU call 31: final tree (incl. hold_gemm_narrow_x6): full GPU suite, build() + smoke(),
# rocprofv3 kernel stats of the default bench command, the default line with its cpu_baseline leg
cd /root/repo; O=/root/repo/gpurun_out/r4c31; mkdir -p $O
timeout 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "gpu suite rc=$rc"; tail -3 $O/pytest_gpu.log | cut -c1-200
if [ $rc -ne 0 ]; then grep -E "Error|assert|error|FAILED" $O/pytest_gpu.log | head -20 | cut -c1-220; fi
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_stats
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o s -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> /tmp/prof_stats.err
find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
cd /root/repo
timeout 700 python bench.py > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench_final.json")); r = d["roofline"]
print(round(d["value"], 1), round(d["ms_per_step"], 2), d["config"].get("sigma_I"), r["bound"], round(r["frac"], 3), r.get("traffic"), r.get("hbm_frac_measured_bytes"), d["cpu_baseline"]["value"], r["end_to_end"]["time_in_mfma_kernels"])
PY
