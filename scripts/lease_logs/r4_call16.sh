#!/bin/bash
# round 4, GPU call 16: is the 1 280-ray step (C3) bound by the GPU or by the host's launch rate?  rocprofv3 kernel stats of
# --mode c3 (sum of kernel durations against the wall clock of the timed steps); plus the RCCL path with one rank in
# --split rays mode (HOLD_FORCE_DIST=1: sampler round exchange, Loss count exchange, grad_mul = world)
cd /root/repo; O=/root/repo/gpurun_out/r4c16; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/c3prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c3prof -o s -- python /root/repo/bench.py --mode c3 --steps 40 --warmup 10 --no-cpu-baseline --no-refine > $O/c3_under_rocprof.json 2> /tmp/c3prof.err
find /tmp/c3prof -name "*kernel_stats.csv" -exec cp {} $O/c3_kernel_stats.csv \;
cd /root/repo
python - <<PY
import csv, json
rows = list(csv.DictReader(open("$O/c3_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
d = json.load(open("$O/c3_under_rocprof.json"))
print("c3 under rocprof: ms/step", round(d["ms_per_step"], 2), "; kernel time per step (50 steps incl. warm-up):", round(tot / 50 / 1e6, 2), "ms; launches per step", calls // 50)
mf = ("rmlp", "rsweep", "wgrad_r6", "wgrad_lds", "rgemm", "gemm_nt", "chain_x6")
m = sum(float(r["TotalDurationNs"]) for r in rows if any(x in r["Name"] for x in mf))
print("  MFMA kernels", round(m / 50 / 1e6, 2), "ms/step; others", round((tot - m) / 50 / 1e6, 2), "ms/step")
for r in [r for r in rows if not any(x in r["Name"] for x in mf)][:14]:
    print("   ", r["Calls"], round(float(r["TotalDurationNs"]) / 50 / 1e6, 3), "ms/step", r["Name"][:90])
PY
HOLD_FORCE_DIST=1 timeout 300 python bench.py --gpus 1 --split rays --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_forcedist_splitrays.json 2> $O/bench_forcedist.err; echo "force-dist split-rays rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench_forcedist_splitrays.json"))
print(round(d["value"], 1), d["scaling"], d["config"]["rccl_ranks"], d["config"]["collective_backend"], d["config"]["per_rank_ms_per_step"], d["config"]["parallelism"][:60])
PY
