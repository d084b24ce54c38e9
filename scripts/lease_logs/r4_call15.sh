#!/bin/bash
# round 4, GPU call 15: evidence for profiles/ on the final kernel tree: rocprofv3 kernel stats + PMC traffic of the default
# bench command, the default line WITH its cpu_baseline leg, and the secondary lines
cd /root/repo; O=/root/repo/gpurun_out/r4c15; mkdir -p $O
bash scripts/prof_r04.sh > $O/prof.log 2>&1; echo "prof rc=$?"; tail -4 $O/prof.log | cut -c1-200
python scripts/make_pmc_json.py > $O/pmc.log 2>&1; tail -2 $O/pmc.log | cut -c1-300
timeout 700 python bench.py > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc=$?"
timeout 400 python bench.py --fp32-mfma --no-cpu-baseline > $O/bench_fp32_mfma.json 2> $O/bench_fp32.err; echo "fp32 rc=$?"
timeout 300 python bench.py --mode render --no-cpu-baseline > $O/bench_render.json 2> $O/bench_render.err; echo "render rc=$?"
timeout 400 python bench.py --two-hands --chunk 16384 --no-cpu-baseline > $O/bench_twohands.json 2> $O/bench_twohands.err; echo "twohands rc=$?"
timeout 300 python bench.py --mode c3 --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"
timeout 300 python bench.py --mode c5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?"
cp profiles/r04_pmc_traffic.json $O/ 2>/dev/null
python - <<PY
import json
for f in ("bench_final", "bench_fp32_mfma", "bench_render", "bench_twohands", "bench_c3", "bench_c5"):
    try:
        d = json.load(open("$O/" + f + ".json"))
    except Exception as e:
        print(f, "unreadable", e); continue
    r = d["roofline"]
    print(f, round(d["value"], 1), round(d["ms_per_step"], 2), d["config"].get("sigma_I"), r["bound"], round(r["frac"], 3), r["kernel"][:30], d.get("cpu_baseline", {}).get("value"), d["config"].get("pose_refine", {}).get("iters_per_s") if isinstance(d["config"].get("pose_refine"), dict) else None)
PY
