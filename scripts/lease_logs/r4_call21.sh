#!/bin/bash
# round 4, GPU call 21: the final tree -- full GPU suite, build() + smoke(), rocprofv3 kernel stats + PMC traffic of the
# default bench command, the default line with its cpu_baseline leg, C3
cd /root/repo; O=/root/repo/gpurun_out/r4c21; mkdir -p $O
timeout 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "gpu suite rc=$rc"; tail -3 $O/pytest_gpu.log | cut -c1-200
if [ $rc -ne 0 ]; then grep -E "Error|assert|error|FAILED" $O/pytest_gpu.log | head -20 | cut -c1-220; fi
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log | cut -c1-200
bash scripts/prof_r04.sh > $O/prof.log 2>&1; echo "prof rc=$?"; tail -3 $O/prof.log | cut -c1-200
python scripts/make_pmc_json.py > $O/pmc.log 2>&1; tail -2 $O/pmc.log | cut -c1-300
cp profiles/r04_pmc_traffic.json profiles/r04_pmc_FETCH_SIZE.json profiles/r04_pmc_WRITE_SIZE.json $O/ 2>/dev/null
timeout 700 python bench.py > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc=$?"
timeout 300 python bench.py --mode c3 --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"
python - <<PY
import json
for f in ("bench_final", "bench_c3"):
    try:
        d = json.load(open("$O/" + f + ".json"))
    except Exception as e:
        print(f, "unreadable", e); continue
    r = d["roofline"]
    print(f, round(d["value"], 1), round(d["ms_per_step"], 2), d["config"].get("sigma_I"), r["bound"], round(r["frac"], 3), r["kernel"][:30], r.get("traffic"), d.get("cpu_baseline"))
    for k, v in r["kernels"].items():
        if "mfma_frac" in v:
            print(f"  {k:22s} share {v['time_share']:.3f} TF-eq {v['fp32_equivalent_tflops']:.1f} mfma {v['mfma_frac']:.3f} hbm {v.get('hbm_frac')} meas {v.get('hbm_frac_measured_bytes')} bound {v.get('bound')}")
PY
