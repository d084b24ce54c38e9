#!/bin/bash
# round 4, GPU call 29: the secondary bench lines on the final tree, and the one-rank RCCL run of the ray-tile mode
cd /root/repo; O=/root/repo/gpurun_out/r4c29; mkdir -p $O
timeout 400 python bench.py --fp32-mfma --no-cpu-baseline > $O/bench_fp32_mfma.json 2> $O/bench_fp32.err; echo "fp32 rc=$?"
timeout 300 python bench.py --mode render --no-cpu-baseline > $O/bench_render.json 2> $O/bench_render.err; echo "render rc=$?"
timeout 400 python bench.py --two-hands --chunk 16384 --no-cpu-baseline > $O/bench_twohands.json 2> $O/bench_twohands.err; echo "twohands rc=$?"
timeout 300 python bench.py --mode c3 --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"
timeout 300 python bench.py --mode c5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?"
HOLD_FORCE_DIST=1 timeout 300 python bench.py --gpus 1 --split rays --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_forcedist_splitrays.json 2> $O/bench_forcedist.err; echo "force-dist split-rays rc=$?"
HOLD_FORCE_DIST=1 timeout 300 python bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_forcedist.json 2> $O/bench_forcedist2.err; echo "force-dist rc=$?"
python - <<PY
import json
for f in ("bench_fp32_mfma", "bench_render", "bench_twohands", "bench_c3", "bench_c5", "bench_forcedist_splitrays", "bench_forcedist"):
    try:
        d = json.load(open("$O/" + f + ".json"))
    except Exception as e:
        print(f, "unreadable", e); continue
    c = d["config"]
    print(f, round(d["value"], 1), round(d["ms_per_step"], 2), c.get("sigma_I"), d["scaling"], c.get("rccl_ranks"), c.get("collective_backend"), (c.get("pose_refine") or {}).get("iters_per_s") if isinstance(c.get("pose_refine"), dict) else None)
PY
