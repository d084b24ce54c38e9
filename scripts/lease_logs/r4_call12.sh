#!/bin/bash
# round 4, GPU call 12: chunk-size check of the headline step (32 768-ray chunks) and the secondary bench lines on this tree
cd /root/repo; O=/root/repo/gpurun_out/r4c12; mkdir -p $O
timeout 400 python bench.py --chunk 32768 --no-cpu-baseline > $O/bench_chunk32k.json 2> $O/bench_chunk32k.err; echo "chunk32k rc=$?"
timeout 300 python bench.py --mode c3 --steps 40 --warmup 10 --no-cpu-baseline --sync-debug $O/c3_sync_sites.txt > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"
timeout 300 python bench.py --mode render --no-cpu-baseline > $O/bench_render.json 2> $O/bench_render.err; echo "render rc=$?"
timeout 400 python bench.py --two-hands --chunk 16384 --no-cpu-baseline > $O/bench_twohands.json 2> $O/bench_twohands.err; echo "twohands rc=$?"
timeout 300 python bench.py --mode c5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?"
python - <<PY
import json
for f in ("bench_chunk32k", "bench_c3", "bench_render", "bench_twohands", "bench_c5"):
    try:
        d = json.load(open("$O/" + f + ".json"))
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, round(d["value"], 1), round(d["ms_per_step"], 2), d["config"].get("sigma_I"), d["roofline"]["end_to_end"]["time_in_mfma_kernels"], d["config"].get("pose_refine"))
PY
head -12 $O/c3_sync_sites.txt
