#!/bin/bash
# round 5, GPU call 19: where the device idles inside a C3 step (torch profiler trace -> scripts/gap_report.py)
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r5c19; mkdir -p $O
timeout 300 python bench.py --mode c3 --steps 4 --warmup 3 --no-cpu-baseline --no-refine --torch-profile /tmp/c3prof.txt > $O/bench.json 2> $O/bench.err; echo "rc=$?"
python scripts/gap_report.py /tmp/c3prof.txt.trace.json 4 > $O/c3_gap_report.txt 2>&1
cp /tmp/c3prof.txt $O/c3_torch_profile.txt; ls -la /tmp/c3prof.txt.trace.json
head -80 $O/c3_gap_report.txt | cut -c1-200
