#!/bin/bash
# round 5, GPU call 18+: torch profiler of 4 C3 steps (device time, host time per operator)
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r5c18; mkdir -p $O
timeout 300 python bench.py --mode c3 --steps 4 --warmup 3 --no-cpu-baseline --no-refine --torch-profile $O/c3_torch_profile.txt > $O/bench.json 2> $O/bench.err; echo "rc=$?"
grep -n -E "nonzero|index_put|aten::index |masked|unique|randperm|sort" $O/c3_torch_profile.txt | cut -c1-200 | head -30
