#!/bin/bash
# round 5, GPU call 21: the host's Python time of a C3 step by function (cProfile, main + autograd thread)
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r5c21; mkdir -p $O
timeout 300 python bench.py --mode c3 --steps 8 --warmup 4 --no-cpu-baseline --no-refine --no-profile --cprofile $O/c3_cprofile.txt > $O/bench.json 2> $O/bench.err; echo "rc=$?"
tail -3 $O/bench.err
