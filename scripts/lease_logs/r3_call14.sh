#!/bin/bash
# round 3, GPU call 14: host synchronisation sites of the 1 280-ray training step
cd /root/repo; O=/root/repo/gpurun_out/r3c14; mkdir -p $O
timeout 300 python bench.py --mode c3 --steps 2 --warmup 3 --no-cpu-baseline --no-refine --sync-debug $O/c3_syncs.txt > /dev/null 2> $O/err.log; echo "rc=$?"
cat $O/c3_syncs.txt; tail -5 $O/err.log
