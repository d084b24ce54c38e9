#!/bin/bash
# round 3, GPU call 16: torch operators of the 1 280-ray training step by issuing source line
cd /root/repo; O=/root/repo/gpurun_out/r3c16; mkdir -p $O
timeout 300 python bench.py --mode c3 --steps 1 --warmup 2 --no-cpu-baseline --no-refine --op-sites $O/c3_sites.txt > /dev/null 2> $O/err.log; echo "rc=$?"
head -150 $O/c3_sites.txt | cut -c1-230; tail -3 $O/err.log
