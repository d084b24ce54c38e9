#!/bin/bash
# round 3, GPU call 15: which source lines issue the small torch kernels of the 1 280-ray training step
cd /root/repo; O=/root/repo/gpurun_out/r3c15; mkdir -p $O
timeout 300 python bench.py --mode c3 --steps 2 --warmup 3 --no-cpu-baseline --no-refine --torch-profile $O/c3_ops.txt > /dev/null 2> $O/err.log; echo "rc=$?"
grep -A 160 "by source line" $O/c3_ops.txt | cut -c1-220; tail -3 $O/err.log
