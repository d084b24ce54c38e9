#!/bin/bash
# round 5, GPU call 22: host-side micro costs (views, stream lookup, pinned uploads) in the main and the autograd thread
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r5c22; mkdir -p $O
timeout 200 python scripts/probes/host_costs.py 2>&1 | tee $O/host_costs.txt
