#!/bin/bash
# round 6, GPU call 16: stress probe of hold_gemm_h3's rare bit-irreproducibility
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r6c16; mkdir -p $O
timeout 600 python scripts/probes/gemm_h3_flake.py 3000 > $O/flake.log 2>&1; tail -25 $O/flake.log | cut -c1-300
