#!/bin/bash
# round 4, GPU call 28: smoke()'s "grad norm sum" differed between two calls (9.12e-2 / 8.46e-2, same loss): is its training step
# reproducible run to run with the RNG seeded, and how much does the sum move with the seed?
cd /root/repo; O=/root/repo/gpurun_out/r4c28; mkdir -p $O
timeout 300 python scripts/dbg_smoke_grads.py 2>&1 | grep -v Warning | tail -8 | cut -c1-900 | tee $O/dbg.log
