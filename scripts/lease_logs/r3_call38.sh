#!/bin/bash
# round 3, GPU call 38: whole-dW weight gradient for N in 129..256 and the split lin0 gradient -- full GPU suite on that tree
cd /root/repo; O=/root/repo/gpurun_out/r3c38; mkdir -p $O
timeout 600 python -m pytest tests -q -m gpu > $O/pytest_gpu_x6.log 2>&1; echo "x6 suite rc=$?"; tail -6 $O/pytest_gpu_x6.log | cut -c1-220
