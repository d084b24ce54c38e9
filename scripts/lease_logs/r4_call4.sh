#!/bin/bash
# round 4, GPU call 4: the full GPU suite on the tree with the register-resident sweeps routed in (DSP+a2, DBWD), the
# conditioning-aware sampler bound, the reference goldens of the two-hand / C1 / C5 configurations and both arithmetics end to end
cd /root/repo; O=/root/repo/gpurun_out/r4c4; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "suite rc=$?"; tail -15 $O/pytest_gpu.log | cut -c1-250
