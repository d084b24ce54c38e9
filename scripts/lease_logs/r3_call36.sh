#!/bin/bash
# round 3, GPU call 36: the corrected Loss chunk test
cd /root/repo; O=/root/repo/gpurun_out/r3c36; mkdir -p $O
timeout 300 python -m pytest tests/test_train_targets_gpu.py -q > $O/pytest.log 2>&1; echo "rc=$?"; tail -3 $O/pytest.log | cut -c1-200
