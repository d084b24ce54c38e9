#!/bin/bash
# round 3, GPU call 33: chunked-vs-unchunked Loss test (round-2 advisor request)
cd /root/repo; O=/root/repo/gpurun_out/r3c33; mkdir -p $O
timeout 300 python -m pytest tests/test_train_targets_gpu.py -q -k "chunked_loss_terms" > $O/pytest.log 2>&1; echo "rc=$?"; tail -25 $O/pytest.log | cut -c1-220
