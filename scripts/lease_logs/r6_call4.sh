#!/bin/bash
# round 6, GPU call 4: tests/test_dropin_gpu.py after the BARF fix in inference_step + the inference-related tests
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r6c4; mkdir -p $O
timeout 1500 python -m pytest tests/test_dropin_gpu.py tests/test_train_targets_gpu.py -m gpu -q -s -k "dropin or inference" > $O/pytest_gpu.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/pytest_gpu.log | cut -c1-200
grep -E "three reference steps|frame after three" $O/pytest_gpu.log | cut -c1-400
if [ $rc -ne 0 ]; then grep -E "^E  |FAILED" $O/pytest_gpu.log | head -40 | cut -c1-300; fi
