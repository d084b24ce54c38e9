#!/bin/bash
# round 4, GPU call 34: the tests that run the background net end to end besides test_path_gpu (call 33), and smoke(), on the final tree
cd /root/repo; O=/root/repo/gpurun_out/r4c34; mkdir -p $O
timeout 170 python -m pytest tests/test_train_targets_gpu.py tests/test_scale_gpu.py -q -x > $O/pytest_sel.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/pytest_sel.log | cut -c1-220
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log | cut -c1-200
