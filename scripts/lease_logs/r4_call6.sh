#!/bin/bash
# round 4, GPU call 6: the tests that failed in calls 4 / 5 after their fixes (composite keys held to the reference on colour
# only, eikonal gate before the tile share), then the evidence passes for profiles/: rocprofv3 kernel stats of the default
# bench command and the two --pmc passes (FETCH_SIZE, WRITE_SIZE)
cd /root/repo; O=/root/repo/gpurun_out/r4c6; mkdir -p $O
timeout 600 python -m pytest tests/test_path_gpu.py tests/test_train_targets_gpu.py -q -k "golden or ray_tile or chunk" > $O/pytest_sel.log 2>&1; echo "selected tests rc=$?"; tail -4 $O/pytest_sel.log | cut -c1-220
bash scripts/prof_r04.sh > $O/prof.log 2>&1; echo "prof rc=$?"; tail -12 $O/prof.log | cut -c1-240
