#!/bin/bash
# round 6, GPU call 15: test_full_frame_512_invariants fails inside the full suite only (two renders of one frame differ) -- which route?
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r6c15; mkdir -p $O
t() { name=$1; shift; env "$@" timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_dropin_gpu.py > $O/suite_$name.log 2>&1; echo "$name rc=$? $(tail -1 $O/suite_$name.log | cut -c1-150)"; grep -E "rays that differ" $O/suite_$name.log | head -2 | cut -c1-900; }
t default X=1
t no_h3_gemm HOLD_H3_GEMM=0
t no_h3_bwd HOLD_H3_BWD=0
