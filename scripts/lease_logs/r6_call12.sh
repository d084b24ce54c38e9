#!/bin/bash
# round 6, GPU call 12: hold_gemm_h3 with the row maxima loaded a block ahead (instead of the 4-byte LDS-DMA): bit-reproducibility,
# the full-frame invariants; then the reference's 1 280-ray step (C3) A/B over the round's routes, alternating, same box
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r6c12; mkdir -p $O
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_scale_gpu.py -m gpu -q -x -k "gemm_h3 or full_frame or scale" > $O/pytest.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/pytest.log | cut -c1-200
if [ $rc -ne 0 ]; then grep -E "^E  |FAILED|Error" $O/pytest.log | head -30 | cut -c1-300; fi
c3() { name=$1; shift; env "$@" timeout 300 python bench.py --mode c3 --steps 30 --warmup 5 --no-cpu-baseline --no-refine > $O/c3_$name.json 2> $O/c3_$name.err; python -c "
import json
try:
    d = json.load(open('$O/c3_$name.json')); print('$name', round(d['ms_per_step'], 2), 'ms/step', round(d['roofline']['end_to_end']['time_in_mfma_kernels'], 3), d['config'].get('c_abi_calls_per_step'))
except Exception as e: print('$name no line', e)
"; }
for i in 1 2 3; do
  c3 default_$i X=1
  c3 no_h3_gemm_$i HOLD_H3_GEMM=0
  c3 r5_routes_$i HOLD_H3_GEMM=0 HOLD_H3_BWD=0
done
