#!/bin/bash
# round 6, GPU call 27: non-temporal hints on the streaming side of the register-resident f16x3 kernels -- result stores (nt) and, in a second
# variant, the side-tile / input-fragment LDS-DMA loads too; weights stay cached.  Micro-benchmarks and the headline, three builds alternating
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r6c27; mkdir -p $O
for v in base ntst ntall; do
  lib=/root/repo/hold_amd/libholdhip.so; [ $v != base ] && lib=/root/repo/hold_amd/libholdhip_$v.so
  echo "--- $v: rgemm"; HOLD_LIB=$lib timeout 300 python scripts/bench_rgemm.py > $O/rgemm_$v.log 2>&1; grep gemm_h3 $O/rgemm_$v.log | cut -c1-120
  echo "--- $v: sweeps"; HOLD_LIB=$lib timeout 300 python scripts/bench_chain.py > $O/chain_$v.log 2>&1; grep -i "h3" $O/chain_$v.log | cut -c1-200 | head -8
done
run() { name=$1; shift; envs=(); while [[ "$1" == *=* ]]; do envs+=("$1"); shift; done; env "${envs[@]}" timeout 400 python bench.py --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python -c "
import json
try:
    d = json.load(open('$O/bench_$name.json')); k = d['roofline']['kernels']; print('$name', round(d['ms_per_step'], 2), 'ms/step', round(d['value'], 1), 'rays/s', d['config'].get('sigma_I'), {n: round(k[n]['avg_launch_ms'], 3) for n in ('rgemm_h3_kernel', 'rchain_h3_kernel', 'rchain_a2_h3_kernel', 'rchain_dbwd_h3_kernel', 'trunk_r6_kernel', 'wgrad_h3_kernel') if n in k})
except Exception as e: print('$name no line', e)
"; }
for i in 1 2; do
  run base_$i X=1 --steps 4 --warmup 2
  run ntst_$i HOLD_LIB=/root/repo/hold_amd/libholdhip_ntst.so --steps 4 --warmup 2
  run ntall_$i HOLD_LIB=/root/repo/hold_amd/libholdhip_ntall.so --steps 4 --warmup 2
done
