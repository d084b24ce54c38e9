#!/bin/bash
# round 5, GPU call 20: C3 step with and without the per-kernel HIP events of the roofline breakdown (same box)
ulimit -c 0
cd /root/repo; O=/root/repo/gpurun_out/r5c20; mkdir -p $O
for v in prof noprof prof noprof; do
  fl=""; [ $v = noprof ] && fl="--no-profile"
  timeout 300 python bench.py --mode c3 --steps 40 --warmup 8 --no-cpu-baseline --no-refine $fl 2> $O/err_$v.txt | python -c "import json,sys; d=json.load(sys.stdin); print('$v', round(d['ms_per_step'],2), 'ms/step', round(d['value']), 'rays/s')"
done
