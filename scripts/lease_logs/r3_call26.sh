#!/bin/bash
# round 3, GPU call 26: the launch forms the driver uses for N > 1, exercised with one rank (RCCL process group up,
# all-reduce of the flat bucket, ray-tile split with synchronised sampler rounds)
cd /root/repo; O=/root/repo/gpurun_out/r3c26; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 1 --warmup 1 --no-cpu-baseline > $O/torchrun_weak.json 2> $O/torchrun_weak.err; echo "weak rc=$?"
HOLD_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 1 --warmup 1 --no-cpu-baseline --split rays > $O/torchrun_rays.json 2> $O/torchrun_rays.err; echo "rays rc=$?"
python - <<PY
import json
for f in ("torchrun_weak", "torchrun_rays"):
    try:
        d = json.load(open("$O/" + f + ".json")); print(f, round(d["value"]), d["n_gpus"], d["scaling"], d["config"]["parallelism"])
    except Exception as e:
        print(f, "unreadable", e); print(open("$O/" + f + ".err").read()[-1500:])
PY
