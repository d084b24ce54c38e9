"""Time the REFERENCE's own training step on this machine's CPU cores (build container only: /root/reference does not travel, so
bench.py's cpu_baseline leg on the GPU box times the oracle port; this is the number SURVEY 8(d) asked for, taken where the
reference exists): src.hold.hold.HOLD.training_step (the reference's HOLDNet + Loss) + backward + clip + its Adam, at the
reference's batch -- 10 frames x 128 random pixels = 1 280 rays -- before the first canonical mesh exists (steps < 200: no kaolin
loss targets, which the CPU shim could only stub).  Prints one JSON line; profiles/r06_reference_cpu_step.json keeps it.

    python scripts/time_reference_cpu.py [steps] [threads]
"""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from hold_amd import synthetic as syn  # noqa: E402
from oracle import ref_shim  # noqa: E402
from make_golden_hold_steps import ref_batch  # noqa: E402


def main():
    from PIL import Image
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else os.cpu_count()
    torch.set_num_threads(threads)
    ref_shim.install()
    import src.hold.hold as H
    n_frames, W = 10, 64
    sc = syn.make_scene(n_frames=n_frames)
    wd = ref_shim.prepare_workdir(sc)
    opt = ref_shim.load_opt()
    opt.model.scene_bounding_sphere = sc["scene_bounding_sphere"]
    args = ref_shim.make_args(n_images=n_frames)
    torch.manual_seed(1)
    np.random.seed(1)
    with ref_shim.chdir(wd):
        hold = H.HOLD(opt, args)
    hold.model.load_state_dict({k: torch.as_tensor(v) for k, v in syn.make_state_dict(sc, barf_iter=3999).items()}, strict=False)
    hold.log = lambda *a, **k: None
    hold.current_epoch = 0
    optim = hold.configure_optimizers()[0][0]
    png = os.path.join(tempfile.mkdtemp(prefix="hold_png_"), "im.png")
    Image.fromarray(np.zeros((W, W, 3), np.uint8)).save(png)
    uv_all = syn.make_uv(W, W)
    rs = np.random.RandomState(0)
    hold.train()
    times = []
    for k in range(steps + 1):  # one warm-up step
        hold.global_step = 1 + k
        uv = uv_all[rs.choice(len(uv_all), 128, replace=False)]  # 128 random pixels per frame (tempo_dataset.py:27-36)
        b = syn.make_batch(sc, list(range(n_frames)), uv, W, W, seed=1 + k)
        t0 = time.perf_counter()
        with ref_shim.chdir(wd):
            loss = hold.training_step(ref_batch(b, png))
        optim.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(hold.parameters(), 0.5)
        optim.step()
        dt = time.perf_counter() - t0
        if k:
            times.append(dt)
        print(f"step {k}: {dt:.2f} s loss {float(loss):.4f}", file=sys.stderr, flush=True)
    med = float(np.median(times))
    print(json.dumps({"what": "the reference's own HOLD.training_step + backward + clip + Adam on CPU (oracle/ref_shim), 10 frames x 128 rays, "
                              "no loss-target geometry (steps < 200)", "rays_per_step": 1280, "steps": steps, "threads": threads,
                      "cores": os.cpu_count(), "median_step_s": med, "rays_per_s": 1280 / med,
                      "min_step_s": float(min(times)), "max_step_s": float(max(times)), "where": "build container (no GPU)"}))


if __name__ == "__main__":
    main()
