"""Record one steady-state training-mode HOLDNet output on the MI355X (loss targets included) together with the HIP
Loss evaluated on it -> npz.  Copied to tests/golden/hip_train_output.npz, it is what tests/test_dropin_cpu.py feeds to
the REFERENCE's own Loss.forward in the build container (the reference tree does not exist on the GPU box).

    python scripts/record_hip_outputs.py gpurun_out/hip_train_output.npz
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(path):
    from hold_amd import meshing as M
    from hold_amd.loss import Loss
    from parity_common import hip_input, hip_net, oracle_input, setup

    torch.manual_seed(11)
    sc, sd_np, sd, osc = setup()
    net = hip_net(sc, sd_np, train=True)
    r = 0.08  # a small sphere inside the object's SDF blob: central rays come within 0.05 of it, outer rays do not
    net.nodes["object"].update_cano(M.generate_mesh(lambda x: {"sdf": x.norm(dim=1) - r},
                                                    np.array([[-r, -r, -r], [r, r, r]]), res_init=24, res_up=0))
    b, _ = oracle_input(sc, sd, [0, 2], 8, 8)
    step, epoch = 400, 25
    inp = hip_input(b, net, epoch=epoch, step=step)
    out = net(inp)
    ld = Loss()(inp, out)
    rec = {"step": step, "epoch": epoch}
    for k, v in out.items():
        if torch.is_tensor(v):
            rec["out." + k] = v.detach().cpu().numpy()
    for k in ("gt.rgb", "gt.mask", "idx"):
        rec["batch." + k] = b[k]
    for k, v in ld.items():
        rec["loss." + k] = np.float64(float(v))
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    np.savez_compressed(path, **rec)
    print("wrote", path, {k: float(v) for k, v in ld.items()})


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "hip_train_output.npz"))
