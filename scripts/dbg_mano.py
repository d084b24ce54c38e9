import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from parity_common import *
g = dict(np.load(os.path.join(ROOT, "tests/golden/mano.npz")))
sc, sd_np, sd, osc = setup()
net = hip_net(sc, sd_np)
node = net.nodes["right"]
idx = torch.arange(sc["n_frames"], device="cuda")
p = node.params(idx)
so = node.server(torch.full((sc["n_frames"],), sc["scene_scale"], device="cuda"), p["right.transl"], p["right.full_pose"], p["right.betas"])
for k in ["verts", "jnts", "tfs", "v_posed"]:
    d = np.abs(so[k].detach().cpu().numpy() - g[k])
    print(k, "max abs", d.max(), "argmax", np.unravel_index(d.argmax(), d.shape))
print(so["tfs"][0, 2].detach().cpu().numpy()); print(g["tfs"][0, 2]); print(node.server.tfs_c_inv[2].cpu().numpy(), node.server.tfs_c_inv.is_contiguous(), node.server.tfs_c_inv.stride())
