"""pose-refinement iteration rate (config C3: B = 10 frames, 300x300 masks, 1554-face hand + ~5000-face object)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch


def run(iters=30, B=10):
    from parity_common import hip_net, setup
    from hold_amd import fitting as ft
    sc, sd_np, sd, osc = setup(n_frames=B)
    net = hip_net(sc, sd_np)
    dev = torch.device("cuda")
    hand, obj = net.nodes["right"], net.nodes["object"]
    idx = torch.arange(B, device=dev)
    hp, op = hand.params(idx), obj.params(idx)
    params = {"scene_scale": torch.tensor([1.0], device=dev), "right.global_orient": hp["right.global_orient"].detach(),
              "right.pose": hp["right.pose"].detach(), "right.betas": hp["right.betas"][:1].detach(),
              "right.transl": hp["right.transl"].detach(), "object.global_orient": op["object.global_orient"].detach(),
              "object.transl": op["object.transl"].detach()}
    w2c = torch.eye(4, device=dev).repeat(B, 1, 1); w2c[:, 2, 3] = 0.6
    K = torch.tensor([[700.0, 0, 150.0], [0, 700.0, 150.0], [0, 0, 1]], device=dev)
    hand_faces = torch.as_tensor(hand.server.faces.astype(np.int64), device=dev)
    nlat, nlon = 50, 52
    th = torch.linspace(0.1, np.pi - 0.1, nlat); ph = torch.linspace(0, 2 * np.pi, nlon + 1)[:-1]
    sv = torch.stack([torch.sin(th)[:, None] * torch.cos(ph)[None], torch.sin(th)[:, None] * torch.sin(ph)[None],
                      torch.cos(th)[:, None].expand(nlat, nlon)], -1).reshape(-1, 3) * 0.06
    obj.server.object_model.v3d_cano = sv.to(dev)
    fl = []
    for a in range(nlat - 1):
        for b in range(nlon):
            i0, i1 = a * nlon + b, a * nlon + (b + 1) % nlon
            fl += [[i0, i1, i0 + nlon], [i1, i1 + nlon, i0 + nlon]]
    obj_faces = torch.tensor(fl, device=dev)
    contact_idx = torch.arange(700, 778, device=dev)
    gt = ft.FittingModel(hand.server, obj.server, hand_faces, obj_faces, params, w2c, K, (300, 300), None, contact_idx)
    with torch.no_grad():
        o = gt.fwd_params()
    targets = {"right": (o["right.mask"] > 0.5).float(), "object": (o["object.mask"] > 0.5).float()}
    p2 = dict(params); p2["right.transl"] = params["right.transl"] + 0.004
    m = ft.FittingModel(hand.server, obj.server, hand_faces, obj_faces, p2, w2c, K, (300, 300), targets, contact_idx)
    m.fit(num_iterations=3)
    torch.cuda.synchronize(); t0 = time.time()
    h = m.fit(num_iterations=iters)
    torch.cuda.synchronize(); dt = time.time() - t0
    return {"iters_per_s": len(h) / dt, "ms_per_iter": dt / len(h) * 1e3, "frames": B, "mask": "300x300",
            "faces_hand": int(hand_faces.shape[0] + 16), "faces_object": int(obj_faces.shape[0]), "loss_first": float(h[0]),
            "loss_last": float(h[-1]), "coverage": [float(targets["right"].mean()), float(targets["object"].mean())]}


if __name__ == "__main__":
    print("pose refinement:", run())
