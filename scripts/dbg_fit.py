import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import test_fitting_gpu as T
src = open(T.__file__).read()
# run the body of the fitting test but print instead of assert
from hold_amd import fitting as ft
from parity_common import *
sc, sd_np, sd, osc = setup(n_frames=4)
net = hip_net(sc, sd_np)
B = 3; dev = torch.device("cuda")
hand, obj = net.nodes["right"], net.nodes["object"]
idx = torch.arange(B, device=dev)
hp, op = hand.params(idx), obj.params(idx)
params = {"scene_scale": torch.tensor([1.0], device=dev), "right.global_orient": hp["right.global_orient"].detach(),
          "right.pose": hp["right.pose"].detach(), "right.betas": hp["right.betas"][:1].detach(),
          "right.transl": hp["right.transl"].detach(), "object.global_orient": op["object.global_orient"].detach(),
          "object.transl": op["object.transl"].detach()}
w2c = torch.eye(4, device=dev).repeat(B, 1, 1); w2c[:, 2, 3] = 0.9
K = torch.tensor([[260.0, 0, 40.0], [0, 260.0, 40.0], [0, 0, 1]], device=dev)
hand_faces = torch.as_tensor(hand.server.faces.astype(np.int64), device=dev)
nv = obj.server.object_model.v3d_cano.shape[0]
obj_faces = torch.arange(0, nv - nv % 3, device=dev).view(-1, 3)
contact_idx = torch.arange(700, 778, device=dev)
gt = ft.FittingModel(hand.server, obj.server, hand_faces, obj_faces, params, w2c, K, (80, 80), None, contact_idx)
o = gt.fwd_params()
for k, v in o.items(): print(k, tuple(v.shape), "nan", torch.isnan(v).sum().item(), "mean", v.float().mean().item(), "min", v.min().item(), "max", v.max().item())
