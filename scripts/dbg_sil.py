import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_fitting_gpu import _scene_verts
from hold_amd import fitting as ft
from oracle import fitting_oracle as fo
net, verts, faces = _scene_verts(1)
H = W = 48; fx = fy = 220.0; cx = cy = 24.0; sigma = 1e-4; blur = np.log(1.0 / 1e-4 - 1.0) * sigma
v = verts.clone().requires_grad_(True)
vs, fs = ft.seal_mano_mesh(v, faces, True)
m = ft.soft_silhouette(vs, fs, fx, fy, cx, cy, H, W, sigma, blur)
print("mask nan", torch.isnan(m).sum().item(), "mean", m.mean().item(), "verts nan", torch.isnan(verts).sum().item(), verts[..., 2].min().item())
wgt = torch.rand(1, H, W, device="cuda")
(m * wgt).sum().backward()
print("grad nan", torch.isnan(v.grad).sum().item(), "absmax", v.grad[~torch.isnan(v.grad)].abs().max().item())
vo, fo_ = fo.seal_mano_mesh(verts.cpu(), faces.cpu(), True)
ref = fo.soft_silhouette(vo, fo_, fx, fy, cx, cy, H, W, sigma, blur)
print("fwd max diff", (m.detach().cpu() - ref).abs().max().item(), ref.mean().item())
