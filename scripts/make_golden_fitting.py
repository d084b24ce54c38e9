"""Pins oracle/fitting_oracle.py:loss_fn_h / loss_fn_ih against the REFERENCE's own code/src/fitting/loss.py (imported here
under oracle/ref_shim with the pytorch3d.renderer names stubbed: the losses only use knn_points, l1_loss and
project2d_batch) and writes tests/golden/fitting_losses.npz (inputs + reference outputs + reference gradients).
Runs only in the build container (needs /root/reference)."""
import sys, os, types, pickle, tempfile
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import ref_shim
ref_shim.install()
# extra stubs for src.fitting.utils' imports
def stub(name, attrs=()):
    m = sys.modules.get(name) or types.ModuleType(name)
    for a in attrs:
        if not hasattr(m, a): setattr(m, a, type(a, (), {}))
    sys.modules[name] = m
    return m
stub("pytorch3d.renderer", ["BlendParams","MeshRasterizer","MeshRenderer","PerspectiveCameras","RasterizationSettings","SoftSilhouetteShader","TexturesVertex"])
stub("pytorch3d.structures", ["Meshes"])
stub("pytorch3d.renderer.mesh", []); stub("pytorch3d.renderer.mesh.shader", ["SoftSilhouetteShader"])
for n in ("PIL", "PIL.Image", "matplotlib", "matplotlib.pyplot", "imageio", "cv2", "tqdm"):
    if n not in sys.modules:
        try: __import__(n)
        except Exception: stub(n)
d = tempfile.mkdtemp(); os.makedirs(d + "/body_models")
idx = {"contact_zones": {0: list(range(700, 720)), 1: list(range(740, 760))}}
pickle.dump(idx, open(d + "/body_models/contact_zones.pkl", "wb"))
os.chdir(d)
try:
    import src.fitting.loss as RL
except Exception as e:
    import traceback; traceback.print_exc(); sys.exit(1)
from oracle import fitting_oracle as fo
g = torch.Generator().manual_seed(0)
B = 3
def mk():
    out = {"right.v3d_c": torch.randn(B, 778, 3, generator=g) * 0.1 + torch.tensor([0.0, 0, 1.0]),
           "left.v3d_c": torch.randn(B, 778, 3, generator=g) * 0.1 + torch.tensor([3.0, 0, 1.5]),
           "object.v3d_c": torch.randn(B, 500, 3, generator=g) * 0.1 + torch.tensor([0.0, 0, 1.0]),
           "object.mask": torch.rand(B, 20, 20, generator=g),
           "K": torch.tensor([[200.0, 0, 10], [0, 210.0, 12], [0, 0, 1]])[None].repeat(B, 1, 1)}
    tg = {"right": (torch.rand(B, 20, 20, generator=g) > 0.7).float(), "left": (torch.rand(B, 20, 20, generator=g) > 0.7).float(),
          "object": (torch.rand(B, 20, 20, generator=g) > 0.5).float()}
    return out, tg
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
gold = {}
out, tg = mk()
gold.update({"in." + k: v.numpy().copy() for k, v in out.items()})
gold.update({"tg." + k: v.numpy().copy() for k, v in tg.items()})
gold["contact_idx"] = np.asarray(RL.contact_idx)
tg2 = {k: v.clone() for k, v in tg.items()}
cidx = torch.as_tensor(RL.contact_idx)
for it in range(2):   # second call: cached 2-D targets, perturbed vertices
    o1 = {k: (v.clone().requires_grad_(True) if k != "K" else v) for k, v in out.items()}
    o2 = {k: (v.clone().requires_grad_(True) if k != "K" else v) for k, v in out.items()}
    a = RL.loss_fn_ih(o1, tg); b = fo.loss_fn_ih(o2, tg2, cidx)
    for k in a:
        print(it, k, float(a[k]), float(b[k]))
        assert abs(float(a[k]) - float(b[k])) <= 1e-6 * max(1.0, abs(float(a[k])))
        gold[f"ih{it}.{k}"] = np.float32(float(a[k]))
    a["loss"].backward(); b["loss"].backward()
    for k in ("right.v3d_c", "left.v3d_c", "object.v3d_c", "object.mask"):
        print("  grad", k, float((o1[k].grad - o2[k].grad).abs().max()), float(o1[k].grad.abs().max()))
        assert float((o1[k].grad - o2[k].grad).abs().max()) <= 1e-6 * max(1.0, float(o1[k].grad.abs().max()))
        gold[f"ih{it}.grad.{k}"] = o1[k].grad.numpy().copy()
    gold[f"ih{it}.in"] = np.stack([out[k].numpy() for k in ("right.v3d_c", "left.v3d_c")])
    gold[f"ih{it}.in_obj"] = out["object.v3d_c"].numpy().copy()
    out = {k: (v + 0.01 * torch.randn(v.shape, generator=g) if k.endswith("v3d_c") else v) for k, v in out.items()}

# single-hand loss (loss_fn_rh) on the same inputs
o1 = {k: (torch.as_tensor(gold["in." + k]).clone().requires_grad_(True) if k != "K" else torch.as_tensor(gold["in." + k])) for k in ("right.v3d_c", "object.v3d_c", "object.mask", "K")}
o1["right.mask"] = torch.rand(B, 20, 20, generator=g).requires_grad_(True)
gold["in.right.mask"] = o1["right.mask"].detach().numpy().copy()
o2 = {k: (v.detach().clone().requires_grad_(True) if k != "K" else v) for k, v in o1.items()}
tgh = {k: torch.as_tensor(gold["tg." + k]) for k in ("right", "object")}
a = RL.loss_fn_rh(o1, tgh); b = fo.loss_fn_h(o2, tgh, "right", cidx)
for k in a:
    print("rh", k, float(a[k]), float(b[k]))
    assert abs(float(a[k]) - float(b[k])) <= 1e-6 * max(1.0, abs(float(a[k])))
    gold[f"rh.{k}"] = np.float32(float(a[k]))
a["loss"].backward(); b["loss"].backward()
for k in ("right.v3d_c", "object.v3d_c", "object.mask", "right.mask"):
    assert float((o1[k].grad - o2[k].grad).abs().max()) <= 1e-6 * max(1.0, float(o1[k].grad.abs().max()))
    gold[f"rh.grad.{k}"] = o1[k].grad.numpy().copy()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "fitting_losses.npz"), **gold)
print("wrote tests/golden/fitting_losses.npz", sum(v.nbytes for v in gold.values()) // 1024, "KiB")
