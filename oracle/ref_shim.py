"""TEST INFRASTRUCTURE ONLY -- imports the *reference* (zc-alexfan/hold, read-only at
/root/reference) on CPU so that (a) the oracle restatement in ``oracle/hold_oracle.py`` can
be pinned against the real thing and (b) golden fixtures under ``tests/golden`` can be
generated (``scripts/make_golden.py``).  /root/reference does not exist on the GPU box:
nothing in ``-m gpu`` tests, ``smoke()`` or ``bench.py`` imports this file.

What the shim does:
* ``.cuda()`` becomes a no-op, ``torch.device("cuda")`` inside get_embedder -> cpu
* stub modules for packages that are absent here and are only imported at module top
  (easydict, loguru, kaolin, pytorch3d.ops.knn_points, trimesh, skimage, cv2, smplx ...)
* ``pytorch3d.ops.knn_points`` restated as cdist^2 + topk (pytorch3d 0.7.4 semantics:
  squared L2, ascending)
* a temp working dir with synthetic ``body_models/MANO_*.pkl`` and ``data/<case>/build/data.npy``
"""
from __future__ import annotations

import os
import pickle
import sys
import tempfile
import types

import numpy as np
import torch

REF_ROOT = "/root/reference"
_INSTALLED = False


class EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            v = EasyDict(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _knn_points(p1, p2, K=1, return_nn=False, **kw):
    # exact (x-y)^2 sums like the pytorch3d kernel (torch.cdist's matmul trick is less accurate)
    ds, ids = [], []
    for c in range(0, p1.shape[1], 4096):
        d = ((p1[:, c:c + 4096, None, :] - p2[:, None, :, :]) ** 2).sum(-1)
        dd, ii = torch.topk(d, K, dim=-1, largest=False, sorted=True)
        ds.append(dd)
        ids.append(ii)
    dists, idx = torch.cat(ds, 1), torch.cat(ids, 1)
    nn = None
    if return_nn:
        nn = torch.gather(p2[:, None].expand(-1, p1.shape[1], -1, -1), 2, idx[..., None].expand(-1, -1, -1, 3))
    return dists, idx, nn


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install():
    """Make ``import src....`` resolve to the reference on CPU. Idempotent."""
    global _INSTALLED
    if _INSTALLED:
        return
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError("reference tree not present (expected on the build container only)")
    sys.dont_write_bytecode = True
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self

    _stub("easydict", EasyDict=EasyDict)

    class _Logger:
        def __getattr__(self, k):
            return lambda *a, **kw: None

    _stub("loguru", logger=_Logger())
    kaolin = _stub("kaolin")
    kops = _stub("kaolin.ops")
    kmesh = _stub("kaolin.ops.mesh", index_vertices_by_faces=lambda v, f: v[:, f])
    kaolin.ops = kops
    kops.mesh = kmesh
    kmet = _stub("kaolin.metrics")
    ktm = _stub("kaolin.metrics.trianglemesh")
    kaolin.metrics = kmet
    kmet.trianglemesh = ktm
    p3d = _stub("pytorch3d")
    p3dops = _stub("pytorch3d.ops", knn_points=_knn_points)
    p3d.ops = p3dops
    for name in ["trimesh", "cv2", "comet_ml", "pytorch_lightning", "torchmetrics", "omegaconf",
                 "pymeshlab", "open3d", "imageio", "pygit2"]:
        _stub(name)
    sys.modules["trimesh"].Trimesh = object
    sk = _stub("skimage")
    sk.measure = _stub("skimage.measure")
    _stub("smplx", MANO=object)
    sys.modules["pytorch_lightning"].LightningModule = torch.nn.Module
    # what code/src/hold/hold.py pulls in at import time (Lightning module, metrics, comet logging, debug plots)
    pll = _stub("pytorch_lightning.loggers", CometLogger=object)
    sys.modules["pytorch_lightning"].loggers = pll

    class _PSNR(torch.nn.Module):  # torchmetrics.image.PeakSignalNoiseRatio(data_range=1.0)
        def __init__(self, data_range=1.0):
            super().__init__()
            self.data_range = data_range

        def forward(self, preds, target):
            return 10.0 * torch.log10(self.data_range ** 2 / ((preds - target) ** 2).mean())

    tmi = _stub("torchmetrics.image", PeakSignalNoiseRatio=_PSNR)
    sys.modules["torchmetrics"].image = tmi
    for name in ["matplotlib", "matplotlib.pyplot"]:
        try:
            __import__(name)
        except Exception:
            _stub(name)
    # src.libmise is a Cython extension that is not built here; only meshing uses it
    _stub("src_libmise_placeholder")

    sys.path[:0] = [os.path.join(REF_ROOT, "code"), REF_ROOT]
    import src  # noqa: F401  (namespace package from the reference)

    _stub("src.libmise", mise=None)
    # embedders.get_embedder hard-codes torch.device("cuda") for BARF buffers
    import src.engine.embedders as emb

    _orig_get = emb.get_embedder

    def get_embedder_cpu(*a, **k):
        real = torch.device
        try:
            torch.device = lambda *aa, **kk: real("cpu")  # type: ignore
            return _orig_get(*a, **k)
        finally:
            torch.device = real  # type: ignore

    emb.get_embedder = get_embedder_cpu
    import src.networks.shape_net as sn
    import src.networks.texture_net as tn

    sn.get_embedder = get_embedder_cpu
    tn.get_embedder = get_embedder_cpu
    _INSTALLED = True


def load_opt():
    """confs/general.yaml as EasyDict (+ scene_bounding_sphere as parser.py:77-78 injects)."""
    import yaml

    with open(os.path.join(REF_ROOT, "code/confs/general.yaml")) as f:
        return EasyDict(yaml.safe_load(f))


def make_args(case="synth", n_images=4, **kw):
    a = EasyDict(case=case, n_images=n_images, barf_s=1000, barf_e=10000, no_barf=False,
                 shape_init="", debug=False, freeze_pose=False, log_dir="/tmp/hold_logs",
                 experiment=None, no_vis=False, render_downsample=1, fast_dev_run=False,
                 no_meshing=True, lr=5e-4, log_every=10_000_000)
    a.update(kw)
    return a


def prepare_workdir(scene: dict, case="synth") -> str:
    """temp cwd with the files the reference opens by relative path."""
    from hold_amd import synthetic as syn

    d = tempfile.mkdtemp(prefix="hold_ref_")
    os.makedirs(os.path.join(d, "body_models"))
    for is_r, nm in [(True, "RIGHT"), (False, "LEFT")]:
        with open(os.path.join(d, "body_models", f"MANO_{nm}.pkl"), "wb") as f:
            pickle.dump(syn.make_mano_model(is_r), f)
    os.makedirs(os.path.join(d, "data", case, "build"))
    data = {"entities": scene["entities"], "scene_bounding_sphere": scene["scene_bounding_sphere"]}
    np.save(os.path.join(d, "data", case, "build", "data.npy"), data, allow_pickle=True)
    return d


class chdir:
    def __init__(self, d):
        self.d = d

    def __enter__(self):
        self.old = os.getcwd()
        os.chdir(self.d)

    def __exit__(self, *a):
        os.chdir(self.old)


def build_holdnet(scene: dict, seed: int = 1, perturb: float = 0.02, sampler: dict = None):
    """Construct the reference HOLDNet on CPU under seed (reset_all_seeds-style).  sampler: overrides of
    confs/general.yaml's model.ray_sampler block (BASELINE.json configs[0] / configs[4]: N_samples = 32 / 128)."""
    install()
    from src.hold.hold_net import HOLDNet

    opt = load_opt()
    opt.model.scene_bounding_sphere = scene["scene_bounding_sphere"]
    if sampler:
        opt.model.ray_sampler.update(sampler)
    args = make_args(n_images=scene["n_frames"])
    wd = prepare_workdir(scene)
    torch.manual_seed(seed)
    np.random.seed(seed)
    ents = scene["entities"]
    with chdir(wd):
        net = HOLDNet(opt.model, ents["right"]["mean_shape"] if "right" in ents else None,
                      ents["left"]["mean_shape"] if "left" in ents else None, scene["n_frames"], args)
    if perturb > 0:
        g = torch.Generator().manual_seed(seed + 12345)
        with torch.no_grad():
            for n, p in net.named_parameters():
                if n.endswith("weight_v") or (n.endswith(".weight") and "lin" in n):
                    p.add_(torch.randn(p.shape, generator=g) * perturb)
                if "frame_latent_encoder" in n:
                    p.copy_(torch.randn(p.shape, generator=g) * 0.5)
    return net, opt, args, wd
