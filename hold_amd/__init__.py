"""hold_amd -- MI355X-native volumetric hand-object rendering path for HOLD."""
from .config import precision, set_precision  # noqa: F401

__all__ = ["build_from_scene", "reference_holdnet", "install", "xdict", "set_precision", "precision"]


def build_from_scene(scene, state_dict=None, device="cuda", **kw):
    """Construct HOLDNet for a synthetic scene (hold_amd.synthetic.make_scene) and optionally load a
    state dict keyed with the reference's parameter names."""
    import torch

    from . import synthetic as syn
    from .hold_net import HOLDNet

    ents = scene["entities"]
    mano = {"right": syn.make_mano_model(True), "left": syn.make_mano_model(False)}
    net = HOLDNet(scene["scene_bounding_sphere"], ents["right"]["mean_shape"] if "right" in ents else None,
                  ents["left"]["mean_shape"] if "left" in ents else None, scene["n_frames"], ents, mano, **kw)
    if state_dict is not None:
        sd = {k: torch.as_tensor(v) for k, v in state_dict.items()}
        missing, unexpected = net.load_state_dict(sd, strict=False)
        assert not unexpected, unexpected
        bad = [m for m in missing if "human_layer" not in m and "object_model" not in m and "alpha_max_iter" not in m]
        assert not bad, bad
    return net.to(device)


def _get(o, k, d=None):
    if o is None:
        return d
    return o.get(k, d) if isinstance(o, dict) else getattr(o, k, d)


def _reference_class():
    """class factory (hold_net imports torch; keep ``import hold_amd`` light)."""
    from .hold_net import DEFAULT_SAMPLER, HOLDNet

    class ReferenceHOLDNet(HOLDNet):
        """HOLDNet with the REFERENCE's constructor ``HOLDNet(opt, betas_r, betas_l, num_frames, args)``
        (code/src/hold/hold_net.py:23-51; built at code/src/hold/hold.py:41-47).  File-backed inputs are read from
        where the reference reads them: ./body_models/MANO_{RIGHT,LEFT}.pkl (mano/server.py:121-128) and
        ./data/<args.case>/build/data.npy (object_model.py:15-27, mano/params.py:20-23, obj/params.py:15-17), the
        per-frame pose tables are loaded from it (``params.load_params(args.case)``), the SDF nets get the
        ``init: geometry`` scheme of the config, and ``init_network()`` honours ``args.shape_init``."""

        def __init__(self, opt, betas_r, betas_l, num_frames, args):
            import os
            import pickle

            import numpy as np

            ents = np.load(os.path.join("./data", args.case, "build/data.npy"), allow_pickle=True).item()["entities"]
            mano = {}
            for side, b in (("right", betas_r), ("left", betas_l)):
                if b is not None:
                    with open(f"./body_models/MANO_{side.upper()}.pkl", "rb") as f:
                        mano[side] = pickle.load(f, encoding="latin1")
            inet = _get(opt, "implicit_network")
            super().__init__(_get(opt, "scene_bounding_sphere", 6.0), betas_r, betas_l, num_frames, ents, mano,
                             sampler_opt={**DEFAULT_SAMPLER, **dict(_get(opt, "ray_sampler") or {})},
                             barf_s=_get(args, "barf_s", 1000), barf_e=_get(args, "barf_e", 10000),
                             no_barf=_get(args, "no_barf", False), init=_get(inet, "init", "geometry"),
                             init_bias=_get(inet, "bias", 0.6))
            self.args, self.opt = args, opt
            self.init_network(_get(args, "shape_init", "") or "")

    return ReferenceHOLDNet


_REF_CLS = None


def reference_holdnet(opt, betas_r, betas_l, num_frames, args):
    """construct ``ReferenceHOLDNet`` (kept as a function for callers that rebind a constructor name)."""
    global _REF_CLS
    if _REF_CLS is None:
        _REF_CLS = _reference_class()
    return _REF_CLS(opt, betas_r, betas_l, num_frames, args)


def install():
    """One-line drop-in for the reference tree: ``import hold_amd; hold_amd.install()`` before
    ``HOLD(opt, args)`` is constructed rebinds the name ``HOLDNet`` that code/src/hold/hold.py:12 imported, so
    hold.py:41-47 builds the MI355X model; the reference's Lightning module, ``Loss`` and datasets stay as they are and
    receive ``common.xdict.xdict`` outputs (hold_amd.xdict.output_class).
    Raises ImportError if the reference's ``src`` package is not importable."""
    import importlib

    global _REF_CLS
    from . import xdict as xd

    hold_mod = importlib.import_module("src.hold.hold")
    xd._OUT_CLS = None  # re-resolve: the reference's common.xdict is importable now
    if _REF_CLS is None:
        _REF_CLS = _reference_class()
    hold_mod.HOLDNet = _REF_CLS
    try:
        importlib.import_module("src.hold.hold_net").HOLDNet = _REF_CLS
    except ImportError:
        pass
    return _REF_CLS
