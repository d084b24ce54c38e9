"""hold_amd -- MI355X-native volumetric hand-object rendering path for HOLD."""
__all__ = ["build_from_scene"]


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
