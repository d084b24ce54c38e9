"""The drop-in boundary against the REAL reference (build container only: /root/reference is imported on CPU through
oracle/ref_shim).  What must hold for ``import hold_amd; hold_amd.install()`` to be a drop-in:

* the reference's own Lightning module ``src.hold.hold.HOLD(opt, args)`` constructs with our model inside,
* its ``state_dict`` has the reference's keys/shapes, loads strictly, and -- under the same torch seed -- the SAME
  initial values (geometric init, pose tables from data.npy, embeddings),
* ``configure_optimizers`` finds the same parameter groups,
* the output container answers the xdict calls the callers make, identically to the reference's xdict,
* the reference's own ``Loss.forward`` consumes an output with our key set (and pins oracle/targets_oracle.loss_forward).
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference tree only exists in the build container")


@pytest.fixture(scope="module")
def ref():
    from hold_amd import synthetic as syn
    from oracle import ref_shim
    ref_shim.install()
    sc = syn.make_scene(n_frames=4)
    wd = ref_shim.prepare_workdir(sc)
    import src.hold.hold as H
    import src.hold.hold_net as HN
    orig = (H.HOLDNet, HN.HOLDNet)

    def build(install):
        import hold_amd
        H.HOLDNet, HN.HOLDNet = orig
        if install:
            hold_amd.install()
        opt = ref_shim.load_opt()
        opt.model.scene_bounding_sphere = sc["scene_bounding_sphere"]
        args = ref_shim.make_args(n_images=sc["n_frames"])
        torch.manual_seed(1)
        np.random.seed(1)
        with ref_shim.chdir(wd):
            return H.HOLD(opt, args)

    yield dict(sc=sc, wd=wd, build=build, shim=ref_shim)
    H.HOLDNet, HN.HOLDNet = orig


def test_real_hold_module_constructs_with_identical_parameters(ref):
    theirs = ref["build"](False)
    ours = ref["build"](True)
    assert type(theirs.model).__module__ == "src.hold.hold_net"
    assert type(ours.model).__module__.startswith("hold_amd")
    sd_t, sd_o = theirs.state_dict(), ours.state_dict()
    assert list(sd_t.keys()) == list(sd_o.keys())
    for k in sd_t:
        assert sd_t[k].shape == sd_o[k].shape and sd_t[k].dtype == sd_o[k].dtype, k
        if sd_t[k].numel():
            assert torch.equal(sd_t[k], sd_o[k]), k  # same seed -> same init (geometry init, pose tables, embeddings)
    ours.load_state_dict(sd_t, strict=True)
    # pose tables are the data.npy values, not zeros (params.load_params)
    ent = ref["sc"]["entities"]["right"]
    assert torch.allclose(ours.model.nodes["right"].params.pose.weight, torch.tensor(ent["hand_poses"][:, 3:], dtype=torch.float32))
    assert float(ours.model.nodes["right"].implicit_network.lin8.bias.detach()[0]) == pytest.approx(-0.6)
    # optimiser groups (hold.py:79-101): one 0.1 x lr group per node + the main group, same sizes
    ot, oo = theirs.configure_optimizers()[0][0], ours.configure_optimizers()[0][0]
    gt = [(g["lr"], sum(p.numel() for p in g["params"])) for g in ot.param_groups]
    go_ = [(g["lr"], sum(p.numel() for p in g["params"])) for g in oo.param_groups]
    assert gt == go_
    # requires_grad pattern after HOLD.__init__ (defrosted pose tables)
    rt = {n: p.requires_grad for n, p in theirs.named_parameters()}
    ro = {n: p.requires_grad for n, p in ours.named_parameters()}
    assert rt == ro
    # node surface the Lightning module touches
    for node in ours.model.nodes.values():
        assert hasattr(node, "meshing_cano") and hasattr(node, "params") and hasattr(node, "server")
    assert hasattr(ours.model.nodes["right"], "spawn_cano_mano") and hasattr(ours.model.nodes["object"], "update_cano")


def test_xdict_matches_reference_semantics(ref):
    from common.xdict import xdict as RX
    from hold_amd.xdict import xdict as OX
    import hold_amd.xdict as xd
    ref["build"](True)
    assert xd.output_class() is RX  # after install() the model returns the reference's own container type
    base = {"rgb": torch.rand(4, 3), "right.fg_rgb.vis": torch.rand(4, 3), "object.fg_rgb.vis": torch.rand(4, 3),
            "right.index_off_surface": torch.rand(4) > 0.5, "step": 3, "names": ["a", "b"]}
    a, b = RX(dict(base)), OX(dict(base))
    for op in (lambda d: d.search("fg_rgb.vis"), lambda d: d.search("right.", "r_"), lambda d: d.prefix("n."),
               lambda d: d.postfix(".x"), lambda d: d.rm("vis"), lambda d: d.subset(["rgb", "step"]),
               lambda d: d.replace_keys("fg_", "FG"), lambda d: d.detach(), lambda d: d.to("cpu")):
        ra, rb = op(a), op(b)
        assert list(ra.keys()) == list(rb.keys())
        for k in ra:
            if torch.is_tensor(ra[k]):
                assert torch.equal(ra[k], rb[k])
            else:
                assert ra[k] == rb[k]
    assert a.sorted_keys() == b.sorted_keys()
    with pytest.raises(AssertionError):
        b["rgb"] = 1  # no silent overwrite
    with pytest.raises(AssertionError):
        b.merge({"rgb": 1})
    b.overwrite("step", 4)
    b.merge({"new": 1})
    assert b["step"] == 4 and b["new"] == 1
    assert b.fuzzy_get("index_off") is base["right.index_off_surface"]
    assert not b.subset(["rgb"]).has_invalid()


def _fake_outputs(n_frames=2, n_pix=64, seed=0, step=12345):
    g = torch.Generator().manual_seed(seed)
    N = n_frames * n_pix
    r = lambda *s: torch.rand(*s, generator=g)
    out = {"rgb": r(N, 3), "semantics": r(N, 4), "step": step, "epoch": 1}
    for nid in ("right", "object"):
        out[f"{nid}.mask_prob"] = r(N, 1)
        out[f"{nid}.index_off_surface"] = r(N) > 0.4
        out[f"{nid}.grad_theta"] = torch.randn(n_frames, 307, 3, generator=g) * 2.0 + 1.0
    out["right.pts2mano_sdf_cano"] = torch.randn(n_frames, 307, generator=g) * 0.02
    out["right.pred_sdf"] = torch.randn(n_frames, 307, generator=g) * 0.02
    batch = {"gt.rgb": r(n_frames, n_pix, 3), "idx": torch.arange(n_frames),
             "gt.mask": torch.tensor([0, 50, 150, 250])[torch.randint(0, 4, (n_frames, n_pix), generator=g)]}
    return batch, out


def test_reference_loss_consumes_our_outputs_and_pins_the_loss_oracle(ref, tmp_path):
    """the reference's Loss.forward (code/src/hold/loss.py:17-93) on an output mapping with HOLDNet's key set: same
    numbers from the real Loss, from the oracle restatement, and (key set) from hold_amd.loss.Loss's expectations."""
    from PIL import Image
    from common.xdict import xdict as RX
    from oracle import targets_oracle as to
    from src.hold.loss import Loss as RefLoss
    import hold_amd.xdict as xd
    png = tmp_path / "im.png"
    Image.fromarray(np.zeros((8, 8, 3), np.uint8)).save(png)
    for step in (0, 12345, 40000):
        batch, out = _fake_outputs(step=step)
        batch["im_path"] = [[str(png)]]
        ref_ld = RefLoss(ref["shim"].make_args())(RX(dict(batch)), xd.xdict(dict(out)))  # OUR container class in
        ora_ld = to.loss_forward(batch, out)
        assert set(ref_ld.keys()) == set(ora_ld.keys())
        for k in ref_ld:
            assert float(ref_ld[k]) == pytest.approx(float(ora_ld[k]), rel=1e-6, abs=1e-9), (step, k)


def test_recorded_hip_training_output_feeds_the_reference_loss(ref, tmp_path, gold_dir):
    """a HOLDNet training-mode output recorded on the MI355X (tests/golden/hip_train_output.npz, written by
    scripts/record_hip_outputs.py) goes through the reference's own Loss; value == the HIP Loss recorded with it."""
    path = os.path.join(gold_dir, "hip_train_output.npz")
    if not os.path.exists(path):
        pytest.skip("fixture not recorded yet (scripts/record_hip_outputs.py on the GPU box)")
    from PIL import Image
    from common.xdict import xdict as RX
    from src.hold.loss import Loss as RefLoss
    import hold_amd.xdict as xd
    z = np.load(path, allow_pickle=True)
    out = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("out.")}
    out["step"], out["epoch"] = int(z["step"]), int(z["epoch"])
    batch = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("batch.")}
    png = tmp_path / "im.png"
    Image.fromarray(np.zeros((8, 8, 3), np.uint8)).save(png)
    batch["im_path"] = [[str(png)]]
    for need in ("right.index_off_surface", "right.grad_theta", "right.pts2mano_sdf_cano", "right.pred_sdf",
                 "object.index_off_surface", "object.grad_theta"):
        assert need in out, need
    ld = RefLoss(ref["shim"].make_args())(RX(batch), xd.xdict(out))
    for k in ("loss", "loss/rgb", "loss/sem", "loss/mano_cano", "loss/opacity_sparse"):
        assert float(ld[k]) == pytest.approx(float(z["loss." + k]), rel=2e-5, abs=1e-7), k
