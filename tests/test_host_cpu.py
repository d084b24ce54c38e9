"""CPU-only checks of the host side: the C-ABI library loads and exports every declared symbol, the
module tree carries the reference's state_dict names, synthetic inputs are deterministic."""
import os
import re

import numpy as np
import pytest
import torch

from parity_common import ROOT, syn


def test_library_exports_every_declared_symbol():
    from hold_amd import _lib, build

    build.build()
    hdr = open(os.path.join(ROOT, "include", "hold_hip.h")).read()
    assert "HOLD_DEV" not in hdr and "hold_diag" not in hdr  # the public header declares the product's ABI only
    dev_hdr = open(os.path.join(ROOT, "hold_amd", "csrc", "dev", "hold_hip_dev.h")).read()
    dev_only = set(re.findall(r"\b(?:int|int64_t)\s+(hold_[a-z0-9_]+)\s*\(", dev_hdr))
    assert dev_only == set(_lib.DEV_SIGNATURES)  # diagnostics: developer build only, absent from the product library
    declared = set(re.findall(r"\b(?:int|int64_t|float)\s+(hold_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 27
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert declared - {"hold_abi_version", "hold_wgrad_workspace_floats", "hold_fused_sdf_pack_floats", "hold_chain_pack_floats", "hold_fused_sdf_x6_pack_bytes",
                       "hold_silhouette_workspace_floats", "hold_reduce_workspace_floats", "hold_chain_x6_pack_bytes", "hold_trunk_r6_pack_bytes", "hold_chain_r6_pack_bytes", "hold_gemm_r6_pack_bytes",
                       "hold_wcolsum_workspace_floats", "hold_head3_workspace_floats", "hold_wgrad_group_workspace_floats", "hold_trunk_h3_pack_bytes",
                       "hold_trunk_h3_act_scale", "hold_alive_blocks", "hold_chain_h3_pack_bytes", "hold_gemm_h3_pack_bytes"} == set(_lib.SIGNATURES)
    assert L.hold_abi_version() == 1
    assert not any(hasattr(L, n) for n in dev_only)
    # the product library reads no environment variables (stateless C ABI): no getenv import in the shared object
    import subprocess
    syms = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "getenv" not in syms


def test_header_is_plain_c_and_a_c_program_links_the_library(tmp_path):
    """the drop-in boundary is a C ABI: include/hold_hip.h compiles as strict C99 and as C++, and a C program that includes it links
    libholdhip.so, gets the ABI version, the workspace sizes (pure host arithmetic) and HOLD_E_ARG for null pointers -- the calls
    a cgo / JNI / ctypes binding makes first, none of which needs a GPU."""
    import shutil
    import subprocess
    from hold_amd import _lib, build

    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    build.build()
    src = tmp_path / "consumer.c"
    src.write_text("""
#include <stdio.h>
#include <string.h>
#include "hold_hip.h"
int main(void) {
  hold_gemm_desc d;
  memset(&d, 0, sizeof d);
  if (hold_abi_version() != 1) return 1;
  if (hold_gemm_nt(&d, NULL) != HOLD_E_ARG) return 2;              /* null pointers are refused before any launch */
  if (hold_wgrad_workspace_floats(256, 256, 8) <= 0) return 3;
  if (hold_trunk_h3_pack_bytes() != 116 * 16384) return 4;          /* 116 k steps of one 16 KiB slot */
  if (hold_trunk_h3_act_scale() != 64.0f) return 5;
  if (hold_alive_blocks(1025) != 2) return 6;                       /* 1 024 samples per block */
  printf("abi %d\\n", hold_abi_version());
  return 0;
}
""")
    inc = os.path.join(ROOT, "include")
    for cc, std in (("gcc", "-std=c99"), ("g++", "-std=c++17")):
        r = subprocess.run([cc, std, "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-fsyntax-only", "-x", "c" if cc == "gcc" else "c++",
                            str(src)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    exe = tmp_path / "consumer"
    libdir = os.path.dirname(_lib.LIB_PATH)
    r = subprocess.run(["gcc", "-std=c99", "-I", inc, str(src), "-o", str(exe), "-L", libdir, "-l:" + os.path.basename(_lib.LIB_PATH),
                        "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "abi 1", (r.returncode, r.stdout, r.stderr)


def test_bad_arguments_are_rejected_without_a_gpu():
    import ctypes as C
    from hold_amd import _lib

    L = _lib.lib()
    d = _lib.GemmDesc()
    assert L.hold_gemm_nt(C.byref(d), None) == -1  # HOLD_E_ARG: null pointers
    assert L.hold_gemm_nt(None, None) == -1


def test_state_dict_names_match_reference_layout():
    import hold_amd

    sc = syn.make_scene(3)
    sd = syn.make_state_dict(sc)
    net = hold_amd.build_from_scene(sc, sd, device="cpu")
    mine = net.state_dict()
    for k, v in sd.items():
        assert k in mine and tuple(mine[k].shape) == tuple(np.shape(v)), k
    assert float(mine["nodes.right.implicit_network.lin3.weight_v"].shape[0]) == 217


def test_product_path_refuses_to_run_without_gpu():
    import hold_amd

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    sc = syn.make_scene(2)
    net = hold_amd.build_from_scene(sc, syn.make_state_dict(sc), device="cpu")
    with pytest.raises(RuntimeError):
        net({"uv": torch.zeros(1, 4, 2)})


def test_synthetic_inputs_are_deterministic():
    a, b = syn.make_mano_model(True), syn.make_mano_model(True)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    s1, s2 = syn.make_state_dict(syn.make_scene(3)), syn.make_state_dict(syn.make_scene(3))
    assert all(np.array_equal(s1[k], s2[k]) for k in s1)
    w = a["weights"]
    assert np.allclose(w.sum(1), 1.0) and a["v_template"].shape == (778, 3)


def test_x6_limb_pack_layout_and_arithmetic():
    """hold_fused_sdf_x6's weight pack (field.pack_x6) read back with the KERNEL's index arithmetic
    (csrc/fused_sdf.hip: unit = step*1536 + limb*512 + wave*64 + lane, lane = 32*h + i, 8 bf16 per unit) and combined
    with on-the-fly activation limbs through the six limb products reproduces W @ x to fp32 accuracy."""
    import torch
    from hold_amd import field as F

    g = torch.Generator().manual_seed(0)
    W8 = [torch.randn(256, 40, generator=g) / 6] + [torch.randn(256, 256, generator=g) / 16 for _ in range(7)]
    W8[3] = W8[3][:217]
    pack = F.pack_x6(W8)
    assert pack.dtype == torch.bfloat16 and pack.numel() * 2 == (3 + 7 * 16) * 1536 * 16
    units = pack.float().reshape(-1, 8)  # 16-byte units
    step0 = 0
    for l, wl in enumerate(W8):
        K = 48 if l == 0 else 256
        steps = K // 16
        x = torch.zeros(32, K)
        x[:, :wl.shape[1]] = torch.randn(32, wl.shape[1], generator=g)  # 32 points = one MFMA tile
        xl = [t.float() for t in F.split_limbs(x)]
        assert torch.equal(xl[0] + xl[1] + xl[2], x)  # three bf16 limbs hold an fp32 exactly
        for wave in (0, 3, 6, 7):
            acc = torch.zeros(32, 32)  # D[i = feature][j = point]
            for s in range(steps):
                for wlimb, alimb in ((0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)):
                    A = torch.zeros(32, 16)  # A[i][k]: lane (h, i) supplies k = 8h + e
                    for h in range(2):
                        lanes = torch.arange(32) + 32 * h
                        A[:, 8 * h:8 * h + 8] = units[(step0 + s) * 1536 + wlimb * 512 + wave * 64 + lanes]
                    B = xl[alimb][:, 16 * s:16 * s + 16]  # B[k][j] = act[j][16 s + k]
                    acc += A @ B.t()
            ref = torch.zeros(256, K, dtype=torch.float64)
            ref[:wl.shape[0], :wl.shape[1]] = wl.double()
            ref = ref[32 * wave:32 * wave + 32] @ x.double().t()
            err = (acc.double() - ref).abs().max().item()
            assert err < 2e-6 * max(1.0, ref.abs().max().item()), (l, wave, err)
        step0 += steps


def test_reference_signature_drop_in(tmp_path, monkeypatch):
    """hold_amd.install() rebinds the HOLDNet name code/src/hold/hold.py:12 imported; the replacement takes the
    reference's constructor arguments and reads MANO pickles / data.npy from the reference's relative paths."""
    import pickle
    import sys
    import types

    import numpy as np

    import hold_amd

    sc = syn.make_scene(3)
    (tmp_path / "body_models").mkdir()
    (tmp_path / "data" / "synth" / "build").mkdir(parents=True)
    for side, r in (("RIGHT", True), ("LEFT", False)):
        pickle.dump(syn.make_mano_model(r), open(tmp_path / "body_models" / f"MANO_{side}.pkl", "wb"))
    np.save(tmp_path / "data" / "synth" / "build" / "data.npy", {"entities": sc["entities"]}, allow_pickle=True)
    monkeypatch.chdir(tmp_path)
    # stand-ins for the reference's modules (the real ones need Lightning etc.)
    for name in ("src", "src.hold", "src.hold.hold", "src.hold.hold_net"):
        m = types.ModuleType(name)
        m.__path__ = []
        monkeypatch.setitem(sys.modules, name, m)
    sys.modules["src.hold.hold"].HOLDNet = object
    sys.modules["src.hold.hold_net"].HOLDNet = object
    ctor = hold_amd.install()
    assert sys.modules["src.hold.hold"].HOLDNet is ctor and sys.modules["src.hold.hold_net"].HOLDNet is ctor
    opt = types.SimpleNamespace(scene_bounding_sphere=sc["scene_bounding_sphere"],
                                ray_sampler=dict(near=0.0, N_samples=64, N_samples_eval=128, N_samples_extra=32, eps=0.1,
                                                 beta_iters=10, max_total_iters=5))
    args = types.SimpleNamespace(case="synth", barf_s=1000, barf_e=10000, no_barf=False)
    net = ctor(opt, sc["entities"]["right"]["mean_shape"], None, sc["n_frames"], args)
    assert sorted(net.nodes.keys()) == ["object", "right"]
    names = set(net.state_dict().keys())
    assert "nodes.right.implicit_network.lin0.weight_v" in names and "background.bg_implicit_network.lin0.weight" in names


def test_chain_weight_packs_follow_the_header_layout():
    """field.pack_weights: pk["fused"] (forward-type sweeps) and pk["chain_bwd"] (descending sweeps, M_j = W_l^T with
    l = 7 - j, zero padded) are in the fragment order include/hold_hip.h documents for hold_chain / hold_fused_sdf:
    [K/8 chunks][8 n-tiles][2 halves h][32 rows i][4] = M[32*nt + i][8*chunk + 4*h + c]."""
    import torch
    from hold_amd import field as F

    g = torch.Generator().manual_seed(0)
    spec = F.FieldSpec("object")
    iw = [torch.randn(256, 39, generator=g)] + [torch.randn(256, 256, generator=g) for _ in range(2)] + \
         [torch.randn(217, 256, generator=g)] + [torch.randn(256, 256, generator=g) for _ in range(4)] + \
         [torch.randn(257, 256, generator=g)]
    ib = [torch.randn(w.shape[0], generator=g) for w in iw]
    rw = [torch.randn(256, spec.rin_dim, generator=g)] + [torch.randn(256, 256, generator=g) for _ in range(3)] + \
         [torch.randn(3, 256, generator=g)]
    rb = [torch.randn(w.shape[0], generator=g) for w in rw]
    pk = F.pack_weights(spec, iw, ib, rw, rb, need_bwd=True)

    def unpack(flat, K):
        return flat.reshape(K // 8, 8, 2, 32, 4).permute(1, 3, 0, 2, 4).reshape(256, K)

    wpack, bias8 = pk["fused"]
    off = 0
    for l in range(8):
        K = 40 if l == 0 else 256
        m = unpack(wpack[off:off + 256 * K], K)
        off += 256 * K
        ref = torch.zeros(256, K)
        ref[:pk["W"][l].shape[0], :pk["W"][l].shape[1]] = pk["W"][l]
        assert torch.equal(m, ref), l
        assert torch.equal(bias8[l, :pk["b"][l].shape[0]], pk["b"][l])
    assert off == wpack.numel()
    cb = pk["chain_bwd"]
    assert cb.numel() == 7 * 65536
    for j in range(7):
        l = 7 - j
        m = unpack(cb[j * 65536:(j + 1) * 65536], 256)
        ref = torch.zeros(256, 256)
        ref[:pk["W"][l].shape[1], :pk["W"][l].shape[0]] = pk["W"][l].t()
        assert torch.equal(m, ref), j
    # the skip folding: layer 4 carries the 1/sqrt(2) of cat([x, input]) / sqrt(2)  (shape_net.py:122-123)
    assert torch.allclose(pk["W"][4], iw[4] / 2 ** 0.5)


def test_render_input_column_permutation_is_a_bijection_onto_the_reference_order():
    """hold_amd.field.rin_perm: our rendering-net input = the reference's columns with the 256 feature columns first"""
    import torch
    from hold_amd import field as F
    for rin_dim in (270, 302):
        perm = F.rin_perm(rin_dim)
        assert sorted(perm.tolist()) == list(range(rin_dim))
        ref_cols = torch.arange(rin_dim)
        ours = ref_cols[perm]  # column j of our layout holds reference column perm[j]
        assert ours[F.RIN_FEAT:F.RIN_FEAT + 256].tolist() == list(range(14, 270))        # feature_vectors
        assert ours[F.RIN_X:F.RIN_X + 3].tolist() == [0, 1, 2]                           # canonical points
        assert ours[F.RIN_N:F.RIN_N + 3].tolist() == [3, 4, 5]                           # normals
        assert ours[F.RIN_POSE:F.RIN_POSE + 8].tolist() == list(range(6, 14))            # pose embedding
        if rin_dim > 270:
            assert ours[F.RIN_TIME:].tolist() == list(range(270, rin_dim))               # frame encoding
        g = torch.randn(4, rin_dim)
        back = torch.empty_like(g)
        back[:, perm] = g[:, perm][:, torch.arange(rin_dim)]
        assert torch.equal(back, g)


def _cpu_net(n_frames=4):
    import torch
    import hold_amd
    from hold_amd import synthetic as syn
    torch.manual_seed(0)
    sc = syn.make_scene(n_frames=n_frames)
    net = hold_amd.build_from_scene(sc, syn.make_state_dict(sc), device="cpu")
    for node in net.nodes.values():
        node.params.defrost()
    return net


def test_flat_adam_state_dict_round_trip_and_rehome():
    """round-2 advisor: the optimiser state (moments, step count) must survive checkpoint / resume, and a parameter whose
    storage was replaced after construction (net.to(), .float(), GenericParams.init_parameters) must be brought back into
    the bucket instead of being silently left behind (host logic only: the update kernels themselves need the GPU)"""
    import torch
    from hold_amd.optim import FlatAdam
    a = FlatAdam(_cpu_net(), lr=5e-4, clip_norm=0.5)
    g = torch.Generator().manual_seed(1)
    a.m.copy_(torch.randn(a.m.shape, generator=g))
    a.v.copy_(torch.rand(a.v.shape, generator=g))
    a.step_count = 17
    sd = a.state_dict()
    b = FlatAdam(_cpu_net(), lr=1e-3, clip_norm=0.0)
    b.load_state_dict(sd)
    assert b.step_count == 17 and torch.equal(b.m, a.m) and torch.equal(b.v, a.v)
    assert b.lr == a.lr and b.clip_norm == a.clip_norm and tuple(b.betas) == tuple(a.betas)
    sd["m"].zero_()  # the state dict owns copies, not views of the live buffers
    assert float(a.m.abs().max()) > 0
    with pytest.raises(ValueError):
        FlatAdam(_cpu_net(n_frames=6), lr=5e-4).load_state_dict(sd)  # other pose-table sizes: another layout
    # re-homing
    p = b.params[3]
    off = b.offsets[3]
    assert p.data_ptr() == b.flat.data_ptr() + 4 * off and b.rehome() == 0
    p.data = p.data.clone() * 2.0  # what net.float() / init_parameters do: a new storage behind the same Parameter
    assert p.data_ptr() != b.flat.data_ptr() + 4 * off
    want = p.data.clone()
    assert b.rehome() == 1
    assert p.data_ptr() == b.flat.data_ptr() + 4 * off and torch.equal(p.data, want)
    assert torch.equal(b.flat[off:off + p.numel()].view(p.shape), want)


def test_weight_pack_key_follows_writes_torch_cannot_see():
    """round-2 advisor: the weight-pack cache is keyed on torch's version counters; load_state_dict, .to()/.float() and the
    optimiser's raw-pointer update must invalidate it too (hooks / config.bump_weights_epoch)"""
    import torch
    from hold_amd import config
    from hold_amd.hold_net import _pack_key
    net = _cpu_net()
    mod = net.nodes["object"].implicit_network
    k0 = _pack_key((mod,), True)
    assert _pack_key((mod,), True) == k0  # stable while nothing changes
    with torch.no_grad():
        mod.lin1.bias.add_(1.0)  # an in-place torch update: the version counter moves
    k1 = _pack_key((mod,), True)
    assert k1 != k0
    net.load_state_dict(net.state_dict())  # post-hook
    k2 = _pack_key((mod,), True)
    assert k2 != k1
    net.float()  # _apply
    k3 = _pack_key((mod,), True)
    assert k3 != k2
    mod.lin1.bias.data.mul_(0.5)  # a write through .data: invisible to torch -- the documented contract is the explicit bump
    assert _pack_key((mod,), True) == k3
    config.bump_weights_epoch()
    assert _pack_key((mod,), True) != k3


def test_barf_counter_is_mirrored_on_the_host():
    """BarfEmbedder.step() must not read its device buffer back (a host sync per node and step): the counter lives on the
    host, the checkpointed buffer follows it, and load_state_dict re-synchronises the host copy"""
    import torch
    net = _cpu_net()
    emb = net.nodes["object"].implicit_network.embedder_obj
    it0 = emb._iter_host
    assert int(emb.alpha_iter) == it0
    emb.step()
    assert emb._iter_host == it0 + 1 and int(emb.alpha_iter) == it0 + 1
    sd = net.state_dict()
    key = next(k for k in sd if k.endswith("nodes.object.implicit_network.embedder_obj.alpha_iter"))
    sd[key] = torch.tensor(1234)
    net.load_state_dict(sd)
    assert emb._iter_host == 1234 and int(emb.alpha_iter) == 1234
    w = emb.weights("cpu")
    assert w is None or (w.shape[0] == 39 and bool(torch.isfinite(w).all()))


def test_row_split_size_passes_the_kernels_32_bit_offset_check():
    """hold_trunk_r6 / hold_chain_* reject (P + 128) * ld * 4 >= 2^32; the host splits larger batches by rows.  The split size
    must itself pass that check (round-3 advisor: 4 194 176 rows at ld = 256 gave exactly 2^32) and be the largest that does."""
    from hold_amd import kernels as K
    for ld in (256, 260, 272, 304, 320):
        rows = K._max_rows(ld)
        assert rows % 128 == 0 and (rows + 128) * ld * 4 < 2 ** 32
        assert (rows + 128 + 128) * ld * 4 >= 2 ** 32
    assert K._TRUNK_MAX_ROWS == K._max_rows(256) == K._CHAIN_MAX_ROWS


def test_flat_adam_loads_an_optimizer_state_shaped_like_the_reference_checkpoint():
    """round-4 advisor (medium): `HOLD.configure_optimizers` (code/src/hold/hold.py:79-101) builds ONE 0.1 lr group PER NODE
    from ALL of node.params.parameters() -- frozen and zero-sized parameters included -- and a main group with every other
    parameter.  A state dict with exactly that structure, produced by a plain torch.optim.Adam built the reference's way over
    a model with a frozen table and a zero-sized weight, must (a) load into FlatAdam.torch_optimizer() through torch's own
    load_state_dict (group count and sizes agree), (b) come back through import_from, and (c) load directly through
    load_reference_state, which maps node groups by shape and refuses an ambiguous assignment; a failed load leaves the
    optimiser untouched (advisor, low)."""
    import torch
    from torch import nn
    from hold_amd.optim import FlatAdam

    class _Params(nn.Module):
        def __init__(self, two_same):
            super().__init__()
            self.pose = nn.Embedding(4, 6)
            self.betas = nn.Embedding(1, 10)
            self.betas.weight.requires_grad_(False)  # frozen, as GenericParams.freeze leaves the shape table
            self.transl = nn.Embedding(4, 3)
            if two_same:
                self.orient = nn.Embedding(4, 3)

    class _Node(nn.Module):
        def __init__(self, two_same):
            super().__init__()
            self.params = _Params(two_same)
            self.net = nn.Linear(5, 3)
            self.empty = nn.Linear(0, 8, bias=False)  # the object's lin_pose.weight is [8, 0]

    class _Net(nn.Module):
        def __init__(self, two_same=False):
            super().__init__()
            self.nodes = nn.ModuleDict({"right": _Node(two_same), "object": _Node(False)})
            self.bg = nn.Linear(7, 2)

    def reference_adam(model, lr):  # the reference's construction, with list() in place of the set for a defined order
        node_params, groups = set(), []
        for node in model.nodes.values():
            ps = list(node.params.parameters())
            node_params.update(ps)
            groups.append({"params": ps, "lr": lr * 0.1})
        groups.append({"params": [p for p in model.parameters() if p not in node_params], "lr": lr})
        return torch.optim.Adam(groups, lr=lr, eps=1e-8)

    torch.manual_seed(0)
    net = _Net()
    ref = reference_adam(net, 5e-4)
    for p in net.parameters():
        if p.requires_grad and p.numel():
            p.grad = torch.randn_like(p)
    for _ in range(3):
        ref.step()
    sd = ref.state_dict()
    assert [len(g["params"]) for g in sd["param_groups"]] == [3, 3, 8]  # 3 tables per node; 2 x (W, b, empty W) + bg W, b
    opt = FlatAdam(net, lr=5e-4)
    t = opt.torch_optimizer()
    assert [len(g["params"]) for g in t.param_groups] == [3, 3, 8]
    t.load_state_dict(sd)  # (a): round 4 raised here on group count and sizes
    opt.import_from(t)     # (b)
    assert opt.step_count == 3
    for p, off in zip(opt.params, opt.offsets):
        assert torch.equal(opt.m[off:off + p.numel()].view(p.shape), ref.state[p]["exp_avg"])
        assert torch.equal(opt.v[off:off + p.numel()].view(p.shape), ref.state[p]["exp_avg_sq"])
    m0, v0 = opt.m.clone(), opt.v.clone()
    opt.m.zero_(); opt.v.zero_(); opt.step_count = 0
    opt.load_reference_state(sd)  # (c)
    assert opt.step_count == 3 and torch.equal(opt.m, m0) and torch.equal(opt.v, v0)
    # a state whose entries disagree on the step count is rejected BEFORE anything is written
    bad = {k: dict(v) for k, v in opt.named_state().items()}
    first = next(iter(bad))
    bad[first]["step"] = torch.tensor(9.0)
    bad[first]["exp_avg"] = bad[first]["exp_avg"] + 1
    with pytest.raises(ValueError):
        opt.load_named_state(bad)
    assert opt.step_count == 3 and torch.equal(opt.m, m0) and torch.equal(opt.v, v0)
    wrong = {k: dict(v) for k, v in opt.named_state().items()}
    wrong[first]["exp_avg"] = torch.zeros(2, 2)
    with pytest.raises(ValueError):
        opt.load_named_state(wrong)
    assert torch.equal(opt.m, m0)
    # a parameter torch holds no state for yet (no gradient so far) gets zero moments under the common step
    part = opt.named_state()
    del part[first]
    opt.load_named_state(part)
    off0 = opt.offsets[0]
    assert opt.step_count == 3 and float(opt.m[off0:off0 + opt.params[0].numel()].abs().max()) == 0.0
    # two tables of one shape in a node group with different state: the reference's set order decides -> ambiguous -> raises
    net2 = _Net(two_same=True)
    ref2 = reference_adam(net2, 5e-4)
    for p in net2.parameters():
        if p.requires_grad and p.numel():
            p.grad = torch.randn_like(p)
    ref2.step()
    opt2 = FlatAdam(net2, lr=5e-4)
    with pytest.raises(ValueError, match="ambiguous"):
        opt2.load_reference_state(ref2.state_dict())
    # ... unless the caller chooses: (1) zero moments for the ambiguous tables, everything else loaded (advisor r5: every REAL
    # reference checkpoint has MANO's global_orient / transl tables, both [n_frames, 3], both trained)
    sd2 = ref2.state_dict()
    with pytest.warns(UserWarning, match="start from zero"):
        opt2.load_reference_state(sd2, ambiguous="zero")
    names2 = {id(p): n for n, p in net2.named_parameters()}
    n_zero = 0
    for p, off in zip(opt2.params, opt2.offsets):
        got = opt2.m[off:off + p.numel()].view(p.shape)
        if torch.equal(got, ref2.state[p]["exp_avg"]):
            continue
        assert float(got.abs().max()) == 0.0, names2[id(p)]
        n_zero += 1
    assert n_zero == 2 and opt2.step_count == 1
    # (2) an explicit index -> name map for the node group (here: the construction order, which this process knows)
    idx_map = {}
    for gi, (a, b) in enumerate(zip(sd2["param_groups"], opt2.reference_groups())):
        if gi < len(net2.nodes):
            idx_map.update({i: names2[id(p)] for i, p in zip(a["params"], b["params"])})
    opt2.load_reference_state(sd2, index_to_name=idx_map)
    for p, off in zip(opt2.params, opt2.offsets):
        assert torch.equal(opt2.m[off:off + p.numel()].view(p.shape), ref2.state[p]["exp_avg"])
    t2 = opt2.torch_optimizer()
    t2.load_state_dict(ref2.state_dict())  # with the order known (same construction) torch's positional load is exact
    opt2.import_from(t2)
    for p, off in zip(opt2.params, opt2.offsets):
        assert torch.equal(opt2.m[off:off + p.numel()].view(p.shape), ref2.state[p]["exp_avg"])


def test_flat_adam_state_interchange_with_torch_adam_and_rehome_errors():
    """round-3 advisor: FlatAdam's moments must be exportable to / importable from torch.optim.Adam's per-parameter state
    (what the reference's Lightning checkpoints carry, hold.py:79-101), and rehome() must refuse a parameter that moved to
    another device instead of silently pulling it back.  (The bucket layout needs no GPU; step() does.)"""
    import torch
    from torch import nn
    from hold_amd.optim import FlatAdam

    class _Node(nn.Module):
        def __init__(self):
            super().__init__()
            self.params = nn.Embedding(4, 6)
            self.net = nn.Linear(5, 3)

    class _Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.nodes = nn.ModuleDict({"right": _Node(), "object": _Node()})
            self.bg = nn.Linear(7, 2)

    torch.manual_seed(0)
    net = _Net()
    opt = FlatAdam(net, lr=5e-4)
    opt.m.copy_(torch.rand(opt.n))
    opt.v.copy_(torch.rand(opt.n))
    opt.step_count = 17
    t = opt.torch_optimizer()
    # the reference's group structure (hold.py:79-101): one 0.1 lr group per node, then the main group
    assert [g["lr"] for g in t.param_groups] == [5e-5, 5e-5, 5e-4]
    for g, n in zip(t.param_groups, net.nodes):
        assert [id(p) for p in g["params"]] == [id(net.nodes[n].params.weight)]
    for p, off in zip(opt.params, opt.offsets):
        st = t.state[p]
        assert float(st["step"]) == 17 and torch.equal(st["exp_avg"].reshape(-1), opt.m[off:off + p.numel()])
        assert torch.equal(st["exp_avg_sq"].reshape(-1), opt.v[off:off + p.numel()])
    sd = t.state_dict()  # torch's own (index-keyed) checkpoint format round-trips through a second torch optimiser
    t2 = torch.optim.Adam(t.param_groups)
    t2.load_state_dict(sd)
    m0, v0 = opt.m.clone(), opt.v.clone()
    opt.m.zero_(); opt.v.zero_(); opt.step_count = 0
    opt.import_from(t2)
    # padding elements between the 256-byte aligned pieces are not parameters: compare the parameter ranges
    for p, off in zip(opt.params, opt.offsets):
        assert torch.equal(opt.m[off:off + p.numel()], m0[off:off + p.numel()])
        assert torch.equal(opt.v[off:off + p.numel()], v0[off:off + p.numel()])
    assert opt.step_count == 17
    named = opt.named_state()
    assert set(named) == {n for n, p in net.named_parameters()}
    # a parameter detached from the bucket on the same device is re-attached together with its gradient view ...
    p = net.bg.weight
    p.data = p.data.clone() + 1.0
    p.grad = torch.ones_like(p)
    assert opt.rehome() == 1
    off = opt.offsets[[id(q) for q in opt.params].index(id(p))]
    assert p.data_ptr() == opt.flat.data_ptr() + 4 * off and p.grad.data_ptr() == opt.grad.data_ptr() + 4 * off
    assert float(opt.grad[off]) == 1.0
    # ... one that moved to another device is an error
    opt.flat = torch.empty(opt.n, device="meta")  # (no second real device here: the bucket "lives elsewhere")
    with pytest.raises(RuntimeError):
        opt.rehome()


def test_silhouette_oracle_per_pixel_depth_cull_follows_pytorch3d():
    """the rasteriser oracle culled a face when ANY vertex lay behind the camera; pytorch3d
    (rasterize_meshes.cu) skips a face only when ALL do (zmax < 0) and otherwise drops (pixel, face) pairs whose interpolated
    depth -- perspective-corrected, clipped barycentric weights -- is negative.  The oracle now restates that rule
    (cull="pixel", its default); for meshes in front of the camera -- the only configuration hold_amd.fitting accepts, see
    check_faces_per_pixel -- it equals the whole-face rule of the HIP kernel (cull="face") bit for bit, and for a triangle
    that straddles the image plane the two differ, which is what the check refuses."""
    import sys
    import numpy as np
    import torch
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle import fitting_oracle as fo
    g = torch.Generator().manual_seed(0)
    v = torch.randn(2, 30, 3, generator=g) * 0.03
    v[..., 2] += 0.5
    f = torch.randint(0, 30, (40, 3), generator=g)
    sig = 1e-4
    blur = float(np.log(1 / 1e-4 - 1) * sig)
    args = (300.0, 300.0, 32.0, 32.0, 64, 64, sig, blur)
    a = fo.soft_silhouette(v, f, *args)
    b = fo.soft_silhouette(v, f, *args, cull="face")
    assert 0.02 < float(a.mean()) < 0.9 and torch.equal(a, b)
    # one large triangle with one vertex behind the camera: dropped as a whole by the face rule, kept by pytorch3d's where the
    # interpolated depth is >= 0
    tri = torch.tensor([[[-0.05, -0.05, 0.5], [0.05, -0.05, 0.5], [0.0, 0.4, -0.1]]])
    ft = torch.tensor([[0, 1, 2]])
    a = fo.soft_silhouette(tri, ft, *args)
    b = fo.soft_silhouette(tri, ft, *args, cull="face")
    assert float(b.abs().max()) == 0.0 and float(a.max()) > 0.01
    # all vertices behind the camera: nothing is drawn under either rule
    behind = tri.clone()
    behind[..., 2] = -behind[..., 2].abs()
    assert float(fo.soft_silhouette(behind, ft, *args).abs().max()) == 0.0


def test_sampler_round_prediction_is_the_recent_minimum():
    """hold_amd.sampler (speculative rounds): a low prediction costs one more flag read, a high one the whole call again -- so the
    next call launches the SMALLEST round count of the last PRED_WINDOW calls before it reads the flags."""
    from hold_amd.sampler import ErrorBoundSampler as S
    assert S.predict_rounds([]) == 0 and S.predict_rounds([3]) == 3
    assert S.predict_rounds([3, 2, 3, 2]) == 2 and S.predict_rounds([5, 5, 5, 5]) == 5
    s = S(3.0)
    for it, want in ((3, 3), (2, 2), (3, 2), (3, 2), (3, 2), (3, 3)):  # the 2 leaves the window after PRED_WINDOW = 4 calls
        s._recent_rounds = (s._recent_rounds + [it])[-S.PRED_WINDOW:]
        assert S.predict_rounds(s._recent_rounds) == want


def test_gap_report_charges_idle_time_to_the_launching_operator(tmp_path):
    """scripts/gap_report.py (the tool behind profiles/r05_c3_gap_report.txt) on a hand-made trace: three kernels with two idle gaps, the
    first launched from inside an autograd Function's backward, the second by a bare runtime call."""
    import json
    import subprocess
    import sys
    ev = [
        {"ph": "X", "cat": "kernel", "name": "k0", "ts": 100, "dur": 50, "args": {"correlation": 1}},
        {"ph": "X", "cat": "kernel", "name": "k1", "ts": 180, "dur": 20, "args": {"correlation": 2}},   # 30 us after k0 ended
        {"ph": "X", "cat": "kernel", "name": "k2", "ts": 500, "dur": 10, "args": {"correlation": 3}},   # 300 us after k1 ended
        {"ph": "X", "cat": "cuda_runtime", "name": "hipLaunchKernel", "ts": 90, "dur": 4, "tid": 7, "args": {"correlation": 1}},
        {"ph": "X", "cat": "cuda_runtime", "name": "hipLaunchKernel", "ts": 170, "dur": 4, "tid": 7, "args": {"correlation": 2}},
        {"ph": "X", "cat": "cuda_runtime", "name": "hipLaunchKernel", "ts": 490, "dur": 4, "tid": 9, "args": {"correlation": 3}},
        {"ph": "X", "cat": "cpu_op", "name": "_FieldFnBackward", "ts": 80, "dur": 120, "tid": 7},
        {"ph": "X", "cat": "cpu_op", "name": "aten::mul", "ts": 165, "dur": 12, "tid": 7},
    ]
    p = tmp_path / "t.json"
    p.write_text(json.dumps({"traceEvents": ev}))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gap_report.py"), str(p), "1"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    out = r.stdout
    assert "device events 3 (3 per step)" in out and "busy 0.08 ms" in out and "idle 0.33 ms per step" in out
    rows = [ln.split() for ln in out.splitlines() if ln.startswith("    ")]
    by = {ln[-1]: float(ln[0]) for ln in rows if len(ln) >= 4}
    assert abs(by["_FieldFnBackward"] - 0.030) < 1e-9        # the OUTERMOST operator around the launch, not aten::mul
    assert any("no host operator" in ln and " 0.300 " in ln for ln in out.splitlines())
    assert "300" in out.split("gaps over 100 us")[1]
