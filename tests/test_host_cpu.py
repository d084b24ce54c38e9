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
    declared = set(re.findall(r"\b(?:int|int64_t)\s+(hold_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 27
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert declared - {"hold_abi_version", "hold_wgrad_workspace_floats", "hold_fused_sdf_pack_floats", "hold_chain_pack_floats",
                       "hold_silhouette_workspace_floats"} == set(_lib.SIGNATURES)
    assert L.hold_abi_version() == 1


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
