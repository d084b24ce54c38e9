"""ctypes binding of libholdhip.so (the C ABI declared in include/hold_hip.h).

The library is built in-tree by ``hold_amd.build.build()`` (hipcc --offload-arch=gfx950).  There is
NO fallback: if the shared object is missing or a symbol cannot be resolved, importing the
product path raises.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# HOLD_LIB: load another build of the same ABI (the developer build libholdhip_dev.so, `HOLD_DEV=1 python -m hold_amd.build`)
LIB_PATH = os.environ.get("HOLD_LIB") or os.path.join(_HERE, "libholdhip.so")

EPI_NONE, EPI_SOFTPLUS, EPI_RELU, EPI_SIGMOID, EPI_MUL_DSP, EPI_MUL_DRELU, EPI_DBWD, EPI_MUL_DSIG = range(8)


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("lda", C.c_int32),
        ("W", C.c_void_p), ("ldw", C.c_int32),
        ("bias", C.c_void_p),
        ("C", C.c_void_p), ("ldc", C.c_int32),
        ("P", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("alpha", C.c_float),
        ("epilogue", C.c_int32),
        ("n_split", C.c_int32),
        ("C2", C.c_void_p), ("ldc2", C.c_int32),
        ("aux1", C.c_void_p), ("ldaux1", C.c_int32),
        ("aux2", C.c_void_p), ("ldaux2", C.c_int32),
        ("out2", C.c_void_p), ("ldout2", C.c_int32),
        ("accumulate", C.c_int32),
        ("r1_row", C.c_void_p), ("ldr1", C.c_int32),
        ("r1_col", C.c_void_p),
    ]


class ChainDesc(C.Structure):
    _fields_ = [
        ("P", C.c_int64),
        ("mode", C.c_int32), ("n_layers", C.c_int32), ("first_chunks", C.c_int32), ("skip_layer", C.c_int32),
        ("in_", C.c_void_p), ("ld_in", C.c_int32),
        ("side", C.c_void_p), ("ld_side", C.c_int32),
        ("wpack", C.c_void_p), ("ld", C.c_int32),
        ("bias", C.c_void_p * 8), ("aux1", C.c_void_p * 8), ("aux2", C.c_void_p * 8),
        ("out", C.c_void_p * 8), ("out2", C.c_void_p * 8),
        ("skip_out", C.c_int32),
    ]


class WnLayer(C.Structure):
    _fields_ = [("v", C.c_void_p), ("g", C.c_void_p), ("w", C.c_void_p), ("dw", C.c_void_p), ("dv", C.c_void_p),
                ("dg", C.c_void_p), ("rows", C.c_int32), ("cols", C.c_int32), ("ldv", C.c_int32), ("ldw", C.c_int32)]


class WnDesc(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("accumulate", C.c_int32), ("layers", WnLayer * 16)]


class WgradItem(C.Structure):  # hold_wgrad_item
    _fields_ = [("R", C.c_void_p), ("X", C.c_void_p), ("dW", C.c_void_p), ("db", C.c_void_p),
                ("ldr", C.c_int32), ("ldx", C.c_int32), ("lddw", C.c_int32), ("N", C.c_int32),
                ("accumulate", C.c_int32), ("reserved", C.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hold_amd has no CPU / eager fallback)")
        _lib = C.CDLL(LIB_PATH)
        _declare(_lib)
    return _lib


class CompositeDesc(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_int32), ("S", C.c_int32), ("n_rays", C.c_int64),
        ("z", C.c_void_p * 3), ("sdf", C.c_void_p * 3), ("color", C.c_void_p * 3), ("normal", C.c_void_p * 3),
        ("ldc", C.c_int32 * 3), ("ldn", C.c_int32 * 3), ("class_id", C.c_int32 * 3),
        ("beta", C.c_float * 3),
        ("out_node", C.c_void_p * 3), ("out_comp", C.c_void_p), ("out_sem", C.c_void_p), ("out_w", C.c_void_p),
        ("out_zmerge", C.c_void_p),
        ("d_node", C.c_void_p * 3), ("d_comp", C.c_void_p), ("d_sem", C.c_void_p),
        ("d_sdf", C.c_void_p * 3), ("d_color", C.c_void_p * 3), ("d_normal", C.c_void_p * 3), ("d_beta", C.c_void_p),
        ("out_w_node", C.c_void_p * 3),
    ]


class LossNodes(C.Structure):
    _fields_ = [("mask_prob", C.c_void_p * 3), ("off", C.c_void_p * 3), ("d_mask", C.c_void_p * 3)]


class ManoModel(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("v_template", "shapedirs", "posedirs", "J_regressor", "parents",
                                          "lbs_weights", "pose_mean", "tfs_c_inv")]


_P, _I, _L, _F = C.c_void_p, C.c_int32, C.c_int64, C.c_float
# developer-build-only symbols (HOLD_DEV=1 at build time): declared when present
DEV_SIGNATURES = {
    "hold_diag_mfma_peak": [_P, _I, _I, _I, _P],
    "hold_diag_mfma_lds": [_P, _P, _I, _I, _I, _P],
}
# every exported symbol of include/hold_hip.h with its argument types (stream is always last)
SIGNATURES = {
    "hold_gemm_nt": [C.POINTER(GemmDesc), _P],
    "hold_gemm_nt_x6": [C.POINTER(GemmDesc), _P],
    "hold_wgrad": [_P, _I, _P, _I, _I, _I, _I, _P, _I, _P, _I, _I, _P, _P],
    "hold_wgrad_x6": [_P, _I, _P, _I, _I, _I, _I, _P, _I, _P, _I, _I, _P, _P],
    "hold_gemm_narrow_x6": [_P, _I, _P, _I, _P, _I, _L, _I, _I, _P],
    "hold_wgrad_group_x6": [C.POINTER(WgradItem), _I, _L, _P, _P],
    "hold_wgrad_group_h3": [C.POINTER(WgradItem), _I, _L, _P, _P],
    "hold_wgrad_h3": [_P, _I, _P, _I, _I, _I, _I, _P, _I, _P, _I, _I, _P, _P],
    "hold_ray_points": [_P, _P, _P, _I, _I, _L, _P, _I, _P],
    "hold_embed_fwd": [_P, _I, _I, _I, _P, _L, _P, _I, _P, _I, _P, _I, _L, _P],
    "hold_embed_bwd": [_P, _I, _I, _P, _L, _P, _I, _P, _I, _I, _P],
    "hold_embed_bwd2": [_P, _I, _I, _P, _L, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P],
    "hold_knn_invlbs_fwd": [_P, _I, _L, _L, _P, _L, _I, _P, _P, _P, _P, _I, _P],
    "hold_invskin_fwd": [_P, _I, _L, _L, _P, _P, _I, _P, _I, _P],
    "hold_raygen": [_P, _P, _P, _I, _L, _L, _P, _P, _P],
    "hold_skin_fwd": [_P, _I, _L, _L, _P, _P, _I, _P, _I, _P],
    "hold_invskin_bwd": [_P, _I, _P, _P, _I, _L, _L, _P, _I, _P, _P],
    "hold_normal_fwd": [_P, _I, _P, _P, _I, _L, _L, _P, _I, _P],
    "hold_normal_bwd": [_P, _I, _P, _P, _I, _L, _L, _P, _I, _P, _I, _P, _P],
    "hold_frame_colsum": [_P, _I, _I, _I, _L, _L, _P, _P],
    "hold_frame_bcast": [_P, _I, _L, _L, _P, _I, _I, _P],
    "hold_copy_cols": [_P, _I, _P, _I, _I, _L, _I, _P],
    "hold_bg_points": [_P, _P, _P, _I, _L, _F, _P, _I, _P],
    "hold_gemm_r6": [_P, _I, _L, _P, _I, _P, _I, _P, _I, _P, _I, _P],
    "hold_gemm_r6_if": [_P, _I, _L, _P, _I, _P, _I, _P, _I, _P, _I, _P, _P, _P],
    "hold_gemm_h3": [_P, _I, _L, _P, _P, _I, _P, _I, _P, _I, _P, _I, _P, _F, _P, _P, _P, _P],
    "hold_gemm_h3_bits": [_P, _I, _L, _P, _P, _I, _P, _I, _P, _I, _P, _I, _P, _F, _P, _P, _P, _P, _P, _P],
    "hold_weight_norm_fwd": [C.POINTER(WnDesc), _P],
    "hold_weight_norm_bwd": [C.POINTER(WnDesc), _P],
    "hold_rowdot": [_P, _I, _P, _I, _F, _P, _L, _P, _I, _P],
    "hold_seed_dsp": [_P, _I, _P, _I, _L, _P, _I, _P],
    "hold_colsum": [_P, _I, _I, _L, _P, _P],
    "hold_wcolsum": [_P, _I, _I, _L, _P, _P, _I, _P, _P],
    "hold_head3_fwd": [_P, _I, _P, _I, _P, _I, _L, _P, _I, _I, _P],
    "hold_head3_bwd": [_P, _I, _P, _I, _P, _I, _I, _L, _P, _I, _P, _I, _P, _I, _P, _P],
    "hold_sampler_init": [_P, _P, _L, _F, _F, _I, _F, _P, _P, _I, _P, _P, _P, _P],
    "hold_sampler_beta": [_P, _P, _I, _I, _L, _P, _P, _I, _P, _F, _F, _I, _P, _P],
    "hold_sampler_sample": [_P, _P, _I, _I, _L, _P, _I, _F, _P, _L, _I, _P, _P, _P],
    "hold_sampler_final": [_P, _I, _P, _I, _P, _I, _P, _F, _L, _P, _I, _P],
    "hold_composite_fwd": [C.POINTER(CompositeDesc), _P],
    "hold_composite_bwd": [C.POINTER(CompositeDesc), _P],
    "hold_bg_composite_fwd": [_P, _P, _P, _I, _I, _L, _P, _P, _P],
    "hold_bg_composite_bwd": [_P, _P, _P, _I, _I, _L, _P, _P, _P, _P],
    "hold_silhouette_fwd": [_P, _I, _I, _P, _I, _F, _F, _F, _F, _I, _I, _F, _F, _P, _P, _P],
    "hold_silhouette_bwd": [_P, _I, _I, _P, _I, _F, _F, _F, _F, _I, _I, _F, _F, _P, _P, _P, _P, _P],
    "hold_silhouette_max_faces": [_P, _I, _I, _P, _I, _F, _F, _F, _F, _I, _I, _F, _P, _P, _P],
    "hold_knn1_fwd": [_P, _I, _I, _P, _I, _P, _P, _P],
    "hold_knn1_bwd": [_P, _I, _I, _P, _I, _P, _P, _P, _P, _P],
    "hold_fused_sdf": [_P, _I, _L, _P, _P, _P, _F, _P, _P, _I, _P],
    "hold_chain": [C.POINTER(ChainDesc), _P],
    "hold_chain_x6": [C.POINTER(ChainDesc), _P],
    "hold_chain_r6": [C.POINTER(ChainDesc), _P],
    "hold_chain_r6_if": [C.POINTER(ChainDesc), _P, _P],
    "hold_chain_h3": [C.POINTER(ChainDesc), _P, _P, _P, _P],
    "hold_mesh_sdf": [_P, _I, _L, _P, _I, _I, _P, _I, _F, _P, _P, _P],
    "hold_ray_off_surface": [_P, _I, _L, _I, _P, _I, _F, _F, _F, _F, _F, _P, _P, _P, _P, _P, _P],
    "hold_pixel_loss_fwd": [_P, _P, _P, _P, _L, _I, C.POINTER(LossNodes), _P, _P, _P],
    "hold_pixel_loss_bwd": [_P, _P, _P, _P, _L, _I, C.POINTER(LossNodes), _P, _P, _P, _P],
    "hold_sumsq": [_P, _L, _P, _I, _P, _P],
    "hold_adam_step": [_P, _P, _P, _P, _L, _L, _F, _F, _F, _F, _F, _I, _F, _F, _P, _P],
    "hold_mt_classify": [_P, _I, _F, _P, _P, _P, _P, _P],
    "hold_mt_vertices": [_P, _I, _F, _F, _F, _F, _F, _P, _P, _P, _P],
    "hold_mt_triangles": [_P, _I, _F, _P, _P, _P, _P, _P, _P, _P, _P],
    "hold_fused_sdf_x6": [_P, _I, _L, _P, _P, _P, _F, _P, _P, _I, _P],
    "hold_fused_sdf_r6": [_P, _I, _L, _P, _P, _P, _P, _P, _P, _I, _P],
    "hold_trunk_r6": [_P, _I, _L, _P, _P, _P, C.POINTER(C.c_void_p), _I, _P],
    "hold_alive_count": [_P, _I, _L, _F, _P, _P],
    "hold_alive_index": [_P, _I, _L, _F, _P, _P, _P],
    "hold_alive_mask": [_P, _I, _L, _F, _P, _P],
    "hold_fused_sdf_h3": [_P, _I, _L, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P],
    "hold_trunk_h3": [_P, _I, _L, _P, _P, _P, _P, C.POINTER(C.c_void_p), _I, _P, _P, _P, _P],
    "hold_fused_sdf_r6_if": [_P, _I, _L, _P, _P, _P, _P, _P, _P, _I, _P, _P],
    "hold_trunk_r6_if": [_P, _I, _L, _P, _P, _P, C.POINTER(C.c_void_p), _I, _P, _P],
    "hold_mano_lbs_fwd": [C.POINTER(ManoModel), _I, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "hold_mano_lbs_bwd": [C.POINTER(ManoModel), _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
}


def _declare(L):
    L.hold_abi_version.restype = C.c_int
    L.hold_wgrad_workspace_floats.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    L.hold_wgrad_workspace_floats.restype = C.c_int64
    L.hold_wgrad_group_workspace_floats.argtypes = []
    L.hold_wgrad_group_workspace_floats.restype = C.c_int64
    L.hold_reduce_workspace_floats.restype = C.c_int64
    L.hold_fused_sdf_pack_floats.restype = C.c_int64
    L.hold_fused_sdf_x6_pack_bytes.restype = C.c_int64
    L.hold_trunk_r6_pack_bytes.restype = C.c_int64
    L.hold_trunk_h3_pack_bytes.restype = C.c_int64
    L.hold_alive_blocks.restype = C.c_int64
    L.hold_alive_blocks.argtypes = [C.c_int64]
    L.hold_trunk_h3_act_scale.restype = C.c_float
    L.hold_gemm_r6_pack_bytes.restype = C.c_int64
    L.hold_gemm_r6_pack_bytes.argtypes = [C.c_int32]
    L.hold_gemm_h3_pack_bytes.restype = C.c_int64
    L.hold_gemm_h3_pack_bytes.argtypes = [C.c_int32]
    L.hold_chain_r6_pack_bytes.restype = C.c_int64
    L.hold_chain_h3_pack_bytes.restype = C.c_int64
    L.hold_chain_pack_floats.argtypes = [C.c_int32, C.c_int32]
    L.hold_chain_pack_floats.restype = C.c_int64
    L.hold_chain_x6_pack_bytes.argtypes = [C.c_int32, C.c_int32]
    L.hold_chain_x6_pack_bytes.restype = C.c_int64
    L.hold_head3_workspace_floats.argtypes = [C.c_int32]
    L.hold_head3_workspace_floats.restype = C.c_int64
    L.hold_wcolsum_workspace_floats.argtypes = [C.c_int32]
    L.hold_wcolsum_workspace_floats.restype = C.c_int64
    L.hold_silhouette_workspace_floats.argtypes = [C.c_int32, C.c_int32]
    L.hold_silhouette_workspace_floats.restype = C.c_int64
    for name, args in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError here = symbol missing from the shared object
        fn.argtypes = args
        fn.restype = C.c_int
    for name, args in DEV_SIGNATURES.items():
        if hasattr(L, name):
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = C.c_int


CALLS = 0  # number of C-ABI kernel entry points invoked so far (bench.py reports calls per step)


def call(name, *args):
    """invoke an exported entry point on the current torch stream; raise on a non-zero code."""
    global CALLS
    CALLS += 1
    L = lib()
    code = getattr(L, name)(*args, stream_ptr())
    if code != 0:
        raise RuntimeError(f"libholdhip: {name} failed with code {code}")


def h2d(t, device):
    """host tensor -> device without draining the stream: a pageable-memory copy is synchronous (it waits for everything
    queued before it), a copy out of pinned memory is ordered on the stream like a kernel.  torch's caching host allocator
    keeps the pinned block alive until the copy has run."""
    if torch.device(device).type == "cuda":
        return t.pin_memory().to(device, non_blocking=True)
    return t.to(device)


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_GET_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr():
    """torch's current stream of the current device as a hipStream_t.  A 1 280-ray step asks ~250 times: the raw getters cost 0.1 us,
    `torch.cuda.current_stream().cuda_stream` 2.9 us (profiles/r05_host_costs.txt) -- same stream either way."""
    if _RAW_STREAM is not None and _GET_DEVICE is not None:
        return C.c_void_p(_RAW_STREAM(_GET_DEVICE()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.dtype in (torch.float32, torch.int32, torch.int64, torch.uint8, torch.int8, torch.bool, torch.bfloat16, torch.float16), (t.device, t.dtype)
    return C.c_void_p(t.data_ptr())


def check(code, what):
    global CALLS
    CALLS += 1
    if code != 0:
        raise RuntimeError(f"libholdhip: {what} failed with code {code}")
