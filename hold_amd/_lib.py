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
LIB_PATH = os.path.join(_HERE, "libholdhip.so")

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
    ]


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


def _declare(L):
    L.hold_abi_version.restype = C.c_int
    L.hold_gemm_nt.argtypes = [C.POINTER(GemmDesc), C.c_void_p]
    L.hold_gemm_nt.restype = C.c_int
    L.hold_wgrad_workspace_floats.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    L.hold_wgrad_workspace_floats.restype = C.c_int64
    L.hold_wgrad.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                             C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    L.hold_wgrad.restype = C.c_int
    for name, args in _EXTRA.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = C.c_int


# filled by the op modules below this one (name -> argtypes) before first lib() call
_EXTRA: dict = {}


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.dtype in (torch.float32, torch.int32, torch.int64, torch.uint8), (t.device, t.dtype)
    return C.c_void_p(t.data_ptr())


def check(code, what):
    if code != 0:
        raise RuntimeError(f"libholdhip: {what} failed with code {code}")
