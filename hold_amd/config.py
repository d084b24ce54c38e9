"""Arithmetic selection for the MFMA kernels of the path.

``f32``    every matrix product on v_mfma_f32_32x32x2_f32 (true fp32 operands).
``f32x6``  (default) every MFMA kernel of the path -- the sampler's fused SDF trunk (hold_fused_sdf_x6), the layer
           chains (hold_chain_x6), the single-layer GEMM (hold_gemm_nt_x6) and the weight gradients (hold_wgrad_x6) --
           splits every fp32 operand EXACTLY into three bf16 limbs and issues six of the nine limb products on
           v_mfma_f32_32x32x16_bf16 with fp32 accumulation (dropped terms <= 2^-23 relative: fp32-class results,
           measured 1.4e-6 max abs against the fp32 MFMA kernel on 524 288 points; the whole parity suite runs green
           in both modes at the same tolerances).

Set once per process with ``hold_amd.set_precision(...)`` (or HOLD_PRECISION in the environment of the Python host);
the C ABI itself is stateless -- the mode only decides WHICH entry point the host calls.
"""
from __future__ import annotations

import os

_MODES = ("f32", "f32x6")
_precision = os.environ.get("HOLD_PRECISION", "f32x6")
if _precision not in _MODES:
    raise ValueError(f"HOLD_PRECISION must be one of {_MODES}, got {_precision!r}")


def set_precision(mode: str) -> None:
    global _precision
    if mode not in _MODES:
        raise ValueError(f"precision must be one of {_MODES}, got {mode!r}")
    _precision = mode


def precision() -> str:
    return _precision


def x6() -> bool:
    return _precision == "f32x6"


# ---- weight-pack invalidation -------------------------------------------------------------------------------------
# The re-laid-out weight packs of a node (hold_amd.field.pack_weights) are cached across forwards and rebuilt when a
# parameter changes.  In-place torch updates (torch.optim.*, load_state_dict, copy_) bump Tensor._version, which the
# cache key includes; FlatAdam updates the flat bucket from a HIP kernel through raw pointers, which torch cannot see,
# so it bumps this counter instead.  Anything else that writes parameters behind torch's back must call it too.
_weights_epoch = 0


def bump_weights_epoch() -> None:
    global _weights_epoch
    _weights_epoch += 1


def weights_epoch() -> int:
    return _weights_epoch
