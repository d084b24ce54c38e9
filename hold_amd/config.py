"""Arithmetic selection for the MFMA kernels of the path.

``f32``    every matrix product on v_mfma_f32_32x32x2_f32 (true fp32 operands).
``f32x6``  every MFMA kernel of the path -- the sampler's fused SDF trunk (hold_fused_sdf_x6), the layer
           chains (hold_chain_x6), the single-layer GEMM (hold_gemm_nt_x6) and the weight gradients (hold_wgrad_x6) --
           splits every fp32 operand EXACTLY into three bf16 limbs and issues six of the nine limb products on
           v_mfma_f32_32x32x16_bf16 with fp32 accumulation (dropped terms <= 2^-23 relative: fp32-class results,
           measured 1.4e-6 max abs against the fp32 MFMA kernel on 524 288 points; the whole parity suite runs green
           in both modes at the same tolerances).

``f16x3``  (default) as ``f32x6``, except that the kernels that exist in the two-limb fp16 arithmetic use it: every fp32 operand, scaled
           by an exact power of two, is split into hi = RN_f16(x), lo = RN_f16(x - hi) and THREE of the four limb products
           (hi hi + hi lo + lo hi) are issued on v_mfma_f32_32x32x16_f16 with fp32 accumulation -- half the matrix
           instructions of ``f32x6`` at the same measured error against fp64 (csrc/rmlp_h3.hip: the sampler's SDF query
           hold_fused_sdf_h3 and the training forward trunk hold_trunk_h3).  Every other MFMA kernel
           runs its ``f32x6`` variant in this mode.

Set once per process with ``hold_amd.set_precision(...)`` (or HOLD_PRECISION in the environment of the Python host);
the C ABI itself is stateless -- the mode only decides WHICH entry point the host calls.
"""
from __future__ import annotations

import os

_MODES = ("f32", "f32x6", "f16x3")
DEFAULT_PRECISION = "f16x3"
_precision = os.environ.get("HOLD_PRECISION", DEFAULT_PRECISION)
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
    """the limb-split kernel families (bf16 three-limb; in mode f16x3 the kernels without an fp16 variant still run these)"""
    return _precision in ("f32x6", "f16x3")


def h3() -> bool:
    """two-limb fp16 variants where they exist (csrc/rmlp_h3.hip)"""
    return _precision == "f16x3"


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
