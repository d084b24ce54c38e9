"""Thin python wrappers over hold_gemm_nt / hold_wgrad (include/hold_hip.h)."""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib, config
from ._lib import (EPI_DBWD, EPI_MUL_DRELU, EPI_MUL_DSIG, EPI_MUL_DSP, EPI_NONE, EPI_RELU, EPI_SIGMOID,  # noqa: F401
                   EPI_SOFTPLUS, GemmDesc, check, ptr, stream_ptr)


# when set to a list, every launch appends (start_event, end_event, algorithmic_flops, kernel_name, algorithmic_bytes); the
# events are recorded on torch's current stream, which is the stream the kernel is launched on.  algorithmic_bytes = the HBM
# bytes the launch has to move (operands in once, results out once; weights stay in L2), DESIGN.md section 4.
PROFILE = None


def _prof_begin():
    if PROFILE is None:
        return None
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def _prof_end(e0, flops, name, nbytes=0.0):
    if e0 is None:
        return
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    PROFILE.append((e0, e1, flops, name, float(nbytes)))


def _ld(t):
    assert t.stride(-1) == 1, "innermost dim must be contiguous"
    return t.stride(0)


R6_NONE, R6_RELU, R6_MASK = 0, 1, 2


def gemm_r6(A, wpack, out, *, K, bias=None, epi=R6_NONE, aux=None):
    """out[:, :256] = epi(A[:, :K] @ W.T + bias) with the register-resident kernel (hold_gemm_r6, csrc/rgemm.hip); wpack from
    field.pack_gemm_r6(W).  epi: R6_NONE, R6_RELU, R6_MASK (out = y * (aux > 0), no bias).  Split-precision arithmetic."""
    P = A.shape[0]
    L = _lib.lib()
    assert wpack.numel() * wpack.element_size() == L.hold_gemm_r6_pack_bytes(K), (wpack.shape, K)
    ldmax = max(_ld(A), _ld(out), _ld(aux) if aux is not None else 0)
    rows = max(128, ((1 << 32) // (4 * ldmax) - 256) // 128 * 128)  # 32-bit offsets inside the kernel: split by rows
    e0 = _prof_begin()
    for r0 in range(0, P, rows):
        n = min(P, r0 + rows) - r0
        check(L.hold_gemm_r6(ptr(A[r0:]), _ld(A), n, ptr(wpack), K, ptr(bias), int(epi), ptr(None if aux is None else aux[r0:]),
                             0 if aux is None else _ld(aux), ptr(out[r0:]), _ld(out), stream_ptr()), "hold_gemm_r6")
    _prof_end(e0, 2.0 * P * 256 * K, "rgemm_kernel", 4.0 * P * (K + 256 + (256 if aux is not None else 0)))
    return out


def gemm_h3(A, wpack_h3, c3, out, *, K, wpack_r6, bias=None, epi=R6_NONE, aux=None, amax_in=None, amax_floor=0.0, amax_out=None,
            bits_out=None, bits_in=None):
    """gemm_r6 in the two-limb fp16 arithmetic (hold_gemm_h3, csrc/rgemm_h3.hip): wpack_h3 / c3 from field.pack_gemm_h3 (c3 = a
    one-element device tensor, 1 / s_w).  Every row of A is scaled by its own power of two from max(amax_in[p], amax_floor)
    (amax_in [P]: the amax_out of the launch that produced A, or any upper bound; amax_floor: bound of the columns that launch did
    not write / of everything without amax_in); amax_out [P] receives the row maxima of out.  wpack_r6 = gemm_r6's stream of the
    same matrix: the conditional f32x6 fallback behind the overflow guard (kernels.h3_guard).
    bits_out ([P, 8] int32, epi = R6_RELU): receives the ReLU mask of `out` as one bit per element; bits_in (epi = R6_MASK): the mask
    is read from such a matrix instead of streaming `aux` (which stays the operand of the f32x6 fallback)."""
    from . import kernels as _k
    P = A.shape[0]
    L = _lib.lib()
    assert wpack_h3.numel() * wpack_h3.element_size() == L.hold_gemm_h3_pack_bytes(K) and wpack_h3.dtype == torch.float16
    assert wpack_r6.numel() * wpack_r6.element_size() == L.hold_gemm_r6_pack_bytes(K)
    assert c3.numel() == 1 and c3.dtype == torch.float32 and c3.is_cuda
    for t in (amax_in, amax_out):
        assert t is None or (t.numel() == P and t.dtype == torch.float32 and t.is_contiguous())
    for t in (bits_out, bits_in):
        assert t is None or (t.shape == (P, 8) and t.dtype == torch.int32 and t.is_contiguous())
    ldmax = max(_ld(A), _ld(out), _ld(aux) if aux is not None else 0)
    rows = max(128, ((1 << 32) // (4 * ldmax) - 256) // 128 * 128)  # 32-bit offsets inside the kernel: split by rows
    guard = _k.h3_guard(A.device)
    e0 = _prof_begin()
    for r0 in range(0, P, rows):
        n = min(P, r0 + rows) - r0
        check(L.hold_gemm_h3_bits(ptr(A[r0:]), _ld(A), n, ptr(wpack_h3), ptr(c3), K, ptr(bias), int(epi),
                                  ptr(None if aux is None else aux[r0:]), 0 if aux is None else _ld(aux), ptr(out[r0:]), _ld(out),
                                  ptr(None if amax_in is None else amax_in.reshape(-1)[r0:]), float(amax_floor),
                                  ptr(None if amax_out is None else amax_out.reshape(-1)[r0:]),
                                  ptr(None if bits_out is None else bits_out[r0:]), ptr(None if bits_in is None else bits_in[r0:]),
                                  ptr(guard), ptr(wpack_r6), stream_ptr()),
              "hold_gemm_h3")
    _prof_end(e0, 2.0 * P * 256 * K, "rgemm_h3_kernel",
              4.0 * P * (K + 256 + (256 if (aux is not None and bits_in is None) else 0) + 2
                         + (8 if (bits_in is not None or bits_out is not None) else 0)))
    return out


def gemm_nt(A, W, out, *, bias=None, epi=EPI_NONE, alpha=1.0, N=None, K=None, n_split=None, out_raw=None,
            aux1=None, aux2=None, out2=None, accumulate=False, r1_row=None, r1_col=None):
    """out[:, :N] = epi(alpha * A[:, :K] @ W[:N, :K].T + bias).  All tensors are 2-D fp32 CUDA views with unit
    inner stride; row strides (leading dimensions) are taken from the views, so callers can write
    straight into column slices of wider buffers."""
    P = A.shape[0]
    N = W.shape[0] if N is None else N
    K = W.shape[1] if K is None else K
    assert A.shape[1] >= K and W.shape[1] >= K and K % 4 == 0, (A.shape, W.shape, K)
    d = GemmDesc()
    d.A, d.lda = ptr(A), _ld(A)
    d.W, d.ldw = ptr(W), _ld(W)
    d.bias = ptr(bias)
    d.C, d.ldc = ptr(out), _ld(out)
    d.P, d.N, d.K = P, N, K
    d.alpha = float(alpha)
    d.epilogue = int(epi)
    d.n_split = N if n_split is None else int(n_split)
    if out_raw is not None:
        d.C2, d.ldc2 = ptr(out_raw), _ld(out_raw)
    if aux1 is not None:
        d.aux1, d.ldaux1 = ptr(aux1), _ld(aux1)
    if aux2 is not None:
        d.aux2, d.ldaux2 = ptr(aux2), _ld(aux2)
    if out2 is not None:
        d.out2, d.ldout2 = ptr(out2), _ld(out2)
    d.accumulate = 1 if accumulate else 0
    if r1_row is not None:  # y += r1_row[p] * r1_col[n] before the epilogue function
        assert r1_row.numel() >= P and r1_col.numel() >= N
        d.r1_row, d.ldr1, d.r1_col = ptr(r1_row), (r1_row.stride(0) if r1_row.dim() else 1), ptr(r1_col)
    e0 = _prof_begin()
    if config.x6():
        check(_lib.lib().hold_gemm_nt_x6(C.byref(d), stream_ptr()), "hold_gemm_nt_x6")
    else:
        check(_lib.lib().hold_gemm_nt(C.byref(d), stream_ptr()), "hold_gemm_nt")
    n_side = sum(x is not None for x in (aux1, aux2, out2)) + (1 if accumulate else 0)
    _prof_end(e0, 2.0 * P * N * K, "gemm_nt_kernel", 4.0 * P * (K + N * (1 + n_side)))
    return out


_ws_cache = {}


def _workspace(nfloats, device):
    key = (device.index,)
    w = _ws_cache.get(key)
    if w is None or w.numel() < nfloats:
        w = torch.empty(int(nfloats), dtype=torch.float32, device=device)
        _ws_cache[key] = w
    return w


def wgrad(R, X, dW, db=None, *, N=None, K=None, accumulate=False, splits=None):
    """dW[:N,:K] (+)= R[:, :N].T @ X[:, :K]; db[:N] (+)= R[:, :N].sum(0)."""
    P = R.shape[0]
    N = dW.shape[0] if N is None else N
    K = dW.shape[1] if K is None else K
    if splits is None:
        tiles = ((N + 127) // 128) * ((K + 127) // 128)
        splits = max(1, min(1024 // tiles, (P + 255) // 256))
        if config.x6() and 128 < N <= 256 and 256 <= K <= 320:  # whole-dW workgroups (csrc/wgrad_r6.hip): one per CU
            splits = max(1, min(256, (P + 255) // 256))
    L = _lib.lib()
    ws = _workspace(L.hold_wgrad_workspace_floats(N, K, splits), R.device)
    e0 = _prof_begin()
    fn = (L.hold_wgrad_h3 if (config.h3() and USE_H3_WGRAD) else L.hold_wgrad_x6) if config.x6() else L.hold_wgrad
    check(fn(ptr(R), _ld(R), ptr(X), _ld(X), P, N, K, ptr(dW), _ld(dW), ptr(db), 1 if accumulate else 0,
             splits, ptr(ws), stream_ptr()), "hold_wgrad")
    # bench.py prices the families by the matrix instructions they issue: the whole-dW shapes in mode f16x3 run wgrad_h3_kernel
    # (3 limb products per product; for K = 272 / 304 the 16 / 48-column tail runs the bf16 tile kernel: 6 % of the launch's FLOP)
    h3 = (config.h3() and USE_H3_WGRAD and 128 < N <= 256 and 256 <= K <= 320 and P % 16 == 0 and P >= 4096 and _ld(R) >= 256
          and _ld(X) >= K)
    _prof_end(e0, 2.0 * P * N * K, "wgrad_h3_kernel" if h3 else "wgrad_kernel", 4.0 * P * (N + K))
    return dW


USE_NARROW = os.environ.get("HOLD_NARROW", "1") != "0"
USE_H3_WGRAD = os.environ.get("HOLD_H3_WGRAD", "1") != "0"  # mode f16x3: the whole-dW weight gradients too (A/B switch)


def gemm_narrow(A, W, C, *, N=None, accumulate=False):
    """C[:, :N] (+)= A[:, :256] @ W[:N, :256].T for N <= 64 (hold_gemm_narrow_x6, csrc/rnarrow.hip): A streamed once, only
    ceil(N / 32) output tiles of MFMAs.  Falls back to hold_gemm_nt (same contract) outside its domain: not split precision,
    fewer than 4 096 points, operands it cannot address."""
    P = A.shape[0]
    N = C.shape[1] if N is None else N
    if not (USE_NARROW and config.x6() and P >= 4096 and 0 < N <= 64 and A.shape[1] >= 256 and W.shape[1] >= 256
            and _ld(A) % 4 == 0 and _ld(W) % 4 == 0 and A.data_ptr() % 16 == 0 and W.data_ptr() % 16 == 0):
        return gemm_nt(A, W, C, N=N, K=256, accumulate=accumulate)
    e0 = _prof_begin()
    check(_lib.lib().hold_gemm_narrow_x6(ptr(A), _ld(A), ptr(W), _ld(W), ptr(C), _ld(C), P, N, 1 if accumulate else 0,
                                         stream_ptr()), "hold_gemm_narrow_x6")
    _prof_end(e0, 2.0 * P * N * 256, "rnarrow_kernel", 4.0 * P * (256 + N * (2 if accumulate else 1)))
    return C


class WgradGroup:
    """Weight gradients over the same P points collected and issued as ONE launch (hold_wgrad_group_x6): a compute unit
    works on one (pair, share of the points), so the launch writes one partial tile per compute unit for all pairs
    together and one reduction replaces two per pair.  `add` has the meaning of `wgrad(..., K=256)`; the operands must
    stay untouched until `flush()`.  Pairs the grouped kernel does not take (not split precision, rows narrower than 256
    floats in memory, P not a multiple of 16, more than 24 pairs) go through `wgrad` one by one at flush time -- and so
    do all of them above MAX_P points: there a single launch's partial tiles are < 5 % of its operand bytes and 256
    workgroups streaming ONE pair measured 1 % faster than 18 pairs x 14 workgroups (183 vs 181 TF-eq at 1.6 M points,
    round 4 call 17), while at the 125 k points of the 1 280-ray step the grouped launch takes 4.5 % off the whole step."""
    MAX = 24
    MAX_P = 1 << 19
    # above MAX_P: the two pairs of one destination in one launch (128 workgroups each, half the partial tiles) -- measured
    # neutral at 1.6 M points (185.4 vs 184.5 TF-eq for the family, bench line within the noise; GPU call 24): off
    PAIRS_ABOVE = os.environ.get("HOLD_WGRAD_PAIRS", "0") == "1"

    def __init__(self):
        self.items = []

    def add(self, R, X, dW, db=None, *, N=None, accumulate=False):
        self.items.append((R, X, dW, db, dW.shape[0] if N is None else N, bool(accumulate)))

    def flush(self):
        items, self.items = self.items, []
        if not items:
            return
        P = items[0][0].shape[0]
        ok = (config.x6() and P % 16 == 0 and P >= 4096 and len(items) <= self.MAX
              and all(R.shape[0] == P and X.shape[0] == P and _ld(R) >= 256 and _ld(X) >= 256 and _ld(dW) >= 256
                      and 0 < N <= 256 for R, X, dW, db, N, acc in items))
        # pairs of one destination next to each other, destinations in order of first appearance
        order = {}
        for it in items:
            order.setdefault(it[2].data_ptr(), len(order))
        items.sort(key=lambda it: order[it[2].data_ptr()])
        if ok and P <= self.MAX_P:
            return self._launch(items, P)
        runs = [[it for it in items if order[it[2].data_ptr()] == d] for d in range(len(order))]
        for run in runs:
            if ok and self.PAIRS_ABOVE and len(run) > 1:  # the pairs of ONE destination together: half the partial tiles
                self._launch(run, P)
            else:
                for R, X, dW, db, N, acc in run:
                    wgrad(R, X, dW, db, N=N, K=256, accumulate=acc)

    def _launch(self, items, P):
        seen = {}
        for R, X, dW, db, N, acc in items:  # a destination accumulates if any of its pairs was asked to
            seen[dW.data_ptr()] = seen.get(dW.data_ptr(), False) or acc
        arr = (_lib.WgradItem * len(items))()
        for a, (R, X, dW, db, N, acc) in zip(arr, items):
            a.R, a.X, a.dW, a.db = ptr(R), ptr(X), ptr(dW), ptr(db)
            a.ldr, a.ldx, a.lddw, a.N = _ld(R), _ld(X), _ld(dW), N
            a.accumulate = 1 if seen[dW.data_ptr()] else 0
        L = _lib.lib()
        ws = _workspace(L.hold_wgrad_group_workspace_floats(), items[0][0].device)
        e0 = _prof_begin()
        fn = L.hold_wgrad_group_h3 if (config.h3() and USE_H3_WGRAD) else L.hold_wgrad_group_x6
        check(fn(arr, len(items), P, ptr(ws), stream_ptr()), "hold_wgrad_group")
        _prof_end(e0, 2.0 * P * 256 * sum(it[4] for it in items),
                  "wgrad_h3_kernel" if fn is L.hold_wgrad_group_h3 else "wgrad_kernel", 4.0 * P * 512 * len(items))


def wcolsum(X, out, *, weights=None, N=None, accumulate=False):
    """out[:N] (+)= (weights[:, None] * X[:, :N]).sum(0) -- deterministic two-pass column sums (hold_wcolsum)."""
    P = X.shape[0]
    N = out.shape[0] if N is None else N
    L = _lib.lib()
    ws = _workspace(L.hold_wcolsum_workspace_floats(N), X.device)
    check(L.hold_wcolsum(ptr(X), _ld(X), N, P, ptr(weights), ptr(out), 1 if accumulate else 0, ptr(ws), stream_ptr()),
          "hold_wcolsum")
    return out


def head3_fwd(A, W, bias, out, *, K=None, sigmoid=True):
    """out[:, :3] = act(A[:, :K] @ W[:3, :K].T + bias) -- the colour head as a streaming kernel (hold_head3_fwd)."""
    P = A.shape[0]
    K = W.shape[1] if K is None else K
    check(_lib.lib().hold_head3_fwd(ptr(A), _ld(A), ptr(W), _ld(W), ptr(bias), K, P, ptr(out), _ld(out),
                                    1 if sigmoid else 0, stream_ptr()), "hold_head3_fwd")
    return out


def head3_bwd(dy, R, W, rr, dW, db4, *, K=None, accumulate=False):
    """one pass over R: rr = (R > 0) * (dy[:, :3] @ W[:3, :K]); dW[:3, :K] (+)= dy[:, :3].T @ R; db4[:3] (+)= dy.sum(0)."""
    P = R.shape[0]
    K = W.shape[1] if K is None else K
    assert db4.numel() >= 4
    L = _lib.lib()
    ws = _workspace(L.hold_head3_workspace_floats(K), R.device)
    check(L.hold_head3_bwd(ptr(dy), _ld(dy), ptr(R), _ld(R), ptr(W), _ld(W), K, P, ptr(rr), _ld(rr), ptr(dW), _ld(dW),
                           ptr(db4), 1 if accumulate else 0, ptr(ws), stream_ptr()), "hold_head3_bwd")
    return rr
