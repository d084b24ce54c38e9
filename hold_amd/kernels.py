"""Python wrappers (ctypes) for the non-GEMM entry points of libholdhip.so (include/hold_hip.h).

All tensors are fp32 CUDA tensors with unit inner stride; row strides are forwarded as leading
dimensions so kernels can read/write column slices of wider buffers in place.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import CompositeDesc, call, ptr


def _ld(t):
    if t.dim() == 1:
        return 1
    assert t.stride(-1) == 1
    return t.stride(0)


def raygen(uv, pose, intrinsics):
    """uv [B,Pn,2], pose [B,4,4], intrinsics [B,3|4,3|4] -> (ray_dirs [B*Pn,3], cam_loc [B*Pn,3])."""
    B, Pn, _ = uv.shape
    uv, pose, intr = uv.contiguous().float(), pose.contiguous().float(), intrinsics.contiguous().float()
    assert pose.shape[1:] == (4, 4) and intr.shape[1] == intr.shape[2]
    dirs = torch.empty(B * Pn, 3, device=uv.device)
    cam = torch.empty(B * Pn, 3, device=uv.device)
    call("hold_raygen", ptr(uv), ptr(pose), ptr(intr), intr.shape[1], B * Pn, Pn, ptr(dirs), ptr(cam))
    return dirs, cam


def ray_points(cam_loc, ray_dirs, z, S, out):
    call("hold_ray_points", ptr(cam_loc), ptr(ray_dirs), ptr(z), _ld(z), S, cam_loc.shape[0], ptr(out), _ld(out))


def embed_fwd(x, d_in, L, P, out, out2=None, barf_w=None, cond=None, pts_per_frame=1):
    call("hold_embed_fwd", ptr(x), _ld(x), d_in, L, ptr(barf_w), P, ptr(out), _ld(out), ptr(out2),
         _ld(out2) if out2 is not None else 0, ptr(cond), 0 if cond is None else cond.shape[1], pts_per_frame)


def embed_bwd(x, L, P, ge, gx, barf_w=None, accumulate=False):
    call("hold_embed_bwd", ptr(x), _ld(x), L, ptr(barf_w), P, ptr(ge), _ld(ge), ptr(gx), _ld(gx), int(accumulate))


def embed_bwd2(x, L, P, ge, gbar, gebar, xbar=None, barf_w=None, gebar2=None):
    """gebar2: optional second copy of gebar's columns; may be the buffer `ge` itself lives in (see hold_hip.h)"""
    call("hold_embed_bwd2", ptr(x), _ld(x), L, ptr(barf_w), P, ptr(ge), _ld(ge), ptr(gbar), _ld(gbar), ptr(gebar),
         _ld(gebar), ptr(xbar), _ld(xbar) if xbar is not None else 0, ptr(gebar2),
         _ld(gebar2) if gebar2 is not None else 0)


def knn_invlbs(x, P, pts_per_frame, verts, skin_w, tfs=None, w_out=None, xc_out=None):
    """verts [B,V,3] (per frame) or [1,V,3] / [V,3] (shared)."""
    v = verts if verts.dim() == 3 else verts[None]
    stride = 0 if v.shape[0] == 1 else v.stride(0)
    call("hold_knn_invlbs_fwd", ptr(x), _ld(x), P, pts_per_frame, ptr(v), stride, v.shape[1], ptr(skin_w), ptr(tfs),
         ptr(w_out), ptr(xc_out), _ld(xc_out) if xc_out is not None else 0)


def invskin_fwd(x, P, pts_per_frame, w, tfs, n_bones, xc):
    call("hold_invskin_fwd", ptr(x), _ld(x), P, pts_per_frame, ptr(w), ptr(tfs), n_bones, ptr(xc), _ld(xc))


def skin_fwd(x, P, pts_per_frame, w, tfs, n_bones, xd):
    call("hold_skin_fwd", ptr(x), _ld(x), P, pts_per_frame, ptr(w), ptr(tfs), n_bones, ptr(xd), _ld(xd))


def invskin_bwd(xc, w, tfs, n_bones, P, pts_per_frame, xcbar, dtfs):
    call("hold_invskin_bwd", ptr(xc), _ld(xc), ptr(w), ptr(tfs), n_bones, P, pts_per_frame, ptr(xcbar), _ld(xcbar),
         ptr(dtfs))


def normal_fwd(g, w, tfs, n_bones, P, pts_per_frame, n_out):
    call("hold_normal_fwd", ptr(g), _ld(g), ptr(w), ptr(tfs), n_bones, P, pts_per_frame, ptr(n_out), _ld(n_out))


def normal_bwd(g, w, tfs, n_bones, P, pts_per_frame, nbar, gbar, dtfs):
    call("hold_normal_bwd", ptr(g), _ld(g), ptr(w), ptr(tfs), n_bones, P, pts_per_frame, ptr(nbar), _ld(nbar),
         ptr(gbar), _ld(gbar), ptr(dtfs))


def frame_colsum(X, col0, ncols, P, pts_per_frame, out):
    call("hold_frame_colsum", ptr(X), _ld(X), col0, ncols, P, pts_per_frame, ptr(out))


def frame_bcast(src, P, pts_per_frame, out, col0):
    call("hold_frame_bcast", ptr(src), src.shape[1], P, pts_per_frame, ptr(out), _ld(out), col0)


def copy_cols(src, dst, ncols, P, accumulate=False):
    call("hold_copy_cols", ptr(src), _ld(src), ptr(dst), _ld(dst), ncols, P, int(accumulate))


def bg_points(cam_loc, ray_dirs, depth, S, R, out):
    call("hold_bg_points", ptr(cam_loc), ptr(ray_dirs), ptr(depth), S, cam_loc.shape[0], float(R), ptr(out), _ld(out))


def rowdot(A, w, K, b, P, out):
    """out[p] = A[p][:K] . w + b; b a Python float or a one-element device tensor (read by the kernel: no host copy)"""
    if torch.is_tensor(b):
        call("hold_rowdot", ptr(A), _ld(A), ptr(w), K, 0.0, ptr(b), P, ptr(out), _ld(out))
    else:
        call("hold_rowdot", ptr(A), _ld(A), ptr(w), K, float(b), None, P, ptr(out), _ld(out))


def fused_sdf(xc, P, wpack, bias8, w8, b8, barf_w, out_sdf):
    assert wpack.numel() == _lib.lib().hold_fused_sdf_pack_floats()
    from . import gemm as _g
    e0 = _g._prof_begin()
    call("hold_fused_sdf", ptr(xc), _ld(xc), P, ptr(wpack), ptr(bias8), ptr(w8), float(b8), ptr(barf_w), ptr(out_sdf),
         _ld(out_sdf))
    _g._prof_end(e0, 2.0 * P * (40 * 256 + 6 * 65536 + 217 * 256 + 256), "fused_sdf_kernel", 20.0 * P)


def fused_sdf_x6(xc, P, wpack_x6, bias8, w8, b8, barf_w, out_sdf):
    """split-precision (3 bf16 limbs x 6 products, fp32 accumulate) variant of fused_sdf; wpack_x6 from field.pack_x6."""
    assert wpack_x6.numel() * wpack_x6.element_size() == _lib.lib().hold_fused_sdf_x6_pack_bytes()
    from . import gemm as _g
    e0 = _g._prof_begin()
    call("hold_fused_sdf_x6", ptr(xc), _ld(xc), P, ptr(wpack_x6), ptr(bias8), ptr(w8), float(b8), ptr(barf_w),
         ptr(out_sdf), _ld(out_sdf))
    _g._prof_end(e0, 2.0 * P * (40 * 256 + 6 * 65536 + 217 * 256 + 256), "fused_sdf_kernel", 20.0 * P)


def fused_sdf_r6(xc, P, wpack_r6, bias8, w8, b8, barf_w, out_sdf):
    """register-resident trunk (csrc/rmlp.hip) with the contract of fused_sdf_x6; wpack_r6 from field.pack_r6; b8 = a
    one-element DEVICE tensor (the kernel reads the scalar itself)."""
    assert wpack_r6.numel() * wpack_r6.element_size() == _lib.lib().hold_trunk_r6_pack_bytes()
    assert b8.is_cuda and b8.numel() == 1 and b8.dtype == torch.float32
    from . import gemm as _g
    e0 = _g._prof_begin()
    call("hold_fused_sdf_r6", ptr(xc), _ld(xc), P, ptr(wpack_r6), ptr(bias8), ptr(w8), ptr(b8), ptr(barf_w),
         ptr(out_sdf), _ld(out_sdf))
    _g._prof_end(e0, 2.0 * P * (40 * 256 + 6 * 65536 + 217 * 256 + 256), "fused_sdf_kernel", 20.0 * P)


def alive_index(sdf, P, beta):
    """ordered indices (int64, device) of the samples that can contribute anything -- to a pixel or to a gradient -- given the
    node's density beta (csrc/compact.hip: Laplace density != 0 or exp(-|sdf| / beta) != 0, the compositor's expressions); None
    when every sample is live.  ONE host read (the live count sizes the launches that follow)."""
    L = _lib.lib()
    nb = int(L.hold_alive_blocks(P))
    counts = torch.empty(nb, dtype=torch.int32, device=sdf.device)
    ld = sdf.stride(0) if sdf.dim() > 1 else 1
    call("hold_alive_count", ptr(sdf), ld, P, float(beta), ptr(counts))
    off = torch.zeros(nb + 1, dtype=torch.int64, device=sdf.device)
    torch.cumsum(counts, 0, out=off[1:])
    n_live = int(off[-1])  # host read
    if n_live == P:
        return None
    idx = torch.empty(n_live, dtype=torch.int64, device=sdf.device)
    if n_live:
        call("hold_alive_index", ptr(sdf), ld, P, float(beta), ptr(off), ptr(idx))
    return idx


def alive_mask(sdf, P, beta):
    """[P] uint8, 1 = live (the predicate of alive_index); no host read"""
    mask = torch.empty(P, dtype=torch.uint8, device=sdf.device)
    call("hold_alive_mask", ptr(sdf), sdf.stride(0) if sdf.dim() > 1 else 1, P, float(beta), ptr(mask))
    return mask


_H3_GUARD = {}


def h3_guard(device):
    """the overflow guard of the f16x3 kernels (include/hold_hip.h): 4 device words per (device, STREAM) -- the protocol of the
    conditional f32x6 launch assumes that the launches sharing a guard are ordered, i.e. on one stream -- zeroed once: [0] the flag a
    launch sets when a scaled operand left fp16's range, [2] how many launches were recomputed in f32x6 because of it (the
    conditional launch behind every f16x3 call; no host read on the path)"""
    dev = torch.device(device)
    k = (str(dev), torch.cuda.current_stream(dev).cuda_stream)
    g = _H3_GUARD.get(k)
    if g is None:
        g = _H3_GUARD[k] = torch.zeros(4, dtype=torch.int32, device=dev)
    return g


def h3_overflow_count(device):
    """launches of the f16x3 kernels on `device` so far (all streams) whose result came from the f32x6 fallback (host reads)"""
    d = str(torch.device(device))
    return sum(int(g[2]) for (kd, _), g in _H3_GUARD.items() if kd == d)


def fused_sdf_h3(xc, P, wpack_h3, bias8_scaled, c3, w8, b8, barf_w, out_sdf, wpack_r6=None, bias8=None):
    """the sampler's SDF query in the two-limb fp16 arithmetic (csrc/rmlp_h3.hip): wpack_h3 / bias8_scaled / c3 from
    field.pack_weights in mode f16x3 (field.pack_h3); otherwise the contract of fused_sdf_r6.  wpack_r6 / bias8 (fused_sdf_r6's
    operands): the f32x6 kernel follows as a conditional launch that recomputes the query iff an activation overflowed fp16."""
    L = _lib.lib()
    assert wpack_h3.numel() * wpack_h3.element_size() == L.hold_trunk_h3_pack_bytes() and wpack_h3.dtype == torch.float16
    assert b8.is_cuda and b8.numel() == 1 and b8.dtype == torch.float32 and c3.numel() == 8 and c3.dtype == torch.float32
    from . import field as _f, gemm as _g
    assert L.hold_trunk_h3_act_scale() == _f.H3_ACT_SCALE
    e0 = _g._prof_begin()
    assert (wpack_r6 is None) == (bias8 is None)
    call("hold_fused_sdf_h3", ptr(xc), _ld(xc), P, ptr(wpack_h3), ptr(bias8_scaled), ptr(c3), ptr(w8), ptr(b8),
         ptr(barf_w), ptr(out_sdf), _ld(out_sdf), ptr(h3_guard(xc.device)), ptr(wpack_r6), ptr(bias8))
    _g._prof_end(e0, 2.0 * P * (40 * 256 + 6 * 65536 + 217 * 256 + 256), "fused_sdf_kernel", 20.0 * P)


def _max_rows(ld):
    """largest row count per launch of the kernels with 32-bit byte offsets: they reject (P + 128) * ld * 4 >= 2^32"""
    return ((1 << 32) // (4 * ld) - 129) // 128 * 128


_TRUNK_MAX_ROWS = _max_rows(256)


def trunk_r6(xc, P, wpack_r6, bias8, barf_w, h):
    """training forward trunk: h[0..7] [P,256] = softplus outputs of lin0..lin7 (h[3][:, 217:] = the embedding)."""
    import ctypes as C
    assert wpack_r6.numel() * wpack_r6.element_size() == _lib.lib().hold_trunk_r6_pack_bytes()
    from . import gemm as _g
    ld = h[0].stride(0)
    assert all(t.stride(0) == ld and t.stride(1) == 1 for t in h)
    step = min(_TRUNK_MAX_ROWS, _max_rows(ld))
    for r0 in range(0, P, step):
        n = min(step, P - r0)
        arr = (C.c_void_p * 8)(*[t[r0:].data_ptr() for t in h])
        e0 = _g._prof_begin()
        call("hold_trunk_r6", ptr(xc[r0:]), _ld(xc), n, ptr(wpack_r6), ptr(bias8), ptr(barf_w), arr, ld)
        _g._prof_end(e0, 2.0 * n * (40 * 256 + 6 * 65536 + 217 * 256), "trunk_r6_kernel", n * (16.0 + 8 * 1024))


def trunk_h3(xc, P, wpack_h3, bias8_scaled, c3, barf_w, h, wpack_r6=None, bias8=None):
    """training forward trunk in the two-limb fp16 arithmetic: the contract of trunk_r6 (h in fp32, unscaled); wpack_r6 /
    bias8: the conditional f32x6 fallback, as for fused_sdf_h3"""
    assert (wpack_r6 is None) == (bias8 is None)
    import ctypes as C
    assert wpack_h3.numel() * wpack_h3.element_size() == _lib.lib().hold_trunk_h3_pack_bytes()
    from . import gemm as _g
    ld = h[0].stride(0)
    assert all(t.stride(0) == ld and t.stride(1) == 1 for t in h)
    step = min(_TRUNK_MAX_ROWS, _max_rows(ld))
    for r0 in range(0, P, step):
        n = min(step, P - r0)
        arr = (C.c_void_p * 8)(*[t[r0:].data_ptr() for t in h])
        e0 = _g._prof_begin()
        call("hold_trunk_h3", ptr(xc[r0:]), _ld(xc), n, ptr(wpack_h3), ptr(bias8_scaled), ptr(c3), ptr(barf_w), arr, ld,
             ptr(h3_guard(xc.device)), ptr(wpack_r6), ptr(bias8))
        _g._prof_end(e0, 2.0 * n * (40 * 256 + 6 * 65536 + 217 * 256), "trunk_r6_kernel", n * (16.0 + 8 * 1024))


CHAIN_SOFTPLUS, CHAIN_DSP, CHAIN_DBWD = 0, 1, 2


_CHAIN_MAX_ROWS = _max_rows(256)  # 32-bit byte offsets inside the kernel (ld = 256)


def chain(mode, P, x_in, wpack, n_layers, first_chunks, skip_layer=-1, side=None, bias=None, aux1=None, aux2=None,
          out=None, out2=None, wpack_x6=None, wpack_r6=None, skip_out=0, wpack_h3=None, c3=None):
    """hold_chain: n_layers consecutive 256-wide layers of one sweep with the activation resident in LDS.
    bias: per-layer [256] tensors; aux1 / aux2 / out / out2: per-layer [P,256] tensors (one common row stride) or None
    entries.  Batches beyond the kernel's 32-bit offset range are split by rows."""
    from . import gemm as _g
    assert wpack is None or wpack.numel() == _lib.lib().hold_chain_pack_floats(first_chunks, n_layers)
    if wpack_x6 is not None:  # split-precision sweep (hold_chain_x6): same descriptor, the weights as bf16 limbs
        assert wpack_x6.numel() * wpack_x6.element_size() == _lib.lib().hold_chain_x6_pack_bytes(first_chunks, n_layers)
    # register-resident sweeps (hold_chain_r6, csrc/rchain.hip): same descriptor.  DSP: wpack_r6 = field.pack_r6_stack of the
    # seven matrices; DBWD: wpack_r6 = the stream of hold_trunk_r6 (field.pack_r6), and the CALLER has stored the skip
    # layer's side columns in aux2[3][:, 217:256] (the kernel does not read `side`)
    # ... and in the two-limb fp16 arithmetic (hold_chain_h3, csrc/rchain_h3.hip): wpack_h3 = field's chain_bwd_h3 (DSP) / the
    # stream of hold_trunk_h3 (DBWD), c3 = 1 / s_w of the chain layers; wpack_r6 stays the operand of the conditional f32x6
    # fallback the entry point enqueues behind the kernel (the overflow guard of kernels.h3_guard)
    h3 = (wpack_h3 is not None and wpack_r6 is not None and mode in (CHAIN_DSP, CHAIN_DBWD) and
          (skip_out in (0, 217) or (skip_out == 172 and mode == CHAIN_DSP and aux2 is None)))
    if h3:
        want = _lib.lib().hold_chain_h3_pack_bytes() if mode == CHAIN_DSP else _lib.lib().hold_trunk_h3_pack_bytes()
        assert wpack_h3.numel() * wpack_h3.element_size() == want and wpack_h3.dtype == torch.float16
        assert c3 is not None and c3.numel() == n_layers and c3.dtype == torch.float32 and c3.is_cuda
    r6 = wpack_r6 is not None and mode in (CHAIN_DSP, CHAIN_DBWD)
    if r6:
        want = _lib.lib().hold_chain_r6_pack_bytes() if mode == CHAIN_DSP else _lib.lib().hold_trunk_r6_pack_bytes()
        assert wpack_r6.numel() * wpack_r6.element_size() == want
        assert skip_layer == 3
    for r0 in range(0, P, _CHAIN_MAX_ROWS):
        r1 = min(P, r0 + _CHAIN_MAX_ROWS)
        d = _lib.ChainDesc()
        d.P, d.mode, d.n_layers, d.first_chunks, d.skip_layer = r1 - r0, mode, n_layers, first_chunks, skip_layer
        d.skip_out = int(skip_out)  # 0 = 217; 172 = the background net (hold_chain_r6, DSP only)
        d.in_, d.ld_in = x_in[r0:r1].data_ptr(), _ld(x_in)
        if side is not None:
            d.side, d.ld_side = side[r0:r1].data_ptr(), _ld(side)
        assert r6 or wpack is not None
        d.wpack = wpack_h3.data_ptr() if h3 else wpack_r6.data_ptr() if r6 else (wpack if wpack_x6 is None else wpack_x6).data_ptr()
        ld = None
        for name, lst in (("bias", bias), ("aux1", aux1), ("aux2", aux2), ("out", out), ("out2", out2)):
            if lst is None:
                continue
            arr = getattr(d, name)
            for j in range(n_layers):
                t = lst[j]
                if t is None:
                    continue
                if name == "bias":
                    arr[j] = t.data_ptr()
                    continue
                arr[j] = t[r0:r1].data_ptr()
                assert ld is None or ld == _ld(t)
                ld = _ld(t)
        d.ld = ld if ld is not None else 256
        e0 = _g._prof_begin()
        if h3:
            call("hold_chain_h3", C.byref(d), ptr(c3), ptr(h3_guard(x_in.device)), ptr(wpack_r6))
        else:
            call("hold_chain_r6" if r6 else ("hold_chain" if wpack_x6 is None else "hold_chain_x6"), C.byref(d))
        # algorithmic HBM bytes: the chain input row + per layer 1 KiB per side input and per stored result
        n_mats = sum(sum(t is not None for t in lst) for lst in (aux1, aux2, out, out2) if lst is not None)
        _g._prof_end(e0, 2.0 * (r1 - r0) * 256 * (8 * first_chunks + 256 * (n_layers - 1)),
                     (("rchain_dbwd_h3_kernel" if mode == CHAIN_DBWD else "rchain_a2_h3_kernel" if aux2 is not None else
                       "rchain_bg_h3_kernel" if skip_out == 172 else "rchain_h3_kernel") if h3 else
                      ("rchain_dbwd_kernel" if mode == CHAIN_DBWD else "rchain_a2_kernel" if aux2 is not None else
                       "rchain_bg_kernel" if skip_out == 172 else "rchain_kernel")) if r6 else "chain_kernel",
                     (r1 - r0) * (32.0 * first_chunks + 1024.0 * n_mats))


def seed_dsp(h, w, N, P, t):
    call("hold_seed_dsp", ptr(h), _ld(h), ptr(w), N, P, ptr(t), _ld(t))


def colsum(X, N, P, out):
    call("hold_colsum", ptr(X), _ld(X), N, P, ptr(out))


# ------------------------------------------------------------------------------------------ sampler
def sampler_init(cam_loc, ray_dirs, R, near, n0, eps, t_rand, z, beta, far, err_flag):
    call("hold_sampler_init", ptr(cam_loc), ptr(ray_dirs), cam_loc.shape[0], float(R), float(near), n0, float(eps),
         ptr(t_rand), ptr(z), _ld(z), ptr(beta), ptr(far), ptr(err_flag))


def _hbm_prof(name, nbytes, fn):
    """time a streaming kernel for bench.py's roofline.kernels (ALGORITHMIC bytes / time against the HBM roof)"""
    from . import gemm as _g
    e0 = _g._prof_begin()
    fn()
    _g._prof_end(e0, float(nbytes), "hbm:" + name)


def sampler_beta(z, sdf, S, n_rays, sdf_new, slot, n_new, beta, beta0, eps, beta_iters, maxbeta_bits):
    # algorithmic HBM bytes: the sorted window in (z, sdf), the merged sdf window out, the new sdf in
    _hbm_prof("sampler_beta", 4.0 * n_rays * (2 * S + (S + n_new if n_new else 0) + 2),
              lambda: call("hold_sampler_beta", ptr(z), ptr(sdf), _ld(z), S, n_rays, ptr(sdf_new), ptr(slot), n_new, ptr(beta),
                           float(beta0), float(eps), beta_iters, ptr(maxbeta_bits)))


def sampler_sample(z, sdf, S, n_rays, beta, more, add_tiny, u, n_new, samples_out, slot_out):
    u_stride = 0 if u.dim() == 1 else u.stride(0)
    # window in (z, sdf), new samples out, and -- when the window grows -- the merged z window + slots out
    _hbm_prof("sampler_sample", 4.0 * n_rays * (2 * S + n_new + ((S + 2 * n_new) if more else 0)),
              lambda: call("hold_sampler_sample", ptr(z), ptr(sdf), _ld(z), S, n_rays, ptr(beta), int(more), float(add_tiny),
                           ptr(u), u_stride, n_new, ptr(samples_out), ptr(slot_out)))


def sampler_final(z_samples, ns, z, idx_extra, nx, far, near, n_rays, out):
    call("hold_sampler_final", ptr(z_samples), ns, ptr(z), _ld(z), ptr(idx_extra), nx, ptr(far), float(near), n_rays,
         ptr(out), _ld(out))


# ------------------------------------------------------------------------------------------ compositor
def make_composite_desc(S, n_rays, z, sdf, color, normal, class_ids, betas):
    d = CompositeDesc()
    d.n_nodes, d.S, d.n_rays = len(z), S, n_rays
    for i in range(len(z)):
        d.z[i], d.sdf[i] = z[i].data_ptr(), sdf[i].data_ptr()
        d.color[i], d.normal[i] = color[i].data_ptr(), normal[i].data_ptr()
        d.ldc[i], d.ldn[i] = _ld(color[i]), _ld(normal[i])
        d.class_id[i] = class_ids[i]
        d.beta[i] = float(betas[i])
    return d


def composite_fwd(d, out_node, out_comp, out_sem, out_w=None, out_zmerge=None, out_w_node=None):
    for i, o in enumerate(out_node):
        d.out_node[i] = o.data_ptr()
        if out_w_node is not None:
            d.out_w_node[i] = out_w_node[i].data_ptr()
    d.out_comp, d.out_sem = out_comp.data_ptr(), out_sem.data_ptr()
    d.out_w = out_w.data_ptr() if out_w is not None else None
    d.out_zmerge = out_zmerge.data_ptr() if out_zmerge is not None else None
    n, S, N = d.n_nodes, d.S, d.n_rays
    # per node and sample: z, sdf, colour 3, normal 3 in; per ray: 12 floats per node + composite + 4 semantics out,
    # the merged weights (n S - 2 n + 1) and the per-node weights when requested
    nb = 4.0 * N * (n * S * 8 + 12 * (n + 1) + 4 + ((n * S - 2 * n + 1) if out_w is not None else 0)
                    + (n * S if out_w_node is not None else 0))
    _hbm_prof("composite_fwd", nb, lambda: call("hold_composite_fwd", C.byref(d)))


def composite_bwd(d, d_node, d_comp, d_sem, d_sdf, d_color, d_normal, d_beta):
    for i in range(d.n_nodes):
        d.d_node[i], d.d_sdf[i] = d_node[i].data_ptr(), d_sdf[i].data_ptr()
        d.d_color[i], d.d_normal[i] = d_color[i].data_ptr(), d_normal[i].data_ptr()
    d.d_comp, d.d_sem, d.d_beta = d_comp.data_ptr(), d_sem.data_ptr(), d_beta.data_ptr()
    n, S, N = d.n_nodes, d.S, d.n_rays
    nb = 4.0 * N * (n * S * 8 + 12 * (n + 1) + 4 + n * S * 7)  # forward inputs + cotangents in, d sdf / colour / normal out
    _hbm_prof("composite_bwd", nb, lambda: call("hold_composite_bwd", C.byref(d)))


def bg_composite_fwd(z_desc, sdf, rgb, S, n_rays, out_rgb, w_out=None):
    call("hold_bg_composite_fwd", ptr(z_desc), ptr(sdf), ptr(rgb), _ld(rgb), S, n_rays, ptr(out_rgb), ptr(w_out))


def bg_composite_bwd(z_desc, sdf, rgb, S, n_rays, d_out, d_sdf, d_rgb):
    call("hold_bg_composite_bwd", ptr(z_desc), ptr(sdf), ptr(rgb), _ld(rgb), S, n_rays, ptr(d_out), ptr(d_sdf),
         ptr(d_rgb))
