"""The weight-pack layouts of include/hold_hip.h built matrix by matrix (the obvious, slow way): the layout oracle for
hold_amd.field.pack_weights, which builds the same tensors from stacked matrices in a few device ops."""
import math

import torch

from hold_amd import config
from hold_amd.field import FieldSpec, pad4, rin_perm


def split_limbs(w, n=3):
    """exact bf16 limb decomposition w = sum_t limb_t (limb_t = bf16 rounding of the residual), as fp32 values"""
    out, r = [], w.float()
    for _ in range(n):
        l = r.to(torch.bfloat16)
        out.append(l)
        r = r - l.float()
    return out


def pack_x6(W8, first_k=48):
    """limb pack of hold_fused_sdf_x6 / hold_chain_x6 (include/hold_hip.h) from up to 8 matrices W8[l] ([<=256, K_l]; the
    first one zero-padded to first_k = 48 columns for the 40-wide embedding input, or 256): bf16 tensor,
    [K_l/16 steps][3 limbs][8 n-tiles][2 h][32 i][8 e] per layer."""
    parts = []
    for l, wl in enumerate(W8):
        K = first_k if l == 0 else 256
        m = torch.zeros(256, K, device=wl.device)
        m[:wl.shape[0], :wl.shape[1]] = wl
        limbs = torch.stack(split_limbs(m))  # [3, 256, K] bf16
        parts.append(limbs.reshape(3, 8, 32, K // 16, 2, 8).permute(3, 0, 1, 4, 2, 5).reshape(-1))
    return torch.cat(parts).contiguous()


def pack_weights(spec: FieldSpec, iw, ib, rw, rb, need_bwd: bool):
    """iw/ib: 9 effective ImplicitNet weights/biases ([out,in] as nn.Linear); rw/rb: 5 RenderingNet ones (or None).
    Returns the re-laid-out (and, for sweeps that contract over the output index, transposed) copies the
    kernels read.  Tiny (<= 256x304) device ops once per step."""
    dev = iw[0].device
    pk = {}
    W = []
    w0 = torch.zeros(256, spec.K0, device=dev)
    w0[:, :spec.E] = iw[0][:, :spec.E]  # the 45 MANO pose-cond columns multiply zeros (shape_net.py:104-106)
    W.append(w0)
    W += [iw[1].contiguous(), iw[2].contiguous(), iw[3].contiguous()]
    W.append((iw[4] / math.sqrt(2)).contiguous())  # cat([x, input]) / sqrt(2) folded into the weight
    W += [iw[5].contiguous(), iw[6].contiguous(), iw[7].contiguous()]
    w8 = torch.cat([iw[8][1:], iw[8][:1]], 0).contiguous()  # rows: feat(256) then sdf
    W.append(w8)
    pk["W"] = W
    pk["b"] = [b.contiguous() for b in ib[:8]] + [torch.cat([ib[8][1:], ib[8][:1]]).contiguous()]
    pk["iw0_cols"] = iw[0].shape[1]
    pk["w8_sdf"] = iw[8][0].contiguous()
    pk["b8_sdf"] = ib[8][:1]
    pk["W8_feat"], pk["b8_feat"] = w8[:256], pk["b"][8][:256]  # lin8 without its sdf row (a 257th column costs a whole tile)
    # transposes [K_l][pad4(N_l)] for the sweeps that contract over the output index
    WT = []
    for l in range(9):
        n, k = W[l].shape
        wt = torch.zeros(k, pad4(n), device=dev)
        wt[:, :n] = W[l].t()
        WT.append(wt)
    pk["WT"] = WT
    pk["WT8_feat"] = W[8][:256].t().contiguous()  # [k = 256 trunk units][n = 256 feature rows]: lin8's input gradient
    # fragment-ordered pack for the fused SDF-only kernel (hold_fused_sdf)
    parts = []
    for l in range(8):
        wl = W[l]
        if wl.shape[0] < 256:
            wl = torch.cat([wl, torch.zeros(256 - wl.shape[0], wl.shape[1], device=dev)], 0)
        ch = wl.shape[1] // 8
        parts.append(wl.reshape(8, 32, ch, 2, 4).permute(2, 0, 3, 1, 4).reshape(-1))
    bias8 = torch.zeros(8, 256, device=dev)
    for l in range(8):
        bias8[l, :pk["b"][l].shape[0]] = pk["b"][l]
    pk["fused"] = (torch.cat(parts).contiguous(), bias8.contiguous())
    if config.x6():
        pk["fused_x6"] = pack_x6(W[:8])
    # descending sweeps (hold_chain DSP): layer j contracts over the outputs of trunk layer l = 7 - j, M_j = W_l^T
    parts = []
    for l in range(7, 0, -1):
        m = torch.zeros(256, 256, device=dev)
        m[:W[l].shape[1], :W[l].shape[0]] = W[l].t()
        parts.append(m.reshape(8, 32, 32, 2, 4).permute(2, 0, 3, 1, 4).reshape(-1))
    pk["chain_bwd"] = torch.cat(parts).contiguous()
    if config.x6():  # limb packs of the same matrices for hold_chain_x6 (the forward-type sweeps share the sampler trunk's)
        pk["chain_fwd_x6"] = pk["fused_x6"]
        mats = []
        for l in range(7, 0, -1):
            m = torch.zeros(256, 256, device=dev)
            m[:W[l].shape[1], :W[l].shape[0]] = W[l].t()
            mats.append(m)
        pk["chain_bwd_x6"] = pack_x6(mats, first_k=256)
    if rw is None:  # implicit net only (ImplicitNet.forward / gradient)
        return pk
    r0 = torch.zeros(256, spec.Kr, device=dev)
    r0[:, :spec.rin_dim] = rw[0][:, rin_perm(spec.rin_dim, dev)]
    R = [r0, rw[1].contiguous(), rw[2].contiguous(), rw[3].contiguous(), rw[4].contiguous()]
    pk["R"] = R
    pk["rb"] = [b.contiguous() for b in rb]
    if need_bwd:
        RT = []
        for l in range(5):
            n, k = R[l].shape
            rt = torch.zeros(k, pad4(n), device=dev)
            rt[:, :n] = R[l].t()
            RT.append(rt)
        pk["RT"] = RT
    return pk
