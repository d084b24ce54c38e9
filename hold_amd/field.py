"""Canonical SDF / colour field of one foreground node on the HIP kernels.

Host-side orchestration of ImplicitNet.forward (code/src/networks/shape_net.py:84-130), the
canonical normal of extract_features (code/src/engine/volsdf_utils.py:51-105), RenderingNet.forward in
mode 'pose' (code/src/networks/texture_net.py:46-101) and their hand-derived backward (including the
second-order terms torch would get from ``create_graph=True``), as sequences of libholdhip launches.
Nothing here computes on the CPU or with torch math beyond tiny per-layer weight re-layouts.

Buffers (fp32, ray-major points p = ray*S + s):
  x      [P,4]   query points in deformed space          xc   [P,4]  canonical points
  in0    [P,K0]  embedding (+cond)                        h[l] [P,256] softplus outputs (h[3] = [h3|embed])
  rin    [P,Kr]  render-net input [feat256,xc,n,pose8(,time32)] -- the reference's column order with the feature block
                 moved to the front (16-byte aligned for lin8's output / its cotangent); pack_weights permutes lin0
  t[l]   [P,256] reverse-sweep d sdf/d a_l               g    [P,4]   d sdf/d xc
  ge     [P,E]   d sdf/d embed: with the layer chains the view t[3][:, 217:] (row stride 256; the descending sweep leaves
                 the skip part there, layer 0's part is accumulated in place, hold_embed_bwd2 overwrites it in place with the
                 side columns of the ascending sweep); a [P,K0] buffer on the layer-by-layer route
"""
from __future__ import annotations

import math
import os

import torch

from . import config
from . import gemm as G
from . import kernels as K

FEAT = 256
FUSED_SDF = os.environ.get("HOLD_FUSED_SDF", "1") != "0"  # sampler queries through the fused LDS-resident kernel
# training-path sweeps as LDS-resident layer chains (hold_chain) instead of one hold_gemm_nt per layer
USE_CHAIN = os.environ.get("HOLD_CHAIN", "1") != "0"
# register-resident trunk (csrc/rmlp.hip) for the sampler queries and the training forward trunk (f32x6 arithmetic only)
USE_R6 = os.environ.get("HOLD_R6", "1") != "0"
USE_H3_TRUNK = os.environ.get("HOLD_H3_TRUNK", "1") != "0"  # mode f16x3: the training forward trunk too (A/B switch)
COMPACT = os.environ.get("HOLD_COMPACT", "1") != "0"  # exact sample compaction behind the sdf (csrc/compact.hip); A/B switch
COMPACT_ALIGN = 128       # compacted row counts are padded to this (the kernels' fast paths: P % 16 == 0, 128-point blocks)
COMPACT_MAX_LIVE = 0.85   # compaction moves 9 KiB per live sample: above this live fraction it costs more than it saves
COMPACT_BACKOFF = 16      # calls between two compaction attempts of a field while attempts keep failing (each one is a host read)
COMPACT_MIN_POINTS = 1 << 18  # below this many samples per call a step is bound by the host's launch rate: the dozen
#                               host operations of a compaction then cost more than the dropped rows save (measured: the reference's
#                               10 x 128-ray batch at beta = 0.005, 125 k samples per node: 35.5-37.7 ms with, 34.0-34.8 ms without)
USE_R6_BWD = os.environ.get("HOLD_R6_BWD", "1") != "0"  # ... and for the descending sweeps (csrc/rchain.hip)
USE_H3_BWD = os.environ.get("HOLD_H3_BWD", "1") != "0"  # mode f16x3: the three backward sweeps too (csrc/rchain_h3.hip; A/B switch)
# the 256 x 256 weight gradients of a backward as one grouped launch (gemm.WgradGroup / hold_wgrad_group_x6)
USE_WGRAD_GROUP = os.environ.get("HOLD_WGRAD_GROUP", "1") != "0"
USE_R6_GEMM = os.environ.get("HOLD_R6_GEMM", "1") != "0"  # ... and for the rendering net's layers / lin8 (csrc/rgemm.hip)
USE_H3_GEMM = os.environ.get("HOLD_H3_GEMM", "1") != "0"  # mode f16x3: those layers in two fp16 limbs (csrc/rgemm_h3.hip; A/B switch)
USE_RELU_BITS = os.environ.get("HOLD_RELU_BITS", "1") != "0"  # A/B: the backward's ReLU masks from bits (hold_gemm_h3_bits) or from r_l
H3_ROW_FLOOR = 64.0  # assumed bound of activation columns no producer reported a maximum for (= 2^6, the trunk's activation scale)
RIN_FEAT, RIN_X, RIN_N, RIN_POSE, RIN_TIME = 0, 256, 259, 262, 270


_RIN_PERM = {}


def rin_perm(rin_dim, device=None):
    """our rin column j holds the reference's rendering-net input column rin_perm[j]
    (reference order: xc 0:3, normal 3:6, pose 6:14, feature 14:270, time 270:302; texture_net.py:60-83)"""
    key = (rin_dim, str(device))
    if key not in _RIN_PERM:
        idx = list(range(14, 270)) + list(range(0, 14)) + list(range(270, rin_dim))
        _RIN_PERM[key] = torch.tensor(idx, dtype=torch.long, device=device)
    return _RIN_PERM[key]


def pad4(n):
    return (n + 3) // 4 * 4


class Pool:
    """named scratch buffers, grown on demand, zero-initialised on (re)allocation."""

    def __init__(self, device):
        self.device = device
        self.bufs = {}
        # rows a NEW buffer gets at least.  NodeField.forward sets it to the call's uncompacted sample count: the compacted row
        # count changes from call to call, and a pool that grew with it re-allocated every [rows, 256] buffer of the tail and
        # the backward whenever a call had more live samples than any before -- ~19 hipMallocs = 30 GB and up to a second
        # inside a step, the outgrown blocks left behind in torch's cache (reserved memory crept to 285 GiB of the 288 GB;
        # round 6, GPU call 23).  Sized by the uncompacted count once, like the uncompacted path, they never grow.
        self.min_rows = 0

    def get(self, name, rows, cols, dtype=torch.float32):
        b = self.bufs.get(name)
        if b is None or b.shape[0] < rows or b.shape[1] != cols or b.dtype != dtype:
            b = torch.zeros(max(rows, self.min_rows), cols, dtype=dtype, device=self.device)
            self.bufs[name] = b
        return b[:rows]

    def nbytes(self):
        return sum(b.numel() * b.element_size() for b in self.bufs.values())


_SHARED_POOLS = {}


def shared_pool(device):
    """ONE backward-scratch pool per device, shared by every NodeField: the backward of a node runs to completion inside
    one autograd call and only then the next node's starts, so the sweeps' intermediates (a2_l, vbar_l, r_l, the rendering
    net's cotangents: ~24 KB per point) need not exist once per node -- with three foreground nodes that is what lets a
    16 384-ray chunk of the two-hand scene fit (the forward activations a backward reads stay in the node's own pool)."""
    k = str(device)
    if k not in _SHARED_POOLS:
        _SHARED_POOLS[k] = Pool(device)
    return _SHARED_POOLS[k]


class FieldSpec:
    """static shape description of a node's networks."""

    def __init__(self, kind):
        assert kind in ("hand", "object")
        self.kind = kind
        self.L = 6
        self.E = 3 + 3 * 2 * self.L  # 39
        self.K0 = pad4(self.E)  # 40
        self.skip_out = 256 - self.E  # 217
        self.skip_pad = pad4(self.skip_out)  # 220
        self.time = 32 if kind == "object" else 0
        self.rin_dim = 3 + 3 + 8 + FEAT + self.time  # 270 / 302
        self.Kr = pad4(self.rin_dim)
        self.n_bones = 1 if kind == "object" else 16


def zeros_like_many(*lists):
    """zero tensors shaped like the given ones, carved out of ONE zero-filled buffer (256-byte aligned pieces): one fill
    launch instead of one per tensor.  Returns one list per input list."""
    ts = [t for lst in lists for t in lst]
    sizes = [(t.numel() + 63) // 64 * 64 for t in ts]
    flat = torch.zeros(sum(sizes), device=ts[0].device)
    out, off = [], 0
    for t, n in zip(ts, sizes):
        out.append(flat[off:off + t.numel()].view(t.shape))
        off += n
    res, i = [], 0
    for lst in lists:
        res.append(out[i:i + len(lst)])
        i += len(lst)
    return res


def split_limbs(w, n=3):
    """exact bf16 limb decomposition w = sum_t limb_t (limb_t = bf16 rounding of the residual), as fp32 values"""
    out, r = [], w.float()
    for _ in range(n):
        l = r.to(torch.bfloat16)
        out.append(l)
        r = r - l.float()
    return out


# layouts of the limb packs, applied to ALREADY split limbs l3 = [3 limbs, ...] (any dtype: pack_plan() runs them on index
# tensors to derive the gather that builds a pack in one launch)
def _lay_x6(l3, K):
    return l3.reshape(3, 8, 32, K // 16, 2, 8).permute(3, 0, 1, 4, 2, 5).reshape(-1)


def _lay_x6_stack(l3):
    L = l3.shape[1]
    return l3.reshape(3, L, 8, 32, 16, 2, 8).permute(1, 4, 0, 2, 5, 3, 6).reshape(-1)


def _lay_r6_0(l3):
    T, K = l3.shape[0], l3.shape[2]  # limbs: 3 (bf16, hold_trunk_r6: K = 48) or 2 (fp16, hold_trunk_h3: K = 64)
    return l3.reshape(T, 8, 32, K // 16, 2, 8).permute(3, 1, 0, 4, 2, 5).reshape(-1)


def _lay_r6_stack(l3):
    T, L = l3.shape[0], l3.shape[1]
    g = l3[:, :, :, r6_kmap(l3.device)]  # [T limbs, L, 256 out, 16 j, 2 h, 8 e]
    return g.reshape(T, L, 8, 32, 16, 2, 8).permute(1, 4, 2, 0, 5, 3, 6).reshape(-1)


def pack_x6(W8, first_k=48):
    """limb pack of hold_fused_sdf_x6 / hold_chain_x6 (include/hold_hip.h) from up to 8 matrices W8[l] ([<=256, K_l]; the
    first one zero-padded to first_k = 48 columns for the 40-wide embedding input, or 256): bf16 tensor,
    [K_l/16 steps][3 limbs][8 n-tiles][2 h][32 i][8 e] per layer."""
    parts = []
    for l, wl in enumerate(W8):
        K = first_k if l == 0 else 256
        m = torch.zeros(256, K, device=wl.device)
        m[:wl.shape[0], :wl.shape[1]] = wl
        parts.append(_lay_x6(torch.stack(split_limbs(m)), K))  # limbs [3, 256, K] bf16
    return torch.cat(parts).contiguous()


def pack_x6_stack(S):
    """pack_x6 of L zero-padded [256, 256] matrices given as one [L, 256, 256] tensor (same layout, a dozen device ops
    for all of them instead of a dozen per matrix)"""
    return _lay_x6_stack(torch.stack(split_limbs(S)))  # limbs [3, L, 256, 256] bf16: (t, L, 32 nt + i, 16 step + 8 h + e)


_KMAP = {}


def r6_kmap(device=None):
    """virtual k order of hold_trunk_r6 for the 256-wide layers: kmap[j, h, e] = input feature that element e of lane half h
    holds in k step j = the register order of the previous layer's v_mfma_f32_32x32x16_bf16 outputs (csrc/rmlp.hip)"""
    key = str(device)
    if key not in _KMAP:
        j = torch.arange(16, device=device).view(16, 1, 1)
        h = torch.arange(2, device=device).view(1, 2, 1)
        e = torch.arange(8, device=device).view(1, 1, 8)
        _KMAP[key] = 32 * (j // 2) + 16 * (j % 2) + 8 * (e // 4) + 4 * h + e % 4
    return _KMAP[key]


def pack_r6(w0, S):
    """limb pack of hold_fused_sdf_r6 / hold_trunk_r6 (include/hold_hip.h): w0 [256, <=48] (layer 0, zero-padded to K = 48),
    S [7, 256, 256] (layers 1..7, rows zero-padded, lin4 pre-scaled) -> bf16 [115 steps][8 nt][3 t][2 h][32 i][8 e]"""
    dev = S.device
    m0 = torch.zeros(256, 48, device=dev)
    m0[:, :w0.shape[1]] = w0
    p0 = _lay_r6_0(torch.stack(split_limbs(m0)))  # limbs [3, 256, 48] = (t, 32 nt + i, 16 j + 8 h + e)
    return torch.cat([p0, _lay_r6_stack(torch.stack(split_limbs(S)))]).contiguous()


H3_K0 = 64  # layer 0's K in the f16x3 stream (39 embedding columns, zero-padded to four 16-wide k steps)
H3_ACT_SCALE = 64.0  # SA of csrc/rmlp_h3.hip (checked against hold_trunk_h3_act_scale() by kernels.fused_sdf_h3)


def h3_scales(w0, S):
    """per-matrix power-of-two weight scales of the f16x3 trunk: s_w[l] = 2^k with max |W_l| s_w in [2^13, 2^14) (frexp: exact,
    no host read) -> [8] device tensor"""
    amax = torch.cat([w0.abs().amax().view(1), S.abs().amax(dim=(1, 2))])
    _, ex = torch.frexp(amax)  # amax = m 2^ex, m in [0.5, 1)
    return torch.ldexp(torch.ones_like(amax), 14 - ex)


def split_limbs_h(ws):
    """two-limb fp16 decomposition of ALREADY SCALED values: hi = RN_f16(ws), lo = RN_f16(ws - hi)"""
    hi = ws.to(torch.float16)
    return [hi, (ws - hi.float()).to(torch.float16)]


def pack_h3(w0, S):
    """limb pack of hold_fused_sdf_h3 / hold_trunk_h3 (include/hold_hip.h): the k order and tiling of pack_r6 with TWO fp16
    limbs of the scaled weights and layer 0 padded to K = 64 (four k steps: the ring slot of a k step is then a compile-time
    constant in every layer) -> (fp16 [116 steps][8 nt][2 t][2 h][32 i][8 e], s_w [8])"""
    dev = S.device
    sw = h3_scales(w0, S)
    m0 = torch.zeros(256, H3_K0, device=dev)
    m0[:, :w0.shape[1]] = w0
    p0 = _lay_r6_0(torch.stack(split_limbs_h(m0 * sw[0])))
    ps = _lay_r6_stack(torch.stack(split_limbs_h(S * sw[1:].view(7, 1, 1))))
    return torch.cat([p0, ps]).contiguous(), sw


def pack_h3_stack(S):
    """hold_chain_h3 (DSP) stream of L [256, 256] matrices: every layer in the virtual k order (r6_kmap), two fp16 limbs of the
    matrix scaled by its own s_w = 2^k (max |M| s_w in [2^13, 2^14)) -> (fp16 [L x 16 steps][8 nt][2 t][2 h][32 i][8 e], s_w [L])"""
    _, ex = torch.frexp(S.abs().amax(dim=(1, 2)))
    sw = torch.ldexp(torch.ones_like(ex, dtype=torch.float32), 14 - ex)
    return _lay_r6_stack(torch.stack(split_limbs_h(S * sw.view(-1, 1, 1)))).contiguous(), sw


def pack_r6_stack(S):
    """hold_chain_r6 (DSP) stream of L [256, 256] matrices: every layer in the virtual k order (r6_kmap)"""
    return _lay_r6_stack(torch.stack(split_limbs(S))).contiguous()


def pack_gemm_r6(W):
    """limb pack of hold_gemm_r6 (include/hold_hip.h) for one matrix W [N <= 256, K in 256..320]: K padded to KS = 4 ceil(K/64)
    k steps, the k index in the register order of an accumulator tile (r6_kmap continued to KS steps)"""
    N, K = W.shape
    KS = (K + 63) // 64 * 4
    m = torch.zeros(256, 16 * KS, device=W.device)
    m[:N, :K] = W
    return _lay_gemm_r6(torch.stack(split_limbs(m)), KS).contiguous()


def _gemm_kmap(KS, device):
    j = torch.arange(KS, device=device).view(KS, 1, 1)
    h = torch.arange(2, device=device).view(1, 2, 1)
    e = torch.arange(8, device=device).view(1, 1, 8)
    return 16 * j + 8 * (e // 4) + 4 * h + e % 4


def _lay_gemm_r6(l3, KS):
    T = l3.shape[0]  # limbs: 3 (bf16, hold_gemm_r6) or 2 (fp16, hold_gemm_h3)
    g = l3[:, :, _gemm_kmap(KS, l3.device)]  # [T t, 256 out, KS j, 2 h, 8 e]
    return g.reshape(T, 8, 32, KS, 2, 8).permute(3, 1, 0, 4, 2, 5).reshape(-1)


def pack_gemm_h3(W):
    """limb pack of hold_gemm_h3 (include/hold_hip.h) for one matrix W [N <= 256, K in 256..320]: pack_gemm_r6's rows and k order,
    two fp16 limbs of s_w W with s_w = 2^k, max |W| s_w in [2^13, 2^14) -> (stream fp16, c3 = 1 / s_w as a one-element tensor)"""
    N, K = W.shape
    KS = (K + 63) // 64 * 4
    m = torch.zeros(256, 16 * KS, device=W.device)
    m[:N, :K] = W
    _, ex = torch.frexp(m.abs().amax())
    sw = torch.ldexp(torch.ones((), device=W.device), 14 - ex)
    return _lay_gemm_r6(torch.stack(split_limbs_h(m * sw)), KS).contiguous(), (1.0 / sw).reshape(1)


_PLANS = {}
_RPLANS = {}


def render_plan(Kr, need_bwd, device):
    """gather indices of the hold_gemm_r6 packs of the rendering net from the flat source
    [R0 zero-padded to [256, 320] | R1 | R2 | R3 (| R1^T | R2^T | R3^T | (R0[:, :256])^T)] (see pack_plan)"""
    key = (Kr, bool(need_bwd), str(device))
    if key not in _RPLANS:
        n0 = 256 * 320
        nm = 7 if need_bwd else 3
        N = n0 + nm * 65536
        l3 = lambda I: torch.stack([I + t * N for t in range(3)])
        I0 = torch.arange(n0, device=device).view(256, 320)
        parts = [_lay_gemm_r6(l3(I0), 20)]
        for q in range(nm):
            parts.append(_lay_gemm_r6(l3(n0 + q * 65536 + torch.arange(65536, device=device).view(256, 256)), 16))
        # the same packs in two fp16 limbs (hold_gemm_h3): the gather over [2, N] limbs of the SCALED source, and the matrix a
        # source element belongs to (per-matrix power-of-two scales, computed on the device at pack time)
        l2 = lambda I: torch.stack([I + t * N for t in range(2)])
        parts_h = [_lay_gemm_r6(l2(I0), 20)]
        for q in range(nm):
            parts_h.append(_lay_gemm_r6(l2(n0 + q * 65536 + torch.arange(65536, device=device).view(256, 256)), 16))
        seg = torch.cat([torch.zeros(n0, dtype=torch.long, device=device),
                         1 + torch.arange(nm, device=device).repeat_interleave(65536)])
        _RPLANS[key] = dict(N=N, idx=torch.cat(parts).to(torch.int32).contiguous(),
                            sizes=[p.numel() for p in parts], idx_h3=torch.cat(parts_h).to(torch.int32).contiguous(),
                            sizes_h3=[p.numel() for p in parts_h], seg=seg, nm=1 + nm)
    return _RPLANS[key]



def pack_plan(K0, device):
    """int32 gather indices that build every fragment / limb pack of the trunk from ONE flat source
    src = [w0 zero-padded to [256, 48] | S [7, 256, 256]] and its three bf16 limbs [3, N]: a pack is a fixed permutation
    of (limb, element), so it is derived once by running the pack's layout on an index tensor and is then one
    index_select per pack and step -- a training step at the reference's 1 280-ray batch re-packs once per step and
    spent ~600 small launches doing so layer by layer."""
    key = (K0, str(device))
    if key not in _PLANS:
        n0, ns = 256 * 48, 7 * 65536
        N = n0 + ns
        I0 = torch.arange(n0, device=device).view(256, 48)
        IS = n0 + torch.arange(ns, device=device).view(7, 256, 256)
        I8 = n0 + ns + torch.arange(65536, device=device).view(256, 256)  # lin8's 256 feature rows (appended to the source)
        N += 65536
        ISTf = IS.transpose(1, 2).flip(0)  # descending sweeps: layer j = W_{7-j}^T
        l3 = lambda I: torch.stack([I + t * N for t in range(3)])
        l2 = lambda I: torch.stack([I + t * N for t in range(2)])
        i32 = lambda t: t.to(torch.int32).contiguous()
        frag0 = I0[:, :K0].reshape(8, 32, K0 // 8, 2, 4).permute(2, 0, 3, 1, 4).reshape(-1)
        _PLANS[key] = dict(
            N=N,
            fused=i32(torch.cat([frag0, frag_pack_stack(IS)])),
            chain_bwd=i32(frag_pack_stack(ISTf)),
            fused_x6=i32(torch.cat([_lay_x6(l3(I0), 48), _lay_x6_stack(l3(IS))])),
            trunk_r6=i32(torch.cat([_lay_r6_0(l3(I0)), _lay_r6_stack(l3(IS))])),
            h3=_h3_plan(device),
            chain_bwd_x6=i32(_lay_x6_stack(l3(ISTf))),
            chain_bwd_r6=i32(_lay_r6_stack(l3(ISTf))),
            w8_feat_r6=i32(_lay_gemm_r6(l3(I8), 16)))
    return _PLANS[key]


def _h3_plan(device):
    """gather of the f16x3 trunk stream from ITS flat source [w0 zero-padded to [256, 64] | S [7, 256, 256]] (two limbs)"""
    n0, ns = 256 * H3_K0, 7 * 65536
    N = n0 + ns
    I0 = torch.arange(n0, device=device).view(256, H3_K0)
    IS = n0 + torch.arange(ns, device=device).view(7, 256, 256)
    l2 = lambda I: torch.stack([I + t * N for t in range(2)])
    seg = torch.cat([torch.zeros(n0, dtype=torch.long, device=device), 1 + torch.arange(7, device=device).repeat_interleave(65536)])
    ISTf = IS.transpose(1, 2).flip(0)  # descending sweeps (hold_chain_h3, DSP): chain layer j = W_{7-j}^T, scaled by s_w[7 - j]
    I8 = torch.arange(65536, device=device).view(256, 256)  # lin8's feature rows: a source of their own ([2, 65536] limbs)
    l2_8 = torch.stack([I8, I8 + 65536])
    return dict(idx=torch.cat([_lay_r6_0(l2(I0)), _lay_r6_stack(l2(IS))]).to(torch.int32).contiguous(), seg=seg,
                idx_bwd=_lay_r6_stack(l2(ISTf)).to(torch.int32).contiguous(),
                idx_w8=_lay_gemm_r6(l2_8, 16).to(torch.int32).contiguous())


def frag_pack_stack(S):
    """fp32 MFMA-fragment order of hold_fused_sdf / hold_chain for L [256, 256] matrices:
    per matrix [K/8 chunks][8 n-tiles][2 h][32 i][4]"""
    L = S.shape[0]
    return S.reshape(L, 8, 32, 32, 2, 4).permute(0, 3, 1, 4, 2, 5).reshape(-1)


def pack_weights(spec: FieldSpec, iw, ib, rw, rb, need_bwd: bool, trunk=None):
    """iw/ib: 9 effective ImplicitNet weights/biases ([out,in] as nn.Linear); rw/rb: 5 RenderingNet ones (or None).
    Returns the re-laid-out (and, for sweeps that contract over the output index, transposed) copies the
    kernels read.  The seven 256-wide trunk layers are handled as ONE [7, 256, 256] tensor (the per-layer entries of
    pk["W"] / pk["WT"] are views of it): ~70 small device ops per pack instead of ~290 -- what a 1 280-ray training
    step, which packs once per step, spends its host time on."""
    dev = iw[0].device
    if trunk is not None:  # the ImplicitNet part was packed already (same weights): add the RenderingNet's
        return _pack_render(dict(trunk), spec, rw, rb, need_bwd, dev)
    pk = {}
    pad = torch.nn.functional.pad
    w0 = pad(iw[0][:, :spec.E], (0, spec.K0 - spec.E))  # the 45 MANO pose-cond columns multiply zeros (shape_net.py:104-106)
    # trunk layers 1..7, rows zero-padded to 256 (layer 3 has 217); cat([x, input]) / sqrt(2) folded into lin4's weight
    S = torch.stack([(lambda m: m if m.shape[0] == 256 else pad(m, (0, 0, 0, 256 - m.shape[0])))(
        iw[l] / math.sqrt(2) if l == 4 else iw[l]) for l in range(1, 8)])
    w8 = torch.cat([iw[8][1:], iw[8][:1]], 0).contiguous()  # rows: feat(256) then sdf
    W = [w0] + [S[l - 1][:iw[l].shape[0]] for l in range(1, 8)] + [w8]
    pk["W"] = W
    pk["b"] = [b.contiguous() for b in ib[:8]] + [torch.cat([ib[8][1:], ib[8][:1]]).contiguous()]
    pk["iw0_cols"] = iw[0].shape[1]
    pk["w8_sdf"] = iw[8][0].contiguous()
    pk["b8_sdf"] = ib[8][:1].contiguous()  # stays on the device (hold_fused_sdf_r6 reads it there); the scalar-argument
    # entry points (hold_fused_sdf / _x6) take float(pk["b8_sdf"]) at their call sites -- a host read drains the stream
    pk["W8_feat"], pk["b8_feat"] = w8[:256], pk["b"][8][:256]  # lin8 without its sdf row (a 257th column costs a whole tile)
    # transposes [K_l][pad4(N_l)] for the sweeps that contract over the output index
    ST = S.transpose(1, 2).contiguous()  # [7][k][n], columns n >= N_l zero
    wt8 = torch.zeros(256, pad4(257), device=dev)
    wt8[:, :257] = w8.t()
    pk["WT"] = [w0.t().contiguous()] + [ST[l - 1][:, :pad4(iw[l].shape[0])] for l in range(1, 8)] + [wt8]
    pk["WT8_feat"] = w8[:256].t().contiguous()  # [k = 256 trunk units][n = 256 feature rows]: lin8's input gradient
    # fragment-ordered pack for the fused SDF-only kernel (hold_fused_sdf)
    bias8 = torch.stack([b if b.shape[0] == 256 else torch.nn.functional.pad(b, (0, 256 - b.shape[0])) for b in pk["b"][:8]])
    # every fragment / limb pack = one gather from the flat source (pack_plan); the descending sweeps' matrices
    # (hold_chain DSP: layer j contracts over the outputs of trunk layer l = 7 - j, M_j = W_l^T) come from the same source
    plan = pack_plan(spec.K0, dev)
    src = torch.cat([torch.nn.functional.pad(w0, (0, 48 - spec.K0)).reshape(-1), S.reshape(-1), w8[:256].reshape(-1)])
    pk["fused"] = (src.index_select(0, plan["fused"]), bias8)
    pk["chain_bwd"] = src.index_select(0, plan["chain_bwd"])
    if config.x6():  # limb packs (the forward-type sweeps of hold_chain_x6 share the sampler trunk's)
        limbs = torch.stack(split_limbs(src)).reshape(-1)  # [3 N] bf16
        pk["fused_x6"] = limbs.index_select(0, plan["fused_x6"])
        pk["trunk_r6"] = limbs.index_select(0, plan["trunk_r6"])
        pk["chain_fwd_x6"] = pk["fused_x6"]
        pk["chain_bwd_x6"] = limbs.index_select(0, plan["chain_bwd_x6"])
        pk["chain_bwd_r6"] = limbs.index_select(0, plan["chain_bwd_r6"])
        pk["w8_feat_r6"] = limbs.index_select(0, plan["w8_feat_r6"])  # hold_gemm_r6 pack of lin8's feature rows
    if config.h3():  # two-limb fp16 stream of the forward trunk (csrc/rmlp_h3.hip) + its scales, all on the device
        sw = h3_scales(w0, S)
        src3 = torch.cat([torch.nn.functional.pad(w0, (0, H3_K0 - spec.K0)).reshape(-1), S.reshape(-1)])
        limbs_h = torch.stack(split_limbs_h(src3 * sw[plan["h3"]["seg"]])).reshape(-1)
        pk["trunk_h3"] = limbs_h.index_select(0, plan["h3"]["idx"])
        pk["bias8_h3"] = (bias8 * (sw * H3_ACT_SCALE).view(8, 1)).contiguous()
        pk["c3_h3"] = (1.0 / sw).contiguous()
        # the backward sweeps in the same arithmetic (hold_chain_h3): the descending ones read the transposed matrices (chain
        # layer j = W_{7-j}^T, the same limbs gathered in another order), the ascending one the forward stream itself
        pk["chain_bwd_h3"] = limbs_h.index_select(0, plan["h3"]["idx_bwd"])
        pk["c3_bwd_h3"] = pk["c3_h3"][1:].flip(0).contiguous()
        # lin8's 256 feature rows for hold_gemm_h3 (its own scale)
        w8f = w8[:256].reshape(-1)
        _, ex8 = torch.frexp(w8f.abs().amax())
        sw8 = torch.ldexp(torch.ones((), device=dev), 14 - ex8)
        pk["w8_feat_h3"] = torch.stack(split_limbs_h(w8f * sw8)).reshape(-1).index_select(0, plan["h3"]["idx_w8"])
        pk["c3_w8"] = (1.0 / sw8).reshape(1)
    if rw is None:  # implicit net only (ImplicitNet.forward / gradient)
        return pk
    return _pack_render(pk, spec, rw, rb, need_bwd, dev)


def _pack_render(pk, spec, rw, rb, need_bwd, dev):
    r0 = torch.zeros(256, spec.Kr, device=dev)
    r0[:, :spec.rin_dim] = rw[0][:, rin_perm(spec.rin_dim, dev)]
    R = [r0, rw[1].contiguous(), rw[2].contiguous(), rw[3].contiguous(), rw[4].contiguous()]
    pk["R"] = R
    pk["rb"] = [b.contiguous() for b in rb]
    if need_bwd:
        RT123 = torch.stack(R[1:4]).transpose(1, 2).contiguous()
        pk["RT"] = [r0.t().contiguous(), RT123[0], RT123[1], RT123[2], None]  # the head's input gradient: hold_head3_bwd
    if config.x6() and spec.Kr <= 320 and spec.Kr % 16 == 0:  # hold_gemm_r6 packs: lin0..3 (and lin1..3 transposed), one gather
        plan = render_plan(spec.Kr, need_bwd, dev)
        mats = [torch.nn.functional.pad(r0, (0, 320 - spec.Kr)).reshape(-1), R[1].reshape(-1), R[2].reshape(-1), R[3].reshape(-1)]
        if need_bwd:
            mats.append(RT123.reshape(-1))
            # lin0's input gradient, the 256 feature columns of d_rin (rin's columns 0..255 are the feature block): the
            # remaining 16 / 48 columns (xc, normal, pose(, time)) stay a narrow hold_gemm_nt
            mats.append(r0[:, :256].t().contiguous().reshape(-1))
        limbs = torch.stack(split_limbs(torch.cat(mats))).reshape(-1)
        packs = limbs.index_select(0, plan["idx"]).split(plan["sizes"])
        pk["R_r6"] = list(packs[:4])
        if need_bwd:
            pk["RT_r6"] = [packs[7]] + list(packs[4:7])
        if config.h3():  # hold_gemm_h3 streams of the same matrices: per-matrix scales (one segmented maximum), two fp16 limbs
            flat = torch.cat(mats)
            # (the segments are contiguous: r0's 256 x 320 block, then 65 536 elements per matrix -- two plain reductions; a
            # scatter_reduce_("amax") over 600 k elements into 8 bins is 600 k contended float atomics: 17 ms per node and step,
            # which doubled the reference's 1 280-ray step until GPU call 12 of round 6 found it)
            n0_ = 256 * 320
            amax = torch.cat([flat[:n0_].abs().amax().view(1), flat[n0_:].view(-1, 65536).abs().amax(1)])
            _, ex = torch.frexp(amax)
            sw = torch.ldexp(torch.ones_like(amax), 14 - ex)
            limbs_h = torch.stack(split_limbs_h(flat * sw[plan["seg"]])).reshape(-1)
            ph = limbs_h.index_select(0, plan["idx_h3"]).split(plan["sizes_h3"])
            c3 = (1.0 / sw).contiguous()
            pk["R_h3"], pk["c3_R"] = list(ph[:4]), [c3[i:i + 1] for i in range(4)]
            if need_bwd:
                pk["RT_h3"], pk["c3_RT"] = [ph[7]] + list(ph[4:7]), [c3[7:8]] + [c3[i:i + 1] for i in range(4, 7)]
    return pk


class NodeField:
    """runs one node's field on a batch of points; owns its scratch pool."""

    def __init__(self, spec: FieldSpec, device):
        self.spec = spec
        self.pool = Pool(device)
        self.bpool = shared_pool(device)  # backward scratch (not needed across calls)
        self.device = device
        self.gen = 0  # bumped by every call that overwrites the saved activations (checked by the autograd glue)
        self._compact_wait = self._compact_gap = 0  # back-off of the compaction attempts (forward)

    # ------------------------------------------------------------------ deformation
    def _deform(self, x, P, ppf, dfm, want_w):
        """x [P,4] deformed-space points -> xc [P,4].  dfm: dict(tfs [B,nb,16], verts [B,778,3], skin_w)."""
        sp = self.spec
        xc = self.pool.get("xc", P, 4)
        if sp.n_bones == 1:
            K.invskin_fwd(x, P, ppf, None, dfm["tfs"], 1, xc)
            return xc, None
        w = self.pool.get("w_def", P, 16) if want_w else None
        K.knn_invlbs(x, P, ppf, dfm["verts"], dfm["skin_w"], tfs=dfm["tfs"], w_out=w, xc_out=xc)
        return xc, w

    # ------------------------------------------------------------------ implicit net trunk
    def _trunk(self, pk, xc, P, barf_w, keep_all, need_in0=True):
        sp, pool = self.spec, self.pool
        in0 = pool.get("in0", P, sp.K0)
        h = [pool.get(f"h{l}" if (keep_all or l == 3) else f"h_pp{l & 1}", P, 256) for l in range(8)]
        if USE_R6 and USE_CHAIN and keep_all and "trunk_r6" in pk:
            # register-resident trunk: embedding in-kernel, h_0..h_7 (and the skip columns of h_3) stored from the
            # accumulator registers; the embedding matrix itself is only needed by the backward (layer-0 weight gradient)
            if need_in0:
                K.embed_fwd(xc, 3, sp.L, P, in0, barf_w=barf_w)
            if USE_H3_TRUNK and "trunk_h3" in pk:
                K.trunk_h3(xc, P, pk["trunk_h3"], pk["bias8_h3"], pk["c3_h3"], barf_w, h, pk["trunk_r6"], pk["fused"][1])
            else:
                K.trunk_r6(xc, P, pk["trunk_r6"], pk["fused"][1], barf_w, h)
            return in0, h
        K.embed_fwd(xc, 3, sp.L, P, in0, out2=h[3][:, sp.skip_out:], barf_w=barf_w)
        W, b = pk["W"], pk["b"]
        if USE_CHAIN and keep_all:
            wpack, bias8 = pk["fused"]
            K.chain(K.CHAIN_SOFTPLUS, P, in0, wpack, 8, 5, skip_layer=3, side=in0, bias=[bias8[l] for l in range(8)],
                    out=h, wpack_x6=pk.get("chain_fwd_x6"))
            return in0, h
        G.gemm_nt(in0, W[0], h[0], bias=b[0], epi=G.EPI_SOFTPLUS, K=sp.K0)
        G.gemm_nt(h[0], W[1], h[1], bias=b[1], epi=G.EPI_SOFTPLUS)
        G.gemm_nt(h[1], W[2], h[2], bias=b[2], epi=G.EPI_SOFTPLUS)
        G.gemm_nt(h[2], W[3], h[3][:, :sp.skip_out], bias=b[3], epi=G.EPI_SOFTPLUS, N=sp.skip_out)
        G.gemm_nt(h[3], W[4], h[4], bias=b[4], epi=G.EPI_SOFTPLUS)
        G.gemm_nt(h[4], W[5], h[5], bias=b[5], epi=G.EPI_SOFTPLUS)
        G.gemm_nt(h[5], W[6], h[6], bias=b[6], epi=G.EPI_SOFTPLUS)
        G.gemm_nt(h[6], W[7], h[7], bias=b[7], epi=G.EPI_SOFTPLUS)
        return in0, h

    def _reverse_sweep(self, pk, h, t, ge, P, keep_all):
        """t_l = d sdf / d a_l for l = 7..0 and ge = d sdf / d embedding (what torch.autograd.grad(sdf, x) evaluates,
        volsdf_utils.py:79-93)."""
        sp, WT = self.spec, pk["WT"]
        K.seed_dsp(h[7], pk["w8_sdf"], 256, P, t[7])
        if USE_CHAIN:
            outs = [t[l - 1] if (keep_all or l - 1 in (0, 3)) else None for l in range(7, 0, -1)]
            K.chain(K.CHAIN_DSP, P, t[7], pk["chain_bwd"], 7, 32, skip_layer=3, aux1=[h[l - 1] for l in range(7, 0, -1)],
                    out=outs, wpack_x6=pk.get("chain_bwd_x6"), wpack_r6=pk.get("chain_bwd_r6") if USE_R6_BWD else None,
                    **self._h3_bwd(pk, "chain_bwd_h3", "c3_bwd_h3"))
            # raw columns 217.. of t_3 = d sdf / d (skip embedding): `ge` IS that view (self._ge_buffer), no copy
            assert ge.data_ptr() == t[3].data_ptr() + 4 * sp.skip_out
        else:
            for l in range(7, 0, -1):
                if l == 4:
                    G.gemm_nt(t[4], WT[4], t[3][:, :sp.skip_out], epi=G.EPI_MUL_DSP, aux1=h[3], N=256,
                              n_split=sp.skip_out, out_raw=ge)
                elif l == 3:
                    G.gemm_nt(t[3], WT[3], t[2], epi=G.EPI_MUL_DSP, aux1=h[2], K=sp.skip_pad)
                else:
                    G.gemm_nt(t[l], WT[l], t[l - 1], epi=G.EPI_MUL_DSP, aux1=h[l - 1])
        G.gemm_narrow(t[0], WT[0], ge, N=ge.shape[1], accumulate=True)

    @staticmethod
    def _h3_bwd(pk, stream, c3):
        """K.chain's operands of the f16x3 sweeps (hold_chain_h3) when the pack carries them"""
        if USE_H3_BWD and USE_R6_BWD and stream in pk:
            return dict(wpack_h3=pk[stream], c3=pk[c3])
        return {}

    def _ge_buffer(self, t, P):
        """d sdf / d embedding [P, E]: with the layer chains it stays where the descending sweep leaves its skip part --
        columns 217.. of t_3 (row stride 256) -- and layer 0's part is accumulated there; otherwise a [P, K0] buffer."""
        if USE_CHAIN:
            return t[3][:, self.spec.skip_out:]
        return self.pool.get("ge", P, self.spec.K0)

    def sdf_only(self, pk, x, P, ppf, dfm, barf_w, out_sdf):
        """no-grad SDF query of the sampler (sdf_func_with_deformer, volsdf_utils.py:150-169).  out_sdf [P,1]."""
        self.gen += 1  # overwrites the pooled canonical points a pending backward would read
        xc, _ = self._deform(x, P, ppf, dfm, want_w=False)
        if FUSED_SDF and USE_R6 and "trunk_h3" in pk:
            K.fused_sdf_h3(xc, P, pk["trunk_h3"], pk["bias8_h3"], pk["c3_h3"], pk["w8_sdf"], pk["b8_sdf"], barf_w, out_sdf,
                           pk["trunk_r6"], pk["fused"][1])
            return
        if FUSED_SDF and USE_R6 and "trunk_r6" in pk:
            K.fused_sdf_r6(xc, P, pk["trunk_r6"], pk["fused"][1], pk["w8_sdf"], pk["b8_sdf"], barf_w, out_sdf)
            return
        if FUSED_SDF and "fused_x6" in pk:
            K.fused_sdf_x6(xc, P, pk["fused_x6"], pk["fused"][1], pk["w8_sdf"], float(pk["b8_sdf"]), barf_w, out_sdf)
            return
        if FUSED_SDF:
            wpack, bias8 = pk["fused"]
            K.fused_sdf(xc, P, wpack, bias8, pk["w8_sdf"], float(pk["b8_sdf"]), barf_w, out_sdf)
            return
        # layer-by-layer variant (ping-pong activations: two [P,256] buffers + the skip buffer stay live)
        _, h = self._trunk(pk, xc, P, barf_w, keep_all=False)
        K.rowdot(h[7], pk["w8_sdf"], 256, pk["b8_sdf"], P, out_sdf)

    # ------------------------------------------------------------------ full forward
    def forward(self, pk, x, P, ppf, dfm, barf_w, pose_embed, time_code, training, beta=None):
        """-> dict(sdf [P,1], rgb [P,4], normal [P,3], xc, feat, grad).
        ``beta`` (the node's Laplace density beta, as the compositor will receive it): enables EXACT SAMPLE COMPACTION
        (csrc/compact.hip; single- and multi-frame calls, _compaction) -- after the trunk and the sdf row, the samples whose density and whose
        density derivatives are exact fp32 zeros are dropped: the reverse sweep, the normal, the colour net and the whole
        backward run on the compacted rows, the dropped rows of rgb / normal are zeros (the compositor multiplies them by a
        weight of exactly 0).  Outputs are bit-identical to the uncompacted path; nothing changes when every sample is live."""
        sp, pool = self.spec, self.pool
        self.gen += 1
        pool.min_rows = self.bpool.min_rows = P
        xc, w_def = self._deform(x, P, ppf, dfm, want_w=training)
        in0, h = self._trunk(pk, xc, P, barf_w, keep_all=True, need_in0=training)
        sdf = pool.get("sdf", P, 1)
        K.rowdot(h[7], pk["w8_sdf"], 256, pk["b8_sdf"], P, sdf)
        self.last_live = None
        plan = None
        if COMPACT and beta is not None and P >= max(1, COMPACT_MIN_POINTS):
            # every attempt costs a host read (the live count sizes the launches behind it).  While attempts keep finding nothing
            # worth dropping -- the reference's initial beta = 0.1 has no dead sample anywhere in the scene -- they are spaced out:
            # 1, 2, 4 .. COMPACT_BACKOFF calls apart; a successful attempt resets the spacing.  Skipping an attempt only skips an
            # optimisation, results are the same either way.
            if self._compact_wait > 0:
                self._compact_wait -= 1
                self.last_live = ("attempt skipped", self._compact_wait)
            else:
                plan = self._compaction(sdf, P, ppf, beta)
                self._compact_gap = 0 if plan is not None else min(max(1, 2 * self._compact_gap), COMPACT_BACKOFF)
                self._compact_wait = max(0, self._compact_gap - 1)
        if plan is None:
            out = self._forward_tail(pk, xc, w_def, in0, h, P, ppf, dfm, barf_w, pose_embed, time_code, training)
            self.saved.update(sdf=sdf, cidx=None, P_full=P)
            out.update(sdf=sdf, xc=xc)
            return out
        # ---- compacted tail ----
        gidx, sidx, n_sc, Pc, ppf_c = plan
        rgb_f, nrm_f = pool.get("rgb_full", P, 4), pool.get("n_full", P, 4)
        rgb_f.zero_()
        nrm_f.zero_()
        if Pc == 0:  # nothing along these rays: no colour, no normal, no gradient
            self.saved = dict(P=0, P_full=P, cidx=gidx, pk=pk, sdf=sdf)
            return dict(sdf=sdf, rgb=rgb_f, normal=nrm_f[:, :3], xc=xc, feat=None, grad=None)
        gather = lambda src, name, cols: torch.index_select(src, 0, gidx, out=pool.get(name, Pc, cols))
        xc_c = gather(xc, "c_xc", 4)
        h_c = [gather(h[l], f"c_h{l}", 256) for l in range(8)]
        in0_c = gather(in0, "c_in0", sp.K0) if training else in0
        w_c = None if w_def is None else gather(w_def, "c_wdef", w_def.shape[1])
        out = self._forward_tail(pk, xc_c, w_c, in0_c, h_c, Pc, ppf_c, dfm, barf_w, pose_embed, time_code, training)
        rgb_f.index_copy_(0, sidx, out["rgb"][:n_sc])
        nrm_f[:, :3].index_copy_(0, sidx, out["normal"][:n_sc])
        self.saved.update(sdf=sdf, cidx=gidx, P_full=P)
        return dict(sdf=sdf, rgb=rgb_f, normal=nrm_f[:, :3], xc=xc, feat=out["feat"], grad=out["grad"])

    def _compaction(self, sdf, P, ppf, beta):
        """-> None (every stage on every sample) or (gidx [Pc] rows the compacted stages run on, sidx / n_sc: the first n_sc of
        their results are scattered back to the rows sidx, Pc, points per frame of the compacted layout).  ONE host read either
        way (the live count sizes the launches that follow).
        Single frame (ppf == P): the ordered live list, padded to a multiple of COMPACT_ALIGN with copies of one dead sample.
        Several frames (round 6; the reference's own training batch is 10 frames x 128 rays): the stages behind the trunk find
        a row's frame as row // points-per-frame (pose embedding, bone transforms, the per-frame reductions of the backward), so
        every frame keeps ONE common row count ppf_c = the largest live count of a frame, rounded up to COMPACT_ALIGN: frame
        f's rows are its live samples in order, then as many of ITS OWN dead samples as fill the frame up -- real samples with
        exactly zero cotangents, so they add exact zeros to every gradient sum, and what the forward computes for them is what
        the uncompacted path computes there (scattered back with the rest; the compositor multiplies it by a weight of 0)."""
        dev = sdf.device
        if ppf == P:
            cidx = K.alive_index(sdf, P, beta)
            if cidx is None:
                self.last_live = None
                return None
            Pa = int(cidx.numel())
            Pc = (Pa + COMPACT_ALIGN - 1) // COMPACT_ALIGN * COMPACT_ALIGN
            if Pc > COMPACT_MAX_LIVE * P:  # too few dead samples to pay for the gathers
                self.last_live = (Pa, P, "not compacted")
                return None
            self.last_live = (Pa, P)
            if Pc > Pa:
                # the fast kernels want row counts that are multiples of 16 (whole-dW weight gradients) / 128 (point blocks): the
                # list is padded with copies of ONE DEAD sample -- its cotangents are exact zeros; not scattered back
                live = torch.zeros(P, dtype=torch.int8, device=dev)
                live[cidx] = 1
                dead = torch.argmin(live).reshape(1)  # index of the first dead sample (no host read)
                gidx = torch.cat([cidx, dead.expand(Pc - Pa)])
            else:
                gidx = cidx
            return gidx, cidx, Pa, Pc, Pc
        B = P // ppf
        live = K.alive_mask(sdf, P, beta).view(B, ppf)
        csum = torch.cumsum(live, 1, dtype=torch.int32)  # rank + 1 of a live sample among its frame's live ones
        cnt = csum[:, -1]
        mx, Pa = (int(v) for v in torch.stack([cnt.max(), cnt.sum()]).cpu())  # the host read
        if Pa == P:
            self.last_live = None
            return None
        ppf_c = (mx + COMPACT_ALIGN - 1) // COMPACT_ALIGN * COMPACT_ALIGN
        if B * ppf_c > COMPACT_MAX_LIVE * P or ppf_c > ppf:
            self.last_live = (Pa, P, "not compacted")
            return None
        self.last_live = (Pa, P)
        if mx == 0:
            return torch.empty(0, dtype=torch.int64, device=dev), None, 0, 0, 0
        # position of sample j in its frame's order "live samples ascending, then dead samples ascending": a permutation of the
        # frame; its first ppf_c entries are the rows the compacted stages run on
        j = torch.arange(ppf, dtype=torch.int32, device=dev).view(1, ppf)
        pos = torch.where(live != 0, csum - 1, cnt.view(B, 1) + j - csum).long()  # (dead sample j: cnt + number of dead before it)
        order = torch.empty(B, ppf, dtype=torch.int64, device=dev).scatter_(1, pos, j.long().expand(B, ppf))
        gidx = (order[:, :ppf_c] + torch.arange(B, device=dev).view(B, 1) * ppf).reshape(-1)
        return gidx, gidx, B * ppf_c, B * ppf_c, ppf_c

    def _forward_tail(self, pk, xc, w_def, in0, h, P, ppf, dfm, barf_w, pose_embed, time_code, training):
        """everything behind the trunk and the sdf row, on P rows (all samples, or the compacted live ones): lin8's feature
        rows, the reverse sweep and the canonical normal, the rendering net; fills self.saved for backward()."""
        sp, pool = self.spec, self.pool
        rin = pool.get("rin", P, sp.Kr)
        # lin8 = 256 feature rows as a full-tile GEMM + the sdf row as a row dot (N = 257 would add a 256-wide tile for it)
        h3g = USE_H3_GEMM and USE_R6_GEMM and "w8_feat_h3" in pk and "R_h3" in pk
        # row maxima travelling from launch to launch (hold_gemm_h3: every point's operand row is scaled by its own power of two)
        amx = [pool.get(f"amx{i}", P, 1) for i in range(2)] if h3g else None
        if h3g:  # input: the trunk's last softplus output (no reported maximum: the floor); output maxima -> lin0's features
            G.gemm_h3(h[7], pk["w8_feat_h3"], pk["c3_w8"], rin[:, RIN_FEAT:RIN_FEAT + FEAT], K=256, wpack_r6=pk["w8_feat_r6"],
                      bias=pk["b8_feat"], amax_floor=H3_ROW_FLOOR, amax_out=amx[0])
        elif USE_R6_GEMM and "w8_feat_r6" in pk:
            G.gemm_r6(h[7], pk["w8_feat_r6"], rin[:, RIN_FEAT:RIN_FEAT + FEAT], K=256, bias=pk["b8_feat"])
        else:
            G.gemm_nt(h[7], pk["W8_feat"], rin[:, RIN_FEAT:RIN_FEAT + FEAT], bias=pk["b8_feat"], N=256)
        # ---- reverse sweep: t_l = d sdf / d a_l, ge = d sdf / d embed, g = d sdf / d xc ----
        WT = pk["WT"]
        # t[3] always has its own buffer: its columns 217..219 are K-padding of the next GEMM and must stay zero
        t = [pool.get(f"t{l}" if (training or l == 3) else f"t_pp{l & 1}", P, 256) for l in range(8)]
        ge = self._ge_buffer(t, P)
        self._reverse_sweep(pk, h, t, ge, P, keep_all=training)
        g = pool.get("g", P, 4)
        K.embed_bwd(xc, sp.L, P, ge, g, barf_w=barf_w)
        # ---- canonical normal + render-net input assembly ----
        if sp.n_bones == 1:
            w_c = None
        else:
            w_c = pool.get("w_cano", P, 16)
            K.knn_invlbs(xc, P, ppf, dfm["verts_c"], dfm["skin_w"], w_out=w_c)
        K.copy_cols(xc, rin[:, RIN_X:RIN_X + 3], 3, P)
        K.normal_fwd(g, w_c, dfm["tfs"], sp.n_bones, P, ppf, rin[:, RIN_N:RIN_N + 3])
        K.frame_bcast(pose_embed, P, ppf, rin, RIN_POSE)
        if sp.time:
            K.frame_bcast(time_code, P, ppf, rin, RIN_TIME)
        # ---- rendering net ----
        R, rb = pk["R"], pk["rb"]
        r = [pool.get(f"r{l}", P, 256) for l in range(4)]
        rbits = None
        if h3g:
            # the ReLU masks of lin0..2 as one bit per element (32 bytes per point and layer): what the backward's masked input-
            # gradient launches read instead of streaming r_0..r_2 a second time (hold_gemm_h3_bits)
            if training and USE_RELU_BITS:
                rbits = [pool.get(f"rbits{l}", P, 8, torch.int32) for l in range(3)]
            # lin0 reads [features | xc | normal | pose | (time)]: the features' maxima are exact, the floor bounds the rest
            G.gemm_h3(rin, pk["R_h3"][0], pk["c3_R"][0], r[0], K=sp.Kr, wpack_r6=pk["R_r6"][0], bias=rb[0], epi=G.R6_RELU,
                      amax_in=amx[0], amax_floor=H3_ROW_FLOOR, amax_out=amx[1], bits_out=rbits[0] if rbits else None)
            for l in (1, 2, 3):
                G.gemm_h3(r[l - 1], pk["R_h3"][l], pk["c3_R"][l], r[l], K=256, wpack_r6=pk["R_r6"][l], bias=rb[l], epi=G.R6_RELU,
                          amax_in=amx[l & 1], amax_out=amx[(l + 1) & 1] if l < 3 else None,
                          bits_out=rbits[l] if (rbits and l < 3) else None)
        elif USE_R6_GEMM and "R_r6" in pk:
            G.gemm_r6(rin, pk["R_r6"][0], r[0], K=sp.Kr, bias=rb[0], epi=G.R6_RELU)
            for l in (1, 2, 3):
                G.gemm_r6(r[l - 1], pk["R_r6"][l], r[l], K=256, bias=rb[l], epi=G.R6_RELU)
        else:
            G.gemm_nt(rin, R[0], r[0], bias=rb[0], epi=G.EPI_RELU, K=sp.Kr)
            G.gemm_nt(r[0], R[1], r[1], bias=rb[1], epi=G.EPI_RELU)
            G.gemm_nt(r[1], R[2], r[2], bias=rb[2], epi=G.EPI_RELU)
            G.gemm_nt(r[2], R[3], r[3], bias=rb[3], epi=G.EPI_RELU)
        rgb = pool.get("rgb", P, 4)
        G.head3_fwd(r[3], R[4], rb[4], rgb)  # 3-output head: a streaming kernel, not a 3/256-full GEMM tile
        self.saved = dict(P=P, ppf=ppf, xc=xc, w_def=w_def, w_c=w_c, in0=in0, h=h, t=t, ge=ge, g=g, rin=rin, r=r,
                          rgb=rgb, dfm=dfm, barf_w=barf_w, pk=pk, rbits=rbits)
        return dict(rgb=rgb, normal=rin[:, RIN_N:RIN_N + 3], feat=rin[:, RIN_FEAT:RIN_FEAT + FEAT], grad=g)

    # ------------------------------------------------------------------ shared backward sweeps
    def _second_order_sweep(self, pk, h, t, gebar, dW, P, grp=None, side_in_t3=False):
        """ascending sweep of the double backward: tbar_l = W_l vbar_l, ubar_l = tbar_l * s_l,
        a2_l = 100 * tbar_l * t_l * (1 - s_l); dW_l += t_l^T vbar_l.  Returns (a2[8], ubar_7)."""
        sp, pool, W = self.spec, self.bpool, pk["W"]
        a2 = [pool.get(f"a2_{l}", P, 256) for l in range(8)]
        if USE_CHAIN:
            vb = [pool.get(f"vbc{l}", P, 256) for l in range(8)]
            wr6 = pk.get("trunk_r6") if USE_R6_BWD else None
            if wr6 is not None and not side_in_t3:
                # register-resident sweep (csrc/rchain.hip): it takes the skip layer's side columns (the next input's columns
                # 217.. = the embedding cotangent) from aux2[3][:, 217:]; the callers let hold_embed_bwd2 write them there
                # (side_in_t3), this copy is for anyone who did not
                K.copy_cols(gebar, t[3][:, sp.skip_out:], sp.E, P)
            K.chain(K.CHAIN_DBWD, P, gebar, pk["fused"][0], 8, 5, skip_layer=3, side=gebar, aux1=h, aux2=t, out=vb,
                    out2=a2, wpack_x6=pk.get("chain_fwd_x6"), wpack_r6=wr6, **self._h3_bwd(pk, "trunk_h3", "c3_h3"))
            G.wgrad(t[0], gebar, dW[0], None, K=sp.K0, accumulate=True)
            # t_l, vbar_l live in buffers of their own until the next backward: their weight gradients may wait for the
            # grouped launch at the end of the caller (G.WgradGroup)
            wg = G.wgrad if grp is None else grp.add
            for l in range(1, 8):
                if l == 3:
                    wg(t[3], vb[2], dW[3], None, N=sp.skip_out, accumulate=True)
                else:
                    wg(t[l], vb[l - 1], dW[l], None, accumulate=True)
            return a2, vb[7]
        vb = [pool.get(f"vb{i}", P, 256) for i in range(2)]
        # l = 0: tbar_0 = W0 vbar_0 ; ubar_0 = tbar*s ; a2_0 = 100*tbar*t*(1-s)
        G.wgrad(t[0], gebar, dW[0], None, K=sp.K0, accumulate=True)
        G.gemm_nt(gebar, W[0], vb[0], epi=G.EPI_DBWD, aux1=h[0], aux2=t[0], out2=a2[0], K=sp.K0)
        cur = vb[0]
        for l in range(1, 8):
            nxt = vb[1] if cur is vb[0] else vb[0]
            if l == 3:
                G.wgrad(t[3], cur, dW[3], None, N=sp.skip_out, accumulate=True)
                # output lands in the first 217 columns of the vbar_4 buffer; the skip part is gebar
                G.gemm_nt(cur, W[3], nxt[:, :sp.skip_out], epi=G.EPI_DBWD, aux1=h[3], aux2=t[3], out2=a2[3],
                          N=sp.skip_out)
                K.copy_cols(gebar, nxt[:, sp.skip_out:], sp.E, P)
            else:
                G.wgrad(t[l], cur, dW[l], None, accumulate=True)
                G.gemm_nt(cur, W[l], nxt, epi=G.EPI_DBWD, aux1=h[l], aux2=t[l], out2=a2[l])
            cur = nxt
        return a2, cur

    def _first_order_sweep(self, pk, h, a2, r7, in0, dW, dWb, ebar, P, grp=None):
        """descending sweep r_{l-1} = (W_l^T r_l) * s_{l-1} + a2_{l-1} from r_7 down to r_0 with
        dW_l += r_l^T in_l, db_l += sum r_l; with `ebar` true the skip columns of W_4^T r_4 are kept.
        Returns (r_0, ebar) -- with the layer chains ebar is the view r_3[:, 217:] (row stride 256) the sweep wrote them
        to (no copy; r_3's weight gradient only reduces the rows < 217), otherwise a [P, K0] buffer; None if not asked."""
        sp, pool, WT = self.spec, self.bpool, pk["WT"]
        if USE_CHAIN:
            r = [pool.get(f"rbc{l}", P, 256) for l in range(7)] + [r7]
            K.chain(K.CHAIN_DSP, P, r7, pk["chain_bwd"], 7, 32, skip_layer=3, aux1=[h[l - 1] for l in range(7, 0, -1)],
                    aux2=None if a2 is None else [a2[l - 1] for l in range(7, 0, -1)],
                    out=[r[l - 1] for l in range(7, 0, -1)], wpack_x6=pk.get("chain_bwd_x6"),
                    # register-resident sweep, with or without the additive side input: since round 4 its side traffic moves
                    # as whole 128-byte lines through LDS (rtile_kernel: 140 vs 122 TF-eq for hold_chain_x6 with a2)
                    wpack_r6=pk.get("chain_bwd_r6") if USE_R6_BWD else None,
                    **self._h3_bwd(pk, "chain_bwd_h3", "c3_bwd_h3"))
            ebar = r[3][:, sp.skip_out:] if ebar else None
            wg = G.wgrad if grp is None else grp.add
            for l in range(7, 0, -1):
                if l == 3:
                    wg(r[3], h[2], dW[3], dWb[3], N=sp.skip_out, accumulate=True)
                else:
                    wg(r[l], h[l - 1], dW[l], dWb[l], accumulate=True)
            return r[0], ebar
        rb_ = [pool.get(f"rb{i}", P, 256) for i in range(2)]
        ebar = pool.get("ebar", P, sp.K0) if ebar else None
        cur = r7
        A2 = lambda i: None if a2 is None else a2[i]
        for l in range(7, 0, -1):
            nxt = rb_[1] if cur is rb_[0] else rb_[0]
            if l == 4:
                G.wgrad(cur, h[3], dW[4], dWb[4], accumulate=True)
                if ebar is not None:
                    G.gemm_nt(cur, WT[4], nxt[:, :sp.skip_out], epi=G.EPI_MUL_DSP, aux1=h[3], aux2=A2(3), N=256,
                              n_split=sp.skip_out, out_raw=ebar)
                else:
                    G.gemm_nt(cur, WT[4], nxt[:, :sp.skip_out], epi=G.EPI_MUL_DSP, aux1=h[3], aux2=A2(3),
                              N=sp.skip_out)
                # keep K-padding columns of r_3 (217..219) zero for the next GEMM
                nxt[:, sp.skip_out:sp.skip_pad].zero_()
            elif l == 3:
                G.wgrad(cur, h[2], dW[3], dWb[3], N=sp.skip_out, accumulate=True)
                G.gemm_nt(cur, WT[3], nxt, epi=G.EPI_MUL_DSP, aux1=h[2], aux2=A2(2), K=sp.skip_pad)
            else:
                G.wgrad(cur, h[l - 1], dW[l], dWb[l], accumulate=True)
                G.gemm_nt(cur, WT[l], nxt, epi=G.EPI_MUL_DSP, aux1=h[l - 1], aux2=A2(l - 1))
            cur = nxt
        return cur, ebar

    # ------------------------------------------------------------------ ImplicitNet.forward (shape_net.py:84-130)
    def sdf_feat_forward(self, pk, xc, P, barf_w):
        """canonical points xc [P,4] -> out [P,260] = (256 features | sdf | pad); keeps h for sdf_feat_backward."""
        sp, pool = self.spec, self.pool
        self.gen += 1
        in0, h = self._trunk(pk, xc, P, barf_w, keep_all=True)
        out = pool.get("oc_out", P, 260)
        G.gemm_nt(h[7], pk["W"][8], out, bias=pk["b"][8], N=257)
        self.saved = dict(P=P, xc=xc, in0=in0, h=h, barf_w=barf_w, pk=pk)
        return out

    def sdf_feat_backward(self, ob):
        """first-order backward of sdf_feat_forward: ob [P,260] = d/d(features | sdf | 0) -> (g_iw[9], g_ib[9], d_x [P,4])."""
        sp, pool, sv = self.spec, self.bpool, self.saved
        P, pk, h, xc = sv["P"], sv["pk"], sv["h"], sv["xc"]
        dev = self.device
        W, WT = pk["W"], pk["WT"]
        dW, dWb = zeros_like_many(W, pk["b"])
        G.wgrad(ob, h[7], dW[8], dWb[8], N=257, accumulate=True)
        r7 = pool.get("r7", P, 256)
        G.gemm_nt(ob, WT[8], r7, epi=G.EPI_MUL_DSP, aux1=h[7], K=260)
        grp = G.WgradGroup() if (USE_CHAIN and USE_WGRAD_GROUP) else None
        cur, ebar = self._first_order_sweep(pk, h, None, r7, sv["in0"], dW, dWb, True, P, grp)
        G.wgrad(cur, sv["in0"], dW[0], dWb[0], K=sp.K0, accumulate=True)
        if grp is not None:
            grp.flush()
        G.gemm_narrow(cur, WT[0], ebar, N=ebar.shape[1], accumulate=True)
        xbar = pool.get("xbar", P, 4)
        K.embed_bwd(xc, sp.L, P, ebar, xbar, barf_w=sv["barf_w"])
        d0 = torch.zeros(256, pk["iw0_cols"], device=dev)
        d0[:, :sp.E] = dW[0][:, :sp.E]
        g_iw = [d0, dW[1], dW[2], dW[3], dW[4] / math.sqrt(2), dW[5], dW[6], dW[7],
                torch.cat([dW[8][256:257], dW[8][:256]], 0)]
        g_ib = dWb[:8] + [torch.cat([dWb[8][256:257], dWb[8][:256]])]
        return g_iw, g_ib, xbar

    # ------------------------------------------------------------------ eikonal samples (a13)
    def grad_points_forward(self, pk, xc, P, barf_w):
        """d sdf / d x at free canonical points (compute_gradient_samples, volsdf_utils.py:19-48): trunk forward +
        reverse sweep only (no deformer, no colour net).  Keeps h / t / ge for grad_points_backward."""
        sp, pool = self.spec, self.pool
        self.gen += 1
        in0, h = self._trunk(pk, xc, P, barf_w, keep_all=True)
        WT = pk["WT"]
        t = [pool.get(f"t{l}", P, 256) for l in range(8)]
        ge = self._ge_buffer(t, P)
        self._reverse_sweep(pk, h, t, ge, P, keep_all=True)
        g = pool.get("g", P, 4)
        K.embed_bwd(xc, sp.L, P, ge, g, barf_w=barf_w)
        self.saved = dict(P=P, xc=xc, in0=in0, h=h, t=t, ge=ge, g=g, barf_w=barf_w, pk=pk)
        return g

    def grad_points_backward(self, gbar):
        """gradient of a loss on g = d sdf/d x w.r.t. the effective implicit weights (second-order terms only:
        the points themselves and sdf / features carry no upstream gradient)."""
        sp, pool, sv = self.spec, self.bpool, self.saved
        self._consume(sv, "grad_points_backward")
        P, pk, h, t, xc = sv["P"], sv["pk"], sv["h"], sv["t"], sv["xc"]
        dev = self.device
        W, WT = pk["W"], pk["WT"]
        dW, dWb = zeros_like_many(W, pk["b"])
        gb = pool.get("gbar", P, 4)
        K.copy_cols(gbar.contiguous(), gb, 3, P)
        gebar = pool.get("gebar", P, sp.K0)
        # in chain mode ge lives in t_3's skip columns, where the ascending sweep wants gebar next: written in place
        K.embed_bwd2(xc, sp.L, P, sv["ge"], gb, gebar, xbar=None, barf_w=sv["barf_w"],
                     gebar2=t[3][:, sp.skip_out:] if USE_CHAIN else None)
        grp = G.WgradGroup() if (USE_CHAIN and USE_WGRAD_GROUP) else None
        a2, u7 = self._second_order_sweep(pk, h, t, gebar, dW, P, grp, side_in_t3=USE_CHAIN)
        d_w8sdf = torch.zeros(256, device=dev)
        G.wcolsum(u7, d_w8sdf, N=256)
        # first-order sweep driven only by the second-order terms a2_l (out_bar = 0  =>  r_7 = a2_7)
        cur, _ = self._first_order_sweep(pk, h, a2, a2[7], sv["in0"], dW, dWb, False, P, grp)
        G.wgrad(cur, sv["in0"], dW[0], dWb[0], K=sp.K0, accumulate=True)
        if grp is not None:
            grp.flush()
        d0 = torch.zeros(256, pk["iw0_cols"], device=dev)
        d0[:, :sp.E] = dW[0][:, :sp.E]
        g_iw = [d0, dW[1], dW[2], dW[3], dW[4] / math.sqrt(2), dW[5], dW[6], dW[7]]
        d8 = torch.zeros(257, 256, device=dev)
        d8[0] = d_w8sdf
        g_iw.append(d8)
        g_ib = dWb[:8] + [torch.zeros(257, device=dev)]
        return g_iw, g_ib

    @staticmethod
    def _consume(sv, what):
        """ONE backward per forward (advisor r4): the second-order backward overwrites saved forward state in place -- d sdf /
        d embedding lives in t_3's skip columns and hold_embed_bwd2 writes the embedding cotangent over it -- so a second
        backward over the same saved state (retain_graph, calling backward twice) would read the first one's cotangents as
        forward values and return wrong second-order terms: refused instead."""
        if sv.get("consumed"):
            raise RuntimeError(f"hold_amd: NodeField.{what}() called twice for one forward: the backward consumes the saved "
                               "activations in place (one-shot contract); run the forward again")
        sv["consumed"] = True

    # ------------------------------------------------------------------ backward
    def backward(self, d_sdf, d_rgb, d_normal, n_frames):
        """d_sdf [P] / [P,1], d_rgb [P,3], d_normal [P,3] (may be None) -> dict of gradients:
        iw[9], ib[9] (w.r.t. the EFFECTIVE implicit weights as passed to pack_weights), rw[5], rb[5],
        tfs [B,nb,16], pose_embed [B,8], time_code [B,32] (object)."""
        sp, pool = self.spec, self.bpool
        sv = self.saved
        self._consume(sv, "backward")
        pk = sv["pk"]
        dev = self.device
        W, WT, R, RT = pk["W"], pk["WT"], pk["R"], pk["RT"]
        dR, dRb, dW, dWb = zeros_like_many(R, pk["rb"], W, pk["b"])
        if sv.get("cidx") is not None:  # compacted forward: the cotangents of the dropped samples are exact zeros (csrc/compact.hip)
            cidx = sv["cidx"]
            if sv["P"] == 0:  # no live sample: every gradient is zero
                g0 = torch.zeros(256, sp.rin_dim, device=dev)
                d0 = torch.zeros(256, pk["iw0_cols"], device=dev)
                return dict(iw=[d0] + dW[1:8] + [torch.zeros(257, 256, device=dev)], ib=dWb[:8] + [torch.zeros(257, device=dev)],
                            rw=[g0, dR[1], dR[2], dR[3], dR[4]], rb=dRb, tfs=torch.zeros(n_frames, sp.n_bones, 16, device=dev),
                            pose_embed=torch.zeros(n_frames, 8, device=dev),
                            time_code=torch.zeros(n_frames, sp.time, device=dev) if sp.time else None)
            d_sdf = d_sdf.reshape(-1).index_select(0, cidx)
            d_rgb = d_rgb.index_select(0, cidx)
            d_normal = None if d_normal is None else d_normal.index_select(0, cidx)
        P, ppf = sv["P"], sv["ppf"]
        h, t, rin, r, rgb, xc = sv["h"], sv["t"], sv["rin"], sv["r"], sv["rgb"], sv["xc"]
        rbits = sv.get("rbits")
        # ---------- rendering net ----------
        dy = pool.get("dy4", P, 4)
        sg = rgb[:, :3]
        dy[:, :3] = d_rgb * sg * (1.0 - sg)  # sigmoid'  (3 columns, elementwise)
        # one grouped launch for the 256 x 256 weight gradients of the whole backward (3 of the colour net, 14 + 1 of the
        # implicit net): their operands must then survive to the end, so the colour net's cotangents get four buffers
        # instead of two in rotation
        grp = G.WgradGroup() if (USE_CHAIN and USE_WGRAD_GROUP) else None
        rr = [pool.get(f"rr{i}", P, 256) for i in range(2 if grp is None else 4)]
        db4 = torch.zeros(4, device=dev)
        G.head3_bwd(dy, r[3], R[4], rr[1], dR[4], db4)  # input gradient + weight / bias gradients in one pass over r3
        dRb[4] = db4[:3]
        cur, ci = rr[1], 1
        h3g = USE_H3_GEMM and USE_R6_GEMM and "RT_h3" in pk
        if h3g:
            # row bound of d r_3 = (R_3 > 0) * (dy . W_4): |dy_0| + |dy_1| + |dy_2| times the largest |W_4| -- an upper bound a few
            # times above the row's maximum costs the two-limb split nothing (full precision over 14 binades below the bound)
            bamx = [self.bpool.get(f"bamx{i}", P, 1) for i in range(2)]
            torch.mul(dy[:, :3].abs().sum(1, keepdim=True), R[4].abs().amax(), out=bamx[1])
            bi = 1
        for l in (3, 2, 1):
            if grp is None:
                G.wgrad(cur, r[l - 1], dR[l], dRb[l])
            else:
                grp.add(cur, r[l - 1], dR[l], dRb[l])
            ci = (ci + 1) % len(rr)
            nxt = rr[ci]
            if h3g:
                G.gemm_h3(cur, pk["RT_h3"][l], pk["c3_RT"][l], nxt, K=256, wpack_r6=pk["RT_r6"][l], epi=G.R6_MASK, aux=r[l - 1],
                          amax_in=bamx[bi], amax_out=bamx[bi ^ 1], bits_in=rbits[l - 1] if rbits else None)
                bi ^= 1
            elif USE_R6_GEMM and "RT_r6" in pk:
                G.gemm_r6(cur, pk["RT_r6"][l], nxt, K=256, epi=G.R6_MASK, aux=r[l - 1])
            else:
                G.gemm_nt(cur, RT[l], nxt, epi=G.EPI_MUL_DRELU, aux1=r[l - 1])
            cur = nxt
        G.wgrad(cur, rin, dR[0], dRb[0], K=sp.Kr)
        d_rin = pool.get("d_rin", P, sp.Kr)
        if h3g:
            G.gemm_h3(cur, pk["RT_h3"][0], pk["c3_RT"][0], d_rin[:, :FEAT], K=256, wpack_r6=pk["RT_r6"][0], amax_in=bamx[bi])
            G.gemm_narrow(cur, RT[0][FEAT:], d_rin[:, FEAT:], N=sp.Kr - FEAT)
        elif USE_R6_GEMM and "RT_r6" in pk:  # the 256 feature columns register-resident, the 16 / 48 others a narrow GEMM
            G.gemm_r6(cur, pk["RT_r6"][0], d_rin[:, :FEAT], K=256)
            G.gemm_narrow(cur, RT[0][FEAT:], d_rin[:, FEAT:], N=sp.Kr - FEAT)
        else:
            G.gemm_nt(cur, RT[0], d_rin, N=sp.Kr)
        B = n_frames
        d_pose = torch.zeros(B, 8, device=dev)
        K.frame_colsum(d_rin, RIN_POSE, 8, P, ppf, d_pose)
        d_time = None
        if sp.time:
            d_time = torch.zeros(B, sp.time, device=dev)
            K.frame_colsum(d_rin, RIN_TIME, sp.time, P, ppf, d_time)
        # ---------- normal ----------
        nbar = pool.get("nbar", P, 4)
        K.copy_cols(d_rin[:, RIN_N:RIN_N + 3], nbar, 3, P)
        if d_normal is not None:
            K.copy_cols(d_normal, nbar, 3, P, accumulate=True)
        gbar = pool.get("gbar", P, 4)
        dtfs = torch.zeros(B, sp.n_bones, 16, device=dev)
        K.normal_bwd(sv["g"], sv["w_c"], sv["dfm"]["tfs"], sp.n_bones, P, ppf, nbar, gbar, dtfs)
        # ---------- second-order path: gbar -> embedding -> ascending sweep ----------
        xbar = pool.get("xbar", P, 4)
        K.copy_cols(d_rin[:, RIN_X:RIN_X + 3], xbar, 3, P)
        gebar = pool.get("gebar", P, sp.K0)
        K.embed_bwd2(xc, sp.L, P, sv["ge"], gbar, gebar, xbar=xbar, barf_w=sv["barf_w"],
                     gebar2=t[3][:, sp.skip_out:] if USE_CHAIN else None)
        a2, u7 = self._second_order_sweep(pk, h, t, gebar, dW, P, grp, side_in_t3=USE_CHAIN)
        # gradient of the sdf row of W8: ubar_7 (second-order path) + the first-order term d_sdf^T h7, both as
        # deterministic (weighted) column sums -- a 257th wgrad row would cost a whole 128-row tile
        d_sdf = d_sdf.reshape(P)
        G.wcolsum(u7, dW[8][256], N=256, accumulate=True)
        G.wcolsum(h[7], dW[8][256], weights=d_sdf, N=256, accumulate=True)
        dWb[8][256:257] += d_sdf.sum().reshape(1)
        # ---------- first-order backward sweep ----------
        # cotangent of lin8's output = [d_rin's feature block | d_sdf]: the feature block is used in place (no copy),
        # the sdf column enters the input-gradient GEMM as a rank-1 term d_sdf[p] * w8_sdf[n]
        d_feat = d_rin[:, RIN_FEAT:RIN_FEAT + FEAT]
        (G.wgrad if grp is None else grp.add)(d_feat, h[7], dW[8], dWb[8], N=256, accumulate=True)
        r7 = pool.get("r7", P, 256)
        G.gemm_nt(d_feat, pk["WT8_feat"], r7, epi=G.EPI_MUL_DSP, aux1=h[7], aux2=a2[7], K=256, r1_row=d_sdf,
                  r1_col=pk["w8_sdf"])
        cur, ebar = self._first_order_sweep(pk, h, a2, r7, sv["in0"], dW, dWb, True, P, grp)
        G.wgrad(cur, sv["in0"], dW[0], dWb[0], K=sp.K0, accumulate=True)
        G.gemm_narrow(cur, WT[0], ebar, N=ebar.shape[1], accumulate=True)
        K.embed_bwd(xc, sp.L, P, ebar, xbar, barf_w=sv["barf_w"], accumulate=True)
        if grp is not None:
            grp.flush()
        # ---------- deformation ----------
        K.invskin_bwd(xc, sv["w_def"], sv["dfm"]["tfs"], sp.n_bones, P, ppf, xbar, dtfs)
        # ---------- map back to the layouts of the effective nn.Linear weights ----------
        g_iw = []
        d0 = torch.zeros(256, pk["iw0_cols"], device=dev)  # zeroed MANO cond columns get zero gradient
        d0[:, :sp.E] = dW[0][:, :sp.E]
        g_iw.append(d0)
        g_iw += [dW[1], dW[2], dW[3]]
        g_iw.append(dW[4] / math.sqrt(2))
        g_iw += [dW[5], dW[6], dW[7]]
        d8 = torch.cat([dW[8][256:257], dW[8][:256]], 0)
        g_iw.append(d8)
        g_ib = dWb[:8] + [torch.cat([dWb[8][256:257], dWb[8][:256]])]
        g0 = torch.empty(256, sp.rin_dim, device=dev)
        g0[:, rin_perm(sp.rin_dim, dev)] = dR[0][:, :sp.rin_dim]  # back to the reference's column order
        g_rw = [g0, dR[1], dR[2], dR[3], dR[4]]
        return dict(iw=g_iw, ib=g_ib, rw=g_rw, rb=dRb, tfs=dtfs, pose_embed=d_pose, time_code=d_time)
