"""CPU study for DESIGN.md section 7: would bf16 limb-split MFMA (fp32 operand = sum of 3 bf16 limbs, products in
fp32 accumulators) keep the trunk inside the 1e-4 parity bar?  Emulated exactly with torch CPU: a bf16 x bf16 product
is exact in fp32, the accumulation is fp32 (order differs from the MFMA's, as it does between any two fp32 GEMMs).

  x3: a1b1 + a1b2 + a2b1                      (3 MFMAs, dropped terms <= 2^-16 relative)
  x6: + a2b2 + a1b3 + a3b1                    (6 MFMAs, dropped terms <= 2^-24 relative)

Reports, against an fp64 evaluation of the same synthetic ImplicitNet: max relative error of sdf, of the feature
vector, of d sdf / d x (the normal path) and of d(sum sdf)/dW for plain fp32, x3 and x6."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import math
import torch
from hold_amd import synthetic as syn

torch.manual_seed(0)
torch.set_num_threads(16)


def limbs(x, n):
    out, r = [], x.clone()
    for _ in range(n):
        l = r.to(torch.bfloat16).to(torch.float32)
        out.append(l)
        r = r - l
    return out


def mm_split(a, w, mode):
    """a [P,K] @ w[N,K]^T with limb-split operands, fp32 accumulate"""
    if mode == "fp32":
        return a @ w.t()
    n = 2 if mode == "x3" else 3
    A, W = limbs(a, n), limbs(w, n)
    pairs = [(0, 0), (0, 1), (1, 0)] if mode == "x3" else [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)]
    acc = torch.zeros(a.shape[0], w.shape[0])
    for i, j in reversed(pairs):  # small terms first
        acc = acc + A[i] @ W[j].t()
    return acc


class SplitLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, w, mode):
        ctx.save_for_backward(a, w)
        ctx.mode = mode
        return mm_split(a, w, mode)

    @staticmethod
    def backward(ctx, g):
        a, w = ctx.saved_tensors
        return mm_split(g, w.t().contiguous(), ctx.mode), mm_split(g.t().contiguous(), a.t().contiguous(), ctx.mode), None


def embed(x, L=6):
    out = [x]
    for k in range(L):
        out += [torch.sin(x * 2.0 ** k), torch.cos(x * 2.0 ** k)]
    return torch.cat(out, -1)


def trunk(x, W, b, mode, dtype):
    e = embed(x)
    h = e
    for l in range(9):
        if l == 4:
            h = torch.cat([h, e], -1) / math.sqrt(2)
        wl = W[l][:, :h.shape[1]]
        y = (h @ wl.t() if dtype == torch.float64 else SplitLinear.apply(h, wl.contiguous(), mode)) + b[l]
        h = torch.nn.functional.softplus(y, beta=100) if l < 8 else y
    return h


def main():
    sc = syn.make_scene(2)
    sd = syn.make_state_dict(sc, perturb=0.05)
    pre = "nodes.object.implicit_network."
    W, b = [], []
    for l in range(9):
        v, g = torch.as_tensor(sd[pre + f"lin{l}.weight_v"]), torch.as_tensor(sd[pre + f"lin{l}.weight_g"])
        W.append(v * (g / v.norm(dim=1, keepdim=True)))
        b.append(torch.as_tensor(sd[pre + f"lin{l}.bias"]))
    P = 4096
    x0 = (torch.rand(P, 3) * 1.2 - 0.6)
    res = {}
    for mode, dt in (("fp64", torch.float64), ("fp32", torch.float32), ("x3", torch.float32), ("x6", torch.float32)):
        Wd = [w.to(dt).clone().requires_grad_(True) for w in W]
        bd = [t.to(dt) for t in b]
        x = x0.to(dt).clone().requires_grad_(True)
        out = trunk(x, Wd, bd, mode, dt)
        sdf = out[:, 0]
        (gx,) = torch.autograd.grad(sdf.sum(), x, create_graph=True)
        n = gx / gx.norm(dim=1, keepdim=True)
        loss = (n * torch.tensor([0.3, -0.5, 0.8], dtype=dt)).sum() + sdf.sum()
        gW = torch.autograd.grad(loss, Wd)
        res[mode] = dict(sdf=sdf.detach().double(), feat=out[:, 1:].detach().double(), gx=gx.detach().double(),
                         gW=[g.detach().double() for g in gW])
    ref = res["fp64"]
    rel = lambda a, r: float((a - r).abs().max() / r.abs().max())
    print(f"{'mode':6s} {'sdf':>10s} {'feat':>10s} {'dsdf/dx':>10s} {'dL/dW (max over layers, 2nd order incl.)':>40s}")
    for mode in ("fp32", "x3", "x6"):
        r = res[mode]
        print(f"{mode:6s} {rel(r['sdf'], ref['sdf']):10.2e} {rel(r['feat'], ref['feat']):10.2e} "
              f"{rel(r['gx'], ref['gx']):10.2e} {max(rel(a, c) for a, c in zip(r['gW'], ref['gW'])):40.2e}")


if __name__ == "__main__":
    main()
