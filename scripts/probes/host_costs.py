"""Host-side cost of the small things a C3 step does ~1 000 times (GPU box; main thread vs the autograd engine's thread)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
dev = "cuda:0"
ws = [torch.zeros(256, 256, device=dev) for _ in range(20)]


def t(name, f, n=300):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    dt = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    print(f"  {name:58s} {dt:8.2f} us")


def suite(tag):
    print(tag)
    flat = torch.zeros(20 * 65536, device=dev)
    shp = ws[0].shape
    t("flat[a:b]", lambda: flat[64:64 + 65536])
    s = flat[64:64 + 65536]
    t("slice.view(shape)", lambda: s.view(shp))
    t("slice.view(256, 256)", lambda: s.view(256, 256))
    t("flat[a:b].view(shape)", lambda: flat[64:64 + 65536].view(shp))
    t("flat.as_strided", lambda: flat.as_strided((256, 256), (256, 1), 64))
    t("flat.split(65536)", lambda: flat.split(65536))
    t("flat.view(20, 256, 256).unbind(0)", lambda: flat.view(20, 256, 256).unbind(0))
    t("torch.zeros(20 * 65536)", lambda: torch.zeros(20 * 65536, device=dev))
    t("torch.empty(20 * 65536)", lambda: torch.empty(20 * 65536, device=dev))
    t("torch.cuda.current_stream().cuda_stream", lambda: torch.cuda.current_stream().cuda_stream)
    t("torch._C._cuda_getCurrentRawStream(0)", lambda: torch._C._cuda_getCurrentRawStream(0))
    h = torch.rand(1280, 3)
    t("host.pin_memory().to(dev, non_blocking)", lambda: h.pin_memory().to(dev, non_blocking=True), 50)
    hp = torch.empty(1280, 3).pin_memory()
    t("pinned.copy_(host); .to(dev, non_blocking)", lambda: (hp.copy_(h), hp.to(dev, non_blocking=True)), 50)
    t("torch.rand(1280, 3, device=dev)", lambda: torch.rand(1280, 3, device=dev), 50)
    t("ws[0].data_ptr()", lambda: ws[0].data_ptr())
    t("ws[0].stride(0)", lambda: ws[0].stride(0))


suite("main thread")


class F(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x * 2

    @staticmethod
    def backward(ctx, g):
        suite("autograd engine thread (inside Function.backward)")
        return g * 2


x = torch.ones(3, device=dev, requires_grad=True)
F.apply(x).sum().backward()


# ---- zeros_like_many as the backward bodies call it (cProfile of a C3 step charged ~25 us to every .view inside it) ----
from hold_amd.field import zeros_like_many  # noqa: E402

W = [torch.zeros(256, 40, device=dev)] + [torch.zeros(256, 256, device=dev) for _ in range(7)] + [torch.zeros(1, 256, device=dev)]
Bs = [torch.zeros(256, device=dev) for _ in range(8)] + [torch.zeros(1, device=dev)]
R = [torch.zeros(256, 272, device=dev)] + [torch.zeros(256, 256, device=dev) for _ in range(3)] + [torch.zeros(3, 256, device=dev)]
Rb = [torch.zeros(256, device=dev) for _ in range(4)] + [torch.zeros(3, device=dev)]


def zl_strided(*lists):
    ts = [t for lst in lists for t in lst]
    sizes = [(t.numel() + 63) // 64 * 64 for t in ts]
    flat = torch.zeros(sum(sizes), device=ts[0].device)
    out, off = [], 0
    for t, n in zip(ts, sizes):
        out.append(flat.as_strided(t.shape, t.stride(), off))
        off += n
    res, i = [], 0
    for lst in lists:
        res.append(out[i:i + len(lst)])
        i += len(lst)
    return res


def suite2(tag):
    print(tag)
    t("zeros_like_many(R, Rb, W, Wb): 28 tensors", lambda: zeros_like_many(R, Rb, W, Bs), 100)
    t("  same through as_strided", lambda: zl_strided(R, Rb, W, Bs), 100)
    t("zeros_like_many(W, Wb): 18 tensors", lambda: zeros_like_many(W, Bs), 100)
    t("28 x torch.zeros_like", lambda: [torch.zeros_like(x) for x in R + Rb + W + Bs], 100)


suite2("main thread")


class G(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x * 2

    @staticmethod
    def backward(ctx, g):
        suite2("autograd engine thread")
        return g * 2


G.apply(x).sum().backward()
