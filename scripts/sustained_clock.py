"""measure the clock and the power the chip HOLDS under the three kernels whose roofline fraction DESIGN
section 4 explains with "the sustained clock", instead of inferring it: for wgrad_r6_kernel, rmlp_kernel<HEAD> (and its f16x3
twin) and rsweep_kernel<DBWD>, launch the kernel back to back for ~`SECONDS` while a sampling thread reads, every 25 ms,
the amdgpu hwmon / sysfs files of the device (current sclk: hwmon freq1_input or the starred level of pp_dpm_sclk; socket
power: power1_average / power1_input; memory clock; junction temperature), plus one `rocm-smi` call per kernel as a
cross-check.  Writes gpurun_out/r05_sustained_clock.json; scripts/lease_logs/r5_call*.sh copies it to profiles/.

  python scripts/sustained_clock.py [seconds per kernel, default 8]
"""
import glob, json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hold_amd
from hold_amd import field as F, gemm as G, kernels as K

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
dev = "cuda:0"


def _read(p):
    try:
        return open(p).read().strip()
    except OSError:
        return None


def find_device_files():
    """sysfs directory + hwmon directory of THE GPU torch sees as cuda:0 -- the box's /sys lists every card of the host (GPU
    call 1 read card0: another, idle, device), so the card is matched by the PCI address of the torch device"""
    pr = torch.cuda.get_device_properties(0)
    want = None
    try:
        want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
    except AttributeError:
        pass
    out = {"pci": want}
    for card in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        if _read(os.path.join(card, "vendor")) != "0x1002":
            continue
        real = os.path.realpath(card)
        if want and want not in real:
            continue
        hw = sorted(glob.glob(os.path.join(card, "hwmon", "hwmon*")))
        out.update(card=card, real=real, hwmon=hw[0] if hw else None)
        break
    return out


DEVF = find_device_files()


def sample():
    s = {"t": time.time()}
    card, hw = DEVF.get("card"), DEVF.get("hwmon")
    if card:
        for name in ("pp_dpm_sclk", "pp_dpm_mclk"):
            txt = _read(os.path.join(card, name))
            if txt:
                for ln in txt.splitlines():
                    if ln.rstrip().endswith("*"):
                        try:
                            s[name] = float(ln.split(":")[1].strip().rstrip("*").strip().lower().replace("mhz", ""))
                        except (ValueError, IndexError):
                            s[name + "_raw"] = ln
        b = _read(os.path.join(card, "gpu_busy_percent"))
        if b is not None:
            s["busy"] = float(b)
    if hw:
        for f, key, scale in (("freq1_input", "sclk_hz", 1.0), ("freq2_input", "mclk_hz", 1.0), ("power1_average", "power_uw", 1.0),
                              ("power1_input", "power_in_uw", 1.0), ("temp2_input", "tj_mc", 1.0), ("temp1_input", "te_mc", 1.0)):
            v = _read(os.path.join(hw, f))
            if v is not None:
                try:
                    s[key] = float(v) * scale
                except ValueError:
                    pass
    return s


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.stop, self.rows = False, []

    def run(self):
        while not self.stop:
            self.rows.append(sample())
            time.sleep(0.025)


def smi():
    for cmd in (["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], ["rocm-smi", "-c", "-P"]):
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=20)
            if r.returncode == 0 and r.stdout.strip():
                return {"cmd": " ".join(cmd), "out": r.stdout.strip()[:4000]}
        except Exception as e:  # noqa
            last = repr(e)
    return {"cmd": None, "out": None}


def stats(rows, key, scale=1.0):
    v = [r[key] * scale for r in rows if key in r]
    if not v:
        return None
    v.sort()
    return dict(n=len(v), mean=sum(v) / len(v), median=v[len(v) // 2], min=v[0], max=v[-1])


def run_loop(name, fn, flops, issued_per_alg):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    one = e0.elapsed_time(e1)
    n = max(10, int(SECONDS * 1000 / one))
    smp = Sampler(); smp.start()
    time.sleep(0.3)
    idle_n = len(smp.rows)
    e0.record()
    res = {"smi": [], "stop": False}

    def poll_smi():  # rocm-smi sees the right device (ROCR_VISIBLE_DEVICES); one reading takes a few hundred ms
        time.sleep(SECONDS * 0.25)
        while not res["stop"]:
            r = smi()
            r["t"] = time.time()
            res["smi"].append(r)
            time.sleep(0.2)

    t_smi = threading.Thread(target=poll_smi, daemon=True)
    t_smi.start()
    for i in range(n):
        fn()
        if i % 64 == 63:
            torch.cuda.current_stream().synchronize()  # keep the host within 64 launches of the device: the loop's wall time = device time
    e1.record(); torch.cuda.synchronize()
    res["stop"] = True
    ms = e0.elapsed_time(e1) / n
    smp.stop = True; smp.join()
    t_smi.join(timeout=25)
    # samples of the steady state: skip the first 25 % of the loop (ramp) and what was read before / after it
    rows = smp.rows[idle_n:]
    rows = rows[len(rows) // 4:]
    out = dict(kernel=name, launches=n, ms_per_launch=ms, tf_eq=flops / ms / 1e9, issued_tflops=issued_per_alg * flops / ms / 1e9,
               sclk_mhz_hwmon=stats(rows, "sclk_hz", 1e-6), sclk_mhz_dpm=stats(rows, "pp_dpm_sclk"), mclk_mhz_dpm=stats(rows, "pp_dpm_mclk"),
               mclk_mhz_hwmon=stats(rows, "mclk_hz", 1e-6),
               power_w=stats(rows, "power_uw", 1e-6) or stats(rows, "power_in_uw", 1e-6), tj_c=stats(rows, "tj_mc", 1e-3),
               busy=stats(rows, "busy"), idle_before=smp.rows[:idle_n][-1] if idle_n else None, rocm_smi=res["smi"])
    # rocm-smi readings taken INSIDE the loop (after its first quarter)
    sm = []
    for r in res["smi"]:
        try:
            o = next(iter(json.loads(r["out"]).values()))
            sm.append({"sclk_smi": float(o["sclk clock speed:"].strip("()").lower().replace("mhz", "")),
                       "power_smi": float(o["Current Socket Graphics Package Power (W)"]),
                       "tj_smi": float(o["Temperature (Sensor junction) (C)"])})
        except Exception:  # noqa
            pass
    out["rocm_smi_sclk_mhz"], out["rocm_smi_power_w"], out["rocm_smi_tj_c"] = stats(sm, "sclk_smi"), stats(sm, "power_smi"), stats(sm, "tj_smi")
    clk = (out["rocm_smi_sclk_mhz"] or out["sclk_mhz_hwmon"] or out["sclk_mhz_dpm"] or {}).get("median")
    if clk:
        # MFMA peak at the measured clock: 256 CUs x 4 SIMDs x 1024 FLOP per cycle (32x32x16 in 8 passes of 4 cycles) -- bf16 / f16 dense
        out["mfma_peak_tflops_at_sustained_clock"] = 256 * 4 * 1024 * clk * 1e6 / 1e12
        out["mfma_frac_at_sustained_clock"] = out["issued_tflops"] / out["mfma_peak_tflops_at_sustained_clock"]
        out["mfma_frac_at_2p4_ghz_headline"] = out["issued_tflops"] / 2500.0
    print(json.dumps({k: (v if not isinstance(v, dict) or "median" not in v else {"median": v["median"], "min": v["min"], "max": v["max"], "n": v["n"]})
                      for k, v in out.items() if k not in ("rocm_smi", "idle_before")}), flush=True)
    return out


def main():
    torch.manual_seed(0)
    P = 16384 * 98
    g = torch.Generator().manual_seed(0)
    W = [torch.randn(256, 40, generator=g).to(dev) / 6] + [torch.randn(256, 256, generator=g).to(dev) / 16 for _ in range(7)]
    bias = torch.randn(8, 256, generator=g).to(dev) * 0.05
    w8 = (torch.randn(256, generator=g) / 16).to(dev)
    b8 = torch.full((1,), 0.25, device=dev)
    S = torch.stack(W[1:])
    res = {"device_files": DEVF, "seconds_per_kernel": SECONDS, "idle": sample(), "kernels": []}
    # 1. sampler query, bf16 three-limb (6 products) and f16 two-limb (3 products)
    PQ = 128 * 16384
    xc = torch.zeros(PQ, 4, device=dev); xc[:, :3] = torch.rand(PQ, 3, device=dev) * 1.6 - 0.8
    out = torch.empty(PQ, 1, device=dev)
    fl = 2.0 * PQ * (40 * 256 + 6 * 65536 + 217 * 256 + 256)
    pk6 = F.pack_r6(W[0], S)
    res["kernels"].append(run_loop("rmlp_kernel<HEAD> (hold_fused_sdf_r6)", lambda: K.fused_sdf_r6(xc, PQ, pk6, bias, w8, b8, None, out), fl, 6))
    pk3, sw = F.pack_h3(W[0], S)
    bs, c3 = (bias * (sw * F.H3_ACT_SCALE).view(8, 1)).contiguous(), (1.0 / sw).contiguous()
    res["kernels"].append(run_loop("rmlp_h3_kernel<HEAD> (hold_fused_sdf_h3)", lambda: K.fused_sdf_h3(xc, PQ, pk3, bs, c3, w8, b8, None, out), fl, 3))
    del xc, out
    # 2. whole-dW weight gradient
    R = torch.randn(P, 256, device=dev); X = torch.randn(P, 256, device=dev)
    dW = torch.zeros(256, 256, device=dev); db = torch.zeros(256, device=dev)
    res["kernels"].append(run_loop("wgrad_r6_kernel (hold_wgrad_x6, N = K = 256)", lambda: G.wgrad(R, X, dW, db), 2.0 * P * 65536, 6))
    del R, X
    # 3. second-order ascending sweep
    bufs = lambda n: [torch.empty(P, 256, device=dev) for _ in range(n)]
    h, t, a2, o1 = bufs(8), bufs(8), bufs(8), bufs(8)
    for x in h: x.uniform_(0, 0.05)
    for x in t: x.normal_()
    x0 = torch.randn(P, 40, device=dev)
    pack = lambda mats: torch.cat([m.reshape(8, 32, m.shape[1] // 8, 2, 4).permute(2, 0, 3, 1, 4).reshape(-1) for m in mats]).contiguous()
    wf, xf = pack(W), F.pack_x6(W, 48)
    f8 = 2.0 * P * 256 * (40 + 7 * 256)
    res["kernels"].append(run_loop("rsweep_kernel<DBWD> (hold_chain_r6)",
                                   lambda: K.chain(K.CHAIN_DBWD, P, x0, wf, 8, 5, skip_layer=3, side=x0, aux1=h, aux2=t, out=o1, out2=a2,
                                                   wpack_x6=xf, wpack_r6=pk6), f8, 6))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/r05_sustained_clock.json", "w"), indent=1)


if __name__ == "__main__":
    main()
