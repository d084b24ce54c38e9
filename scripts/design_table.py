"""print DESIGN.md section 4's kernel table from a bench line (python scripts/design_table.py profiles/r06_bench_final.json)"""
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]; e = r["kernels"]
desc = {
    "wgrad_h3_kernel": ("**`wgrad_h3_kernel`** (`csrc/wgrad_r6.hip`: `wgrad_h3_body`; every whole-dW shape)", "2 N K flop and 4 (N + K) B per point"),
    "fused_sdf_kernel": ("**`rmlp_h3_kernel<HEAD>`** = `hold_fused_sdf_h3` (`csrc/rmlp_h3.hip`) -- sampler queries", "0.94 MFLOP/point; HBM 16 B in + 4 B out per point"),
    "rchain_dbwd_h3_kernel": ("**`rsweep_h3_kernel<DBWD>`** (`csrc/rchain_h3.hip`, round 6) -- second-order ascending sweep", "131 kFLOP and 4 KiB per point and layer, 8 layers"),
    "rgemm_h3_kernel": ("**`rgemm_h3_kernel<EPI>`** = `hold_gemm_h3` (`csrc/rgemm_h3.hip`, round 6) -- rendering net lin0..3, their input gradients, lin8's feature rows", "2 256 K flop, 4 (K + 256) B per point (+ 1 KiB mask, + 8 B row maxima)"),
    "rchain_a2_h3_kernel": ("**`rsweep_h3_kernel<DSP+a2>`** -- first-order backward sweep", "131 kFLOP and 3 KiB per point and layer, 7 layers"),
    "rchain_h3_kernel": ("**`rsweep_h3_kernel<DSP>`** -- descending sweep of the normal path", "131 kFLOP and 2 KiB per point and layer, 7 layers"),
    "trunk_r6_kernel": ("`rmlp_h3_kernel<STORE>` = `hold_trunk_h3` -- training forward trunk", "0.84 MFLOP/point; HBM 16 B in + 8 KiB out"),
    "gemm_nt_kernel": ("`gemm_nt_kernel<EPI,NT,x6>` -- lin8's input gradient (MUL_DSP + a2 + rank-1), the background net (`f32x6`)", "2 N K flop, 4 (K + N) B per point"),
    "wgrad_kernel": ("`wgrad_lds_kernel<x6>` + grouped launches of the per-frame nets (K = 40, the 16 / 48-column tails; `f32x6`)", "as above"),
    "rchain_bg_kernel": ("`rsweep_kernel<DSP, skip 172>` -- the background's backward sweep (`f32x6`)", "131 kFLOP and 2 KiB per point and layer"),
    "rnarrow_kernel": ("`rnarrow_kernel` = `hold_gemm_narrow_x6` -- the N <= 64 layers (`f32x6`)", "2 256 N flop, 4 (256 + N) B per point"),
}
print("| kernel (file) | share of the step | bound (larger floor): fraction of it, both floors | algorithmic work per unit | TF-eq (bench avg) | HBM: PMC bytes per launch / launch time |")
print("|---|---|---|---|---|---|")
for k, v in sorted(e.items(), key=lambda kv: -kv[1].get("time_share", 0)):
    if k not in desc:
        continue
    name, work = desc[k]
    t = v["avg_launch_ms"]; fl = v["floor_ms"]
    tr = v.get("traffic")
    hb = f"{tr / 1e9:.2f} GB / {t:.2f} ms = {tr / t / 1e9:.2f} TB/s" if tr else "--"
    bound = "**HBM**" if v["bound"] == "hbm" else "MFMA"
    print(f"| {name} | {v['time_share'] * 100:.1f} % | {bound}: {v['frac']:.2f} (measured bytes: {v.get('hbm_frac_measured_bytes') or 0:.2f}); floors HBM {fl['hbm']:.2f} ms / MFMA {fl['mfma']:.2f} ms of {t:.2f} ms | {work} | {v['fp32_equivalent_tflops']:.0f} | {hb} |")
ee = r["end_to_end"]
print()
print(f"value {d['value']:.1f} rays/s, {d['ms_per_step']:.1f} ms/step, sigma_I {d['config']['sigma_I']:.2f}; executed {ee['executed_tflops_end_to_end']:.1f} TF-eq (SURVEY-credited {d['config']['algorithmic_tflops_end_to_end']:.1f}); issued {ee['mfma_tflops_issued']:.0f} TFLOP/s = {ee['frac_of_bf16_mfma_peak_issued']:.3f} of 2.5 PF; "
      f"time in MFMA kernels {ee['time_in_mfma_kernels']:.3f}; HBM {ee['hbm_bytes_per_step'] / 1e12:.2f} TB/step = {ee['hbm_tb_s']:.2f} TB/s; floors HBM {ee['step_floor_s']['hbm']:.2f} s, MFMA {ee['step_floor_s']['mfma']:.2f} s; "
      f"top-level: {r['kernel'][:40]} share {r['time_share']:.3f} bound {r['bound']} frac {r['frac']:.3f} measured {r.get('hbm_frac_measured_bytes')}; c_abi calls/step {d['config']['c_abi_calls_per_step']:.0f}; fallbacks/step {d['config'].get('f16x3_launches_recomputed_in_f32x6_per_step')}")
if "cpu_baseline" in d:
    c = d["cpu_baseline"]; print("cpu_baseline", c["value"], c["network_only_rays_per_s"], c["step_s"]["median"], c["cores"])
