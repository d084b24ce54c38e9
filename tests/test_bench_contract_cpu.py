"""The bench line's contract, checked on the committed output of the default command (profiles/r06_bench_final.json is what
`python bench.py` printed on an MI355X, scripts/lease_logs/r6_call43.sh) and on bench.py's command line -- no GPU needed."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    return json.load(open(os.path.join(ROOT, "profiles", "r06_bench_final.json")))


def test_bench_line_carries_every_contract_field():
    d = _line()
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    # the metric is BASELINE.json's (typography aside: the driver compares the words)
    norm = lambda s: s.replace("×", "x").split(";")[0].strip()
    assert norm(d["metric"]) == norm(base["metric"])
    for k, t in (("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                 ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict)):
        assert isinstance(d[k], t), (k, type(d[k]))
    assert d["unit"] == "rays/s" and d["higher_is_better"] is True and d["scaling"] in ("weak", "strong")
    assert d["data"] == "synthetic" and d["n_gpus"] == 1 and "vs_baseline" in d and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = whole-job rays / the timed region: one 512 x 512 frame per step
    assert abs(d["value"] - 512 * 512 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


def test_bench_line_roofline_and_cpu_baseline_objects():
    d = _line()
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["peak"] in (8000.0, 2500.0)  # HBM GB/s, dense bf16 MFMA TFLOP/s (MI355X_MICROARCH.md)
    assert 0.0 < r["frac"] < 1.0 and r["traffic"] and r["traffic"] > 0.5 * r["algorithmic_bytes_per_launch_avg"]
    # both floors for every MFMA family, and the families account for the time they claim
    fam = {k: v for k, v in r["kernels"].items() if "mfma_frac" in v}
    assert len(fam) >= 8
    for k, v in fam.items():
        assert v["bound"] == ("hbm" if v["floor_ms"]["hbm"] > v["floor_ms"]["mfma"] else "mfma"), k
        assert v["floor_ms"][v["bound"]] <= v["avg_launch_ms"] * 1.001, k  # no family beats its own roofline
    share = sum(v["time_share"] for v in fam.values())
    assert abs(share - r["end_to_end"]["time_in_mfma_kernels"]) < 1e-6 and 0.5 < share < 1.0
    c = d["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(c) and c["kind"] in ("port", "reference")
    assert c["unit"] == d["unit"] and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str)
    # round 5 (advisor): median of >= 3 repeats with the spread printed, the loss-target geometry (kaolin on a GPU
    # in the reference) apart, no GPU / CPU ratio in the line, executed FLOP next to the SURVEY-credited figure
    assert c["repeats"] >= 3 and len(c["step_s"]["all"]) >= 3 and c["step_s"]["min"] <= c["step_s"]["median"] <= c["step_s"]["max"]
    assert c["like_for_like"] is False and c["network_only_rays_per_s"] > c["value"]
    assert "speedup_vs_cpu_baseline" not in d["config"]
    assert r["end_to_end"]["executed_tflops_end_to_end"] < d["config"]["algorithmic_tflops_end_to_end"]
    # the arithmetic of every family is stated, and priced by the matrix instructions it issues
    assert "f16x3" in d["dtype"] and d["config"]["precision"] == "f16x3"
    for k, v in fam.items():
        n = v["limb_products_issued_per_product"]
        assert n in (1.0, 3.0, 6.0) and abs(v["mfma_achieved"] - n * v["fp32_equivalent_tflops"]) < 1e-6 * v["mfma_achieved"], k
    assert fam["fused_sdf_kernel"]["limb_products_issued_per_product"] == 3.0 and fam["wgrad_h3_kernel"]["arithmetic"] == "f16x3"
    # round 6: the sweeps and the single-layer GEMMs are f16x3 families too; the last f32x6 ones are named as such
    for k in ("rchain_h3_kernel", "rchain_a2_h3_kernel", "rchain_dbwd_h3_kernel", "rgemm_h3_kernel"):
        assert fam[k]["limb_products_issued_per_product"] == 3.0 and fam[k]["arithmetic"] == "f16x3", k
    assert fam["gemm_nt_kernel"]["limb_products_issued_per_product"] == 6.0 and fam["rnarrow_kernel"]["arithmetic"] == "f32x6"
    # the top-level object is the dominant kernel TEMPLATE: the sweeps' instantiations taken together
    tmpl = [k for k in fam if k.startswith("rchain_") and k.endswith("_h3_kernel")]
    assert "rsweep_h3_kernel" in r["kernel"] and abs(r["time_share"] - sum(fam[k]["time_share"] for k in tmpl)) < 1e-9
    assert r["launches"] == sum(fam[k]["launches"] for k in tmpl)
    # the overflow guard's fallback count is in the line, and the reference's own CPU timing is quoted as what it is
    assert d["config"]["f16x3_launches_recomputed_in_f32x6_per_step"] == 0.0
    assert c["reference_timed_in_build_container"]["rays_per_s"] > 0 and "NOT measured on this box" in c["reference_timed_in_build_container"]["note"]


def test_bench_command_line_defaults():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr[-500:]
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout
    src = open(os.path.join(ROOT, "bench.py")).read()
    # with no flags: one GPU (the driver's N = 1 run), and nothing under oracle/ outside the cpu_baseline leg
    assert '"--gpus", type=int, default=1' in src
