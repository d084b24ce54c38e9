"""profiles/<round>_pmc_traffic.json from the two --pmc passes of scripts/prof_r03.sh (gpurun_out/prof_r03/{FETCH,WRITE}_SIZE.json).
Unit / correction per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950 FETCH_SIZE
reports half of wide coalesced reads, so read bytes = 2 * FETCH_SIZE * 1024."""
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "gpurun_out", "prof_r06")
rnd = sys.argv[2] if len(sys.argv) > 2 else "r06"
F = json.load(open(os.path.join(src, "FETCH_SIZE.json")))
W = json.load(open(os.path.join(src, "WRITE_SIZE.json")))
# bench.py's kernel families -> rocprof kernel names (rmlp_kernel is both the sampler query <HEAD> and the forward trunk
# <STORE>: the counter CSV keys carry no template arguments, so its traffic is reported under its own entry too)
groups = {"fused_sdf_kernel": ["rmlp_h3_kernel<true,false,0>", "rmlp_kernel<true,false,0>", "fused_sdf_x6p_kernel", "fused_sdf_pipe_kernel",
                               "fused_sdf_kernel"],
          "trunk_r6_kernel": ["rmlp_h3_kernel<false,true,0>", "rmlp_kernel<false,true,0>"],
          # rsweep_kernel<MODE, A2, DIST, ABL, SKIP_OUT>: the foreground nets' sweeps (skip width 217); the background's descending
          # sweep (skip width 172, a third of the points) is reported under its own name so that it does not dilute the average
          "rchain_kernel": ["rsweep_kernel<1,false,1,0,217>", "rsweep_kernel<1,false,1>"],
          "rchain_bg_kernel": ["rsweep_kernel<1,false,1,0,172>"],
          "rchain_a2_kernel": ["rsweep_kernel<1,true,1,0,217>", "rsweep_kernel<1,true,1>"],
          "rchain_dbwd_kernel": ["rsweep_kernel<2,true,1,0,217>", "rsweep_kernel<2,true,1>"],
          "rgemm_kernel": ["rgemm_kernel<0>", "rgemm_kernel<1>", "rgemm_kernel<2>"],
          # round 6: the sweeps and the single-layer GEMM in two fp16 limbs (rsweep_h3_kernel<MODE, A2, DIST, ABL, SKIP_OUT>)
          "rchain_h3_kernel": ["rsweep_h3_kernel<1,false,3,0,217>", "rsweep_h3_kernel<1,false,1,0,217>"],
          "rchain_a2_h3_kernel": ["rsweep_h3_kernel<1,true,3,0,217>", "rsweep_h3_kernel<1,true,1,0,217>"],
          "rchain_dbwd_h3_kernel": ["rsweep_h3_kernel<2,true,1,0,217>", "rsweep_h3_kernel<2,true,3,0,217>"],
          "rchain_bg_h3_kernel": ["rsweep_h3_kernel<1,false,3,0,172>"],
          "rgemm_h3_kernel": ["rgemm_h3_kernel<0,0>", "rgemm_h3_kernel<1,0>", "rgemm_h3_kernel<2,0>", "rgemm_h3_kernel<3,0>", "rgemm_h3_kernel<4,0>",  # <EPI, ABL>
                              "rgemm_h3_kernel<0>", "rgemm_h3_kernel<1>", "rgemm_h3_kernel<2>"],
          "chain_kernel": ["chain_x6_kernel<1,true,16>", "chain_x6_kernel<2,true,3>", "chain_x6_kernel<1,false,16>",
                           "chain_x6_kernel<0,false,3>", "chain_kernel"],
          "sampler_beta_kernel": ["sampler_beta_kernel"], "sampler_sample_kernel": ["sampler_sample_kernel"],
          "composite_fwd_kernel": ["composite_fwd_kernel"], "composite_bwd_kernel": ["composite_bwd_kernel"],
          "gemm_nt_kernel": ["gemm_nt_kernel"], "rnarrow_kernel": ["rnarrow_kernel"],
          "wgrad_h3_kernel": ["wgrad_h3_kernel<true>", "wgrad_h3_kernel<false>", "wgrad_r6_group_kernel<true>"],
          "wgrad_kernel": ["wgrad_r6_group_kernel<false>", "wgrad_r6_kernel<true,0,3>", "wgrad_r6_kernel<false,0,3>", "wgrad_r6_kernel<true>", "wgrad_r6_kernel<false>",
                           "wgrad_r6_kernel", "wgrad_r6_group_kernel", "wgrad_lds_kernel", "wgrad_kernel"]}
notes = {
    "fused_sdf_kernel": "sampler queries: 16 B in (xc row) + 4 B out per point; the 2.8 MiB limb pack stays in L2",
    "chain_kernel": "hold_chain_x6: the first-order backward sweep (DSP + a2: reads 2 + writes 1 KiB per point and layer, 7 layers) and "
                    "the second-order sweep (DBWD: reads 2 + writes 2 KiB, 8 layers) of one node-chunk (P = 1.61 M points)",
    "trunk_r6_kernel": "forward trunk: 16 B in per point, 8 x 1 KiB of h stores out",
    "rchain_kernel": "descending sweep of the normal path: 1 KiB in (chain input) + per layer 1 KiB side in, 1 KiB out, 7 layers",
    "rchain_a2_kernel": "first-order backward sweep: 1 KiB in (chain input) + per layer 2 KiB side in (h, a2), 1 KiB out, 7 layers",
    "rchain_dbwd_kernel": "second-order ascending sweep: 160 B in + per layer 2 KiB side in (h, t), 2 KiB out (vbar, a2), 8 layers",
    "rgemm_kernel": "rendering-net layers / dgrad / lin8 features: 4 (K + 256) B per point (+ 1 KiB mask operand)",
    "rgemm_h3_kernel": "rendering-net layers / dgrad / lin8 features in two fp16 limbs: 4 (K + 256) B per point (+ 1 KiB mask operand) + 8 B of row maxima",
    "rchain_h3_kernel": "descending sweep of the normal path (two fp16 limbs): 1 KiB in (chain input) + per layer 1 KiB side in, 1 KiB out, 7 layers",
    "rchain_a2_h3_kernel": "first-order backward sweep (two fp16 limbs): 1 KiB in + per layer 2 KiB side in (h, a2), 1 KiB out, 7 layers",
    "rchain_dbwd_h3_kernel": "second-order ascending sweep (two fp16 limbs): 160 B in + per layer 2 KiB side in (h, t), 2 KiB out (vbar, a2), 8 layers",
    "gemm_nt_kernel": "per-layer GEMMs (rendering net fwd+bwd, lin8 features, d/d embedding, background): (K + N) * 4 B per "
                      "point (+ N * 4 B per aux operand of the MUL_DSP / DRELU epilogues)",
    "rnarrow_kernel": "the N <= 64 layers (N = 39 / 16 / 48): 1 KiB in per point (A once) + 4 N B out (+ 4 N B in when accumulating)",
    "wgrad_h3_kernel": "dW[256,256] = R^T X over P = 1.61 M points in the two-limb fp16 arithmetic: 2 KiB per point (3.3 GB) of operand "
                       "rows + the scale sample (64 rows per workgroup and operand) + 256 x 256 KiB of partial tiles",
    "wgrad_kernel": "dW[N,K] = R^T X over P = 1.61 M points: (N + K) * 4 B per point = 2 KiB (3.3 GB) for the 256x256 layers; "
                    "split-K partials (<= 256 x 256 KiB) are written here and reduced by wgrad_reduce4_kernel"}
out = {"command": "rocprofv3 --kernel-trace --pmc <C> -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile "
                  "(one pass per counter; scripts/lease_logs/r6_call11.sh), chunk 16384 rays, default precision (f16x3: trunk, sampler queries, backward sweeps, single-layer GEMMs and whole-dW weight gradients in the two-limb fp16 arithmetic; gemm_nt / rnarrow / tile weight gradients / the background sweep f32x6)",
       "correction": "gfx950 FETCH_SIZE reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md HBM section): read bytes = "
                     "2*FETCH_SIZE*1024; WRITE_SIZE taken as KiB", "kernels": {}}
for g, names in groups.items():
    nf = sum(F[n]["launches"] for n in names if n in F)
    sf = sum(F[n]["sum"] for n in names if n in F)
    nw = sum(W[n]["launches"] for n in names if n in W)
    sw = sum(W[n]["sum"] for n in names if n in W)
    if not nf or not nw:
        continue
    out["kernels"][g] = {"pmc_kernel_names": [n for n in names if n in F], "launches_in_pass": nf,
                         "FETCH_SIZE_KB_avg_per_launch": sf / nf, "WRITE_SIZE_KB_avg_per_launch": sw / nw,
                         "hbm_read_bytes_per_launch": 2 * 1024 * sf / nf, "hbm_write_bytes_per_launch": 1024 * sw / nw,
                         "hbm_bytes_per_launch": 2 * 1024 * sf / nf + 1024 * sw / nw, "algorithmic_note": notes.get(g, "")}
dst = os.path.join(root, "profiles", f"{rnd}_pmc_traffic.json")
json.dump(out, open(dst, "w"), indent=1)
print("wrote", dst, {k: round(v["hbm_bytes_per_launch"] / 1e9, 3) for k, v in out["kernels"].items()}, "GB/launch")
