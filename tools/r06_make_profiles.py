"""Builds profiles/r06_pmc_mfma.json, profiles/r06_pmc_traffic.json and profiles/r06_bench_kernel_stats.csv from what
tools/r06_profile.sh left under gpurun_out/r06prof (pmc_table.json = per-kernel counter averages, kernel_stats.csv = the
rocprofv3 --kernel-trace --stats summary of the same bench command).
usage: python tools/r06_make_profiles.py [gpurun_out/r06prof]"""
import csv, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r06prof")
tab = json.load(open(os.path.join(src, "pmc_table.json")))
stats = {r["Name"]: r for r in csv.DictReader(open(os.path.join(src, "kernel_stats.csv")))}

M = 8 * 4 * 1024                        # tokens of one 8-tile batch (4 sub-tiles of 1024 tokens per tile)
C, HID = 1024, 4096
gemm_flops = lambda n, k: 2.0 * M * n * k
gemm_bytes = lambda n, k, extra=0: 2.0 * (M * k + n * k + M * n) + extra
# kernel-name prefix -> (what, algorithmic flops per launch, algorithmic bytes per launch)
KERNELS = [
    # round 5: the MLP runs in two row parts of 16 384 tokens (cpx_net_mlp_parts): an mlp.lin1 / mlp.lin2 LAUNCH covers M / 2 rows
    ("void k_gemm4w<1, 0, false>", "mlp.lin1 (fc1, GELU + folded LayerNorm): one wave per SIMD, persistent; 16 384 rows per launch (two launches per layer)", gemm_flops(HID, C) / 2, gemm_bytes(HID, C) / 2 + HID * C),
    ("void k_gemm256p<1, false, 33>", "mlp.lin1 (fc1, GELU + folded LayerNorm, direct-store epilogue), 8-wave kernel", gemm_flops(HID, C), gemm_bytes(HID, C)),
    ("void k_gemm256p<6, false, 1>", "attn.qkv (+ V^T epilogue, folded LayerNorm), persistent with the balanced tile list", gemm_flops(3 * C, C), gemm_bytes(3 * C, C)),
    ("void k_gemm256<6, false, 1>", "attn.qkv (+ V^T epilogue, folded LayerNorm), one workgroup per tile", gemm_flops(3 * C, C), gemm_bytes(3 * C, C)),
    ("void k_gemm256p<2, false, 66>", "attn.proj (32 768 rows) and mlp.lin2 (two launches of 16 384 rows): residual + row statistics, balanced fragment-read schedule; average over the three launches per layer",
     (gemm_flops(C, C) + gemm_flops(C, HID)) / 3, (gemm_bytes(C, C, 2 * M * C) + gemm_bytes(C, HID, 2 * M * C)) / 3),
    ("void k_attention4p<false", "rel-pos flash attention (4-wave, LDS-DMA ring; production variant 2)",
     4.0 * 32 * 16 * 1024 * 1024 * 64 + 4.0 * 32 * 16 * 1024 * 32 * 64, 2.0 * 4 * M * C),      # algorithmic: 4 T^2 hd heads + 4 heads T sqrt(T) hd
    ("void k_attention<false, false, false>", "rel-pos flash attention (variant 0)",
     4.0 * 32 * 16 * 1024 * 1024 * 64 + 2 * 2.0 * 32 * 16 * 1024 * 64 * 64, 2.0 * 4 * M * C),
]


def find(d, prefix):
    for k in d:
        if k.startswith(prefix):
            return k
    return None


out = {}
for prefix, what, flops, abytes in KERNELS:
    k = find(tab, prefix)
    if k is None:
        continue
    v = tab[k]
    e = {"what": what, "launches_profiled": int(v["launches"])}
    for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA", "SQ_INSTS_VALU_MFMA_MOPS_BF16"):
        e[c] = v.get(c)
    gui = v["GRBM_GUI_ACTIVE"]
    e["GRBM_GUI_ACTIVE_sum_over_xcds"] = gui
    e["cycles_per_xcd"] = gui / 8
    e["MfmaUtil_percent"] = round(100.0 * v["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8 * 1024), 2)
    e["mfma_flops_counted"] = v["SQ_INSTS_VALU_MFMA_MOPS_BF16"] * 512
    e["algorithmic_flops"] = flops
    wc = v["SQ_WAVE_CYCLES"]
    e["SQ_WAVE_CYCLES"] = wc
    e["SQ_WAIT_ANY_frac"] = round(v["SQ_WAIT_ANY"] / wc, 3)
    e["SQ_WAIT_INST_ANY_frac"] = round(v["SQ_WAIT_INST_ANY"] / wc, 3)
    e["SQ_ACTIVE_INST_ANY_frac"] = round(v["SQ_ACTIVE_INST_ANY"] / wc, 3)
    e["SQ_ACTIVE_INST_VALU_frac_of_active"] = round(v["SQ_ACTIVE_INST_VALU"] / max(v["SQ_ACTIVE_INST_ANY"], 1), 3)
    e["SQ_VALU_MFMA_COEXEC_CYCLES"] = v["SQ_VALU_MFMA_COEXEC_CYCLES"]
    e["SQ_LDS_BANK_CONFLICT_frac_of_LDS_active"] = round(v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_LDS_IDX_ACTIVE"], 1), 3)
    e["FETCH_bytes_per_launch_corrected_x2"] = v["FETCH_SIZE"] * 1024 * 2
    e["WRITE_bytes_per_launch"] = v["WRITE_SIZE"] * 1024
    s = stats.get(find(stats, prefix) or "")
    if s:
        us = float(s["AverageNs"]) / 1e3
        e["avg_duration_us_kernel_trace"] = round(us, 1)
        e["effective_clock_GHz"] = round(gui / 8 / us / 1e3, 3)
        e["achieved_TFLOPs_kernel_trace"] = round(flops / us / 1e6, 1)
        e["frac_of_2.5PF"] = round(flops / us / 1e6 / 2500.0, 4)
    e["algorithmic_bytes_per_launch"] = abytes
    out[k] = e

note = ("rocprofv3 --pmc passes of `python3 bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline --no-stages --no-side-lines --no-live-traffic` "
        "(tools/r06_profile.sh; one pass per counter group, kernel-trace only; averages per launch over all launches of the "
        "kernel; assembled by tools/r06_make_profiles.py). MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8 XCDs) "
        "x 1024 SIMDs): the fraction of SIMD cycles in which the matrix pipe executes (rocprofv3's own MfmaUtil formula; "
        "GRBM_GUI_ACTIVE is the sum over the 8 XCDs, MI355X_MICROARCH 'DVFS give-back'). effective_clock_GHz = "
        "(GRBM_GUI_ACTIVE / 8) / average kernel duration from the --kernel-trace --stats pass of the same command "
        "(profiles/r06_bench_kernel_stats.csv). frac_of_2.5PF = algorithmic flops / that duration / 2.5 PFLOP/s: what "
        "bench.py's roofline.frac measures with HIP events. FETCH_SIZE is doubled (gfx950 counts 64 B per 128-B request "
        "for wide coalesced / LDS-DMA reads).")
json.dump({"note": note, "kernels": out}, open(os.path.join(ROOT, "profiles", "r06_pmc_mfma.json"), "w"), indent=1)

dom = find(tab, KERNELS[0][0])
fetch = tab[dom]["FETCH_SIZE"] * 1024 * 2
write = tab[dom]["WRITE_SIZE"] * 1024
traffic = {
    "note": ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, tools/r06_profile.sh) of `bench.py --steps 4 "
             "--warmup 2`; KB per launch averaged over all launches. gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE "
             "counts 64 B per 128-B request for wide coalesced / LDS-DMA reads -> doubled. FETCH_SIZE is the L2's fabric-side "
             "traffic and includes Infinity-Cache hits (the W panels, 8.4 MB, and most of the 67 MB activation panel stay "
             "resident in the 256 MiB MALL), so it bounds HBM traffic from above."),
    "dominant_kernel": "k_gemm4w<GELU + folded LayerNorm> (mlp.lin1, 16 384 rows per launch, N=4096 K=1024): one wave per SIMD, persistent",
    "fetch_bytes_per_launch_corrected": fetch,
    "write_bytes_per_launch": write,
    "traffic_bytes_per_launch": fetch + write,
    "algorithmic_bytes_per_launch": int(KERNELS[0][3]),
    "kernels": {k: {"FETCH_SIZE_KB_avg_per_launch": v["FETCH_SIZE"], "WRITE_SIZE_KB_avg_per_launch": v["WRITE_SIZE"],
                    "launches": int(v["launches"])} for k, v in tab.items()},
}
json.dump(traffic, open(os.path.join(ROOT, "profiles", "r06_pmc_traffic.json"), "w"), indent=1)
shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(ROOT, "profiles", "r06_bench_kernel_stats.csv"))
for k, e in out.items():
    print(f"{k[:44]:46s} MfmaUtil {e['MfmaUtil_percent']:5.1f}%  {e.get('avg_duration_us_kernel_trace', 0):7.1f} us  "
          f"frac {e.get('frac_of_2.5PF', 0):.3f}  clock {e.get('effective_clock_GHz', 0):.3f} GHz")

# post-processing chain alone: sum of kernel time per 8-tile batch (tools/run_post.py runs 12 iterations)
try:
    post = os.path.join(ROOT, "gpurun_out", "r06post", "kernel_stats.csv")
    rows = list(csv.DictReader(open(post)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows) / 12 / 1e3
    calls = sum(int(r["Calls"]) for r in rows) / 12
    json.dump({"sum_of_kernel_time_us_per_8_tile_batch": round(tot, 1), "kernel_launches_per_batch": calls,
               "source": "rocprofv3 --kernel-trace --stats -- python3 tools/run_post.py (12 iterations of cpx_compute_masks on one 8-tile batch; "
                         "tools/r06_profile.sh); per-kernel table: profiles/r06_post_kernel_stats.csv"},
              open(os.path.join(ROOT, "profiles", "r06_post_kernel_sum.json"), "w"), indent=1)
    shutil.copy(post, os.path.join(ROOT, "profiles", "r06_post_kernel_stats.csv"))
    print(f"post-processing: {tot:.1f} us of kernel time per 8-tile batch, {calls:.1f} launches")
except Exception as e:
    print("no post-processing stats:", e)
