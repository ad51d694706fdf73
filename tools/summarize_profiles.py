"""Condense the output of tools/collect_profiles.sh into the files committed under profiles/<round>/:
kernel_stats.csv (rocprofv3 --stats summary), pmc_summary.json (per-kernel PMC means + derived figures),
bench_default.json, and profiles/pmc_traffic.json (the roofline.traffic figure bench.py reports).

    python tools/summarize_profiles.py gpurun_out/<dir> profiles/<round_dir>

FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled per the gfx950 correction in
/opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROUND = 4
PATHS_COUNT, MONTHS_COUNT = 1_000_000, 833            # bench.py workload (configs[1])
PATHS_FULL, T_FULL, RY_FULL = 10_000_000, 48, 40       # bench.py hbm_kernels block (configs[2] shape)


import re
HEADLINE = re.compile(r"path_kernel<0, 0, 3, false, false, [03], false")


def label(kernel_name: str):
    k = kernel_name
    if HEADLINE.search(k):      # the headline variant (Philox, unsplit; PHASE 3 = time-sliced blocks at the bench's 10^6 paths, PHASE 0 = plain)
        return "K1 path_kernel<0,...> (count-only, 1e6 paths x 833 months)"
    if "path_kernel<0, 1," in k:
        return "K1 path_kernel<0,1,...> (count-only, NumPy stream, 1e6 paths x 833 months)"
    if "path_kernel<2" in k:
        return "K1 path_kernel<2,...> (full output, 1e7 paths x 555 months)"
    for tag, name in (("rq_slab_kernel<16>", "K3 rq_slab_kernel<16> (the one pass over the slab)"),
                      ("rq_count_kernel<16, true>", "K3 rq_count_kernel<16,true> (slab pass of rows whose bounds do not fit the bucket table)"),
                      ("rq_count_kernel<16, false>", "K3 rq_count_kernel<16,false> (sample pass)"),
                      ("rq_tiny", "K3 rq_tiny_kernel"), ("rq_refine", "K3 rq_refine_kernel"), ("rq_collect", "K3 rq_collect_kernel"),
                      ("rq_select", "K3 rq_select_kernel"), ("rq_hist", "K3 rq_hist_kernel"),
                      ("rq_cand", "K3 rq_cand_hist_kernel"), ("rq_scan", "K3 rq_scan_kernel")):
        if tag in k:
            return name
    if "hist_kernel" in k or "minmax" in k:
        return "K2 hist/minmax"
    return None


def main(src: str, dst: str) -> None:
    os.makedirs(dst, exist_ok=True)
    # (the newest one: a re-used output directory may still hold an older collection's files)
    stats = max(glob.glob(os.path.join(src, "kt", "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    shutil.copy(stats, os.path.join(dst, "kernel_stats.csv"))
    shutil.copy(os.path.join(src, "bench_default.json"), os.path.join(dst, "bench_default.json"))
    per_kernel = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in ("pmc_sq", "pmc_mix", "pmc_fetch", "pmc_write"):
        f = max(glob.glob(os.path.join(src, d, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
        for r in csv.DictReader(open(f)):
            short = label(r["Kernel_Name"])
            if short:
                per_kernel[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    summ = {k: {c: {"mean_per_launch": sum(v) / len(v), "launches": len(v)} for c, v in cs.items()}
            for k, cs in per_kernel.items()}
    k0 = next(v for k, v in summ.items() if k.startswith("K1 path_kernel<0"))
    k2 = next(v for k, v in summ.items() if k.startswith("K1 path_kernel<2"))
    kb = summ["K3 rq_slab_kernel<16> (the one pass over the slab)"]
    m = lambda k, c: k[c]["mean_per_launch"]
    wave_months = (PATHS_COUNT / 64) * MONTHS_COUNT
    stat_rows = list(csv.DictReader(open(stats)))
    # (the kernel with the most calls among the headline variants: the bench's count-only step)
    head_rows = sorted((r for r in stat_rows if HEADLINE.search(r["Name"])), key=lambda r: -int(r["Calls"]))
    k1_mean_all_ms = float(head_rows[0]["AverageNs"]) / 1e6
    k1_name = head_rows[0]["Name"]
    # The figure to compare with bench.py's HIP-event `kernel_ms`: the MEDIAN over the timed launches of the traced run
    # (tools/kernel_trace_medians.py, computed on the box from the kernel-trace CSV); rocprof's --stats mean includes the warm-up
    # launches (round 3: 7.21 ms mean over 12 launches, max 8.39, against 6.8-7.0 in every bench run).
    med = json.load(open(os.path.join(src, "kt_medians.json")))["kernels"]
    k1_med = med[k1_name] if k1_name in med else next(v for k, v in med.items() if HEADLINE.search(k))
    k1_ms = k1_med["median_timed_ms"]
    alg_full = PATHS_FULL * (8 * (2 * T_FULL + RY_FULL + 6) + 1)
    derived = {
        "K1_count_kernel": k1_name,
        "K1_count_median_timed_ms_from_kernel_trace": k1_ms,
        "K1_count_timed_launches": {k: k1_med[k] for k in ("timed_calls", "skipped_warmup_calls", "min_timed_ms", "max_timed_ms", "mean_timed_ms")},
        "K1_count_avg_ms_from_kernel_stats_all_launches": k1_mean_all_ms,
        "K1_count_roofline_frac_from_profile": (233 * 79 + 600 * 167) * PATHS_COUNT / (k1_ms * 1e-3) / 39.3e12,
        "K1_count_valu_wave_insts_per_path_month": m(k0, "SQ_INSTS_VALU") / wave_months,
        "K1_count_salu_wave_insts_per_path_month": m(k0, "SQ_INSTS_SALU") / wave_months,
        "K1_count_fp64_add_mul_fma_per_path_month":
            (m(k0, "SQ_INSTS_VALU_ADD_F64") + m(k0, "SQ_INSTS_VALU_MUL_F64") + m(k0, "SQ_INSTS_VALU_FMA_F64")) / wave_months,
        "K1_count_trans_f64_per_path_month": m(k0, "SQ_INSTS_VALU_TRANS_F64") / wave_months,
        "K1_count_lds_insts_per_path_month": m(k0, "SQ_INSTS_LDS") / wave_months,
        "K1_count_hbm_bytes_per_launch": (2 * m(k0, "FETCH_SIZE") + m(k0, "WRITE_SIZE")) * 1024,
        # (cycles counted in the PMC pass over the duration measured in the TRACE pass: the shader clock only if both
        #  passes ran the kernel at the same speed — they are separate runs)
        "GRBM_GUI_ACTIVE_per_XCD_over_trace_duration_GHz": m(k0, "GRBM_GUI_ACTIVE") / 8 / (k1_ms * 1e-3) / 1e9,
        "K1_full_WRITE_SIZE_bytes_per_launch": m(k2, "WRITE_SIZE") * 1024,
        "K1_full_algorithmic_write_bytes": alg_full,
        "K1_full_write_efficiency_algorithmic_over_measured": alg_full / (m(k2, "WRITE_SIZE") * 1024),
        "K3_bracket_slab_bytes_algorithmic": 8 * PATHS_FULL * (2 * T_FULL + RY_FULL),
        "K3_bracket_FETCH_SIZE_x2_bytes_per_launch": 2 * m(kb, "FETCH_SIZE") * 1024,
        "K3_bracket_WRITE_SIZE_bytes_per_launch": m(kb, "WRITE_SIZE") * 1024,
    }
    kb_med = next(v for k, v in med.items() if "rq_slab_kernel<16>" in k)
    kb_ms = kb_med["median_ms"]
    derived["K3_bracket_launches"] = {k: kb_med[k] for k in ("calls", "min_ms", "max_ms", "mean_ms")}
    k3_names = ("rq_tiny", "rq_count_kernel", "rq_slab_kernel", "rq_refine", "rq_collect", "rq_select", "rq_hist", "rq_cand", "rq_scan", "rq_init", "rq_flag")
    k3_rows = [r for r in stat_rows if any(t in r["Name"] for t in k3_names)]
    calls = max(1, min(int(r["Calls"]) for r in stat_rows if "rq_slab_kernel<16>" in r["Name"]))
    derived["K3_launches_per_call"] = sum(int(r["Calls"]) for r in k3_rows) / calls
    derived["K3_kernel_ms_per_call_sum"] = sum(float(r["TotalDurationNs"]) for r in k3_rows) / calls / 1e6
    derived["K3_valu_busy_slab_pass"] = (m(kb, "SQ_ACTIVE_INST_VALU") * 4 / (1024 * m(kb, "GRBM_GUI_ACTIVE") / 8)) if "SQ_ACTIVE_INST_VALU" in kb else None
    derived["K3_bracket_avg_ms_from_kernel_stats"] = kb_ms      # (key kept for bench.py; since round 4 the MEDIAN over the launches)
    derived["K3_bracket_achieved_TBps"] = derived["K3_bracket_slab_bytes_algorithmic"] / (kb_ms * 1e-3) / 1e12
    derived["K3_bracket_traffic_over_algorithmic"] = (derived["K3_bracket_FETCH_SIZE_x2_bytes_per_launch"] +
                                                        derived["K3_bracket_WRITE_SIZE_bytes_per_launch"]) / derived["K3_bracket_slab_bytes_algorithmic"]
    # what the figures belong to: the commit the collection ran on (the working tree must be clean when it is sent to the box)
    import subprocess
    import datetime
    try:
        commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True, cwd=os.path.dirname(os.path.abspath(__file__))).strip()
        dirty = bool(subprocess.check_output(["git", "status", "--porcelain", "--", "monte_carlo_retirement_amd", "bench.py"], text=True,
                                             cwd=os.path.dirname(os.path.abspath(__file__))).strip())
    except Exception:  # noqa: BLE001
        commit, dirty = None, None
    try:   # which card: the GPU's unique id (the host name of a pool box is its container runtime's)
        import re
        m = re.search(r"Unique ID:\s*(0x[0-9a-fA-F]+)", open(os.path.join(src, "box.txt")).read())
        box = [("gpu " + m.group(1)) if m else None]
    except OSError:
        box = []
    shutil.copy(os.path.join(src, "kt_medians.json"), os.path.join(dst, "kernel_trace_medians.json"))
    provenance = {"round": ROUND, "commit": commit, "box": box[0] if box else None, "tree_dirty_when_summarised": dirty, "collected_utc": datetime.datetime.utcfromtimestamp(os.path.getmtime(stats)).isoformat() + "Z",
                  "K1_count_kernel_ms": k1_ms, "K3_slab_kernel_ms": kb_ms}
    derived["provenance"] = provenance
    json.dump({
        "command": "rocprofv3 --pmc <set> --output-format csv -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-s60 --no-search "
                   "(separate passes: SQ, SQ instruction mix, FETCH_SIZE, WRITE_SIZE; tools/collect_profiles.sh)",
        "note": "FETCH_SIZE/WRITE_SIZE in KiB; FETCH_SIZE doubled per the gfx950 correction (MI355X_MICROARCH.md, HBM)",
        "kernels": summ, "derived": derived,
    }, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
    json.dump({
        "round": ROUND, "source": os.path.join(dst, "pmc_summary.json"), "provenance": provenance,
        "path_kernel_count_only_bytes_per_launch": derived["K1_count_hbm_bytes_per_launch"],
        "note": "(2*FETCH_SIZE + WRITE_SIZE)*1024 per launch of 1e6 paths; the algorithmic traffic of the count-only kernel "
                "is ~3907 workgroups x (2 + <=102) 8-byte atomics",
    }, open(os.path.join(os.path.dirname(os.path.abspath(dst)), "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(derived, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
