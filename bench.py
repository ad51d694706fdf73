#!/usr/bin/env python3
"""Headline benchmark: paths/sec of the per-path Monte Carlo kernel (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]        (N > 1: starts its own N ranks, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (same ranks, outer launcher)

A "step" is one pass of the hot path over one batch of synthetic paths: ONE launch of the path
kernel over `--paths` (default 1e6) paths per GPU of the `config.json` scenario at
working_months=233 (833 months/path), success-count only (BASELINE.json configs[1]), followed —
when N > 1 — by the single all-reduce of the counter vector (RCCL).  Independent path ranges
shard across ranks by GLOBAL path index (weak scaling: per-GPU work is fixed).

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel: algorithmic fp64 ops / measured
kernel time vs the fp64 vector-issue peak) and `cpu_baseline` (the CPU oracle timed on this box's
host cores on a bounded sample of the same workload: one thread, 16 threads and every usable core), plus one block per
remaining BASELINE config: `accuracy_10k` (configs[0]'s 10k-path fixture), `hbm_kernels` / `hbm_kernels_rho0` (configs[2]:
jorge.json with rho = 0.3 and as shipped), `class_api_1e7` (configs[2] through the drop-in class), `s60` (configs[3]),
`search` (configs[4]: the whole bracket + bisection search at 50 000 paths per probe and the 10^6-path
final run, through the drop-in class) and `numpy_stream` (the literal-seed mode's rate).  `config.ranks_seen` / `config.devices` say what the process
group actually contained.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

WORKING_MONTHS = 233          # config.json scenario, SURVEY §8 C1: 833 months/path, T=71
ALGO_OPS_PER_PATH = 233 * 79 + 600 * 167  # 79 ops/accumulation month, 167/retirement month
# fp64 vector peak: 78.6 TFLOP/s (AMD spec; = 256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz x 2 flop/FMA).
# The algorithmic count of SURVEY 8d is in OPERATIONS of the reference's arithmetic (an add, a multiply, a division and
# an exp call each count 1), whose roundings the state machine reproduces one by one (-ffp-contract=off), so the
# applicable ceiling is one operation per lane and clock: 39.3 T fp64 lane-ops/s.  (The kernel does issue FMAs where
# the reference's roundings allow: inside exp / log / sincos, the Newton steps of the divisions, a + b z.)
# MEASURED issue ceiling: tools/ubench/valu_cost.hip puts a dependent-free stream of fp64 add / mul / fma at 4.8-5.3
# cycles per wave-instruction on a SIMD (profiles/valu_cost_gfx950.txt), i.e. ~33 T lane-ops/s at the 2.27 GHz the
# chip holds under this load; `roofline.frac_of_measured_issue_ceiling` is quoted against that.
FP64_LANE_OPS_PEAK_T = 39.3
FP64_MEASURED_ISSUE_CEILING_T = 33.0
HBM_PEAK_GBS = 8000.0


def cpu_baseline(params, n_threads: int, paths_per_thread: int):
    """Oracle (C restatement of the reference, scalar fp64) on host threads; bounded sample."""
    from oracle import oracle as O

    O.lib()
    done = [0] * n_threads

    def work(t):
        r = O.run_batch(params, 12345, 1, t * paths_per_thread, paths_per_thread, WORKING_MONTHS,
                        want_summary=False, want_trajectories=False)
        done[t] = int(r["counters"][1])

    threads = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
    t0 = time.perf_counter()
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    dt = time.perf_counter() - t0
    return sum(done) / dt, dt


def host_cpu_info():
    """What the CPU baseline ran on (`nproc` and CPU model stated)."""
    model = None
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = os.cpu_count() or 1
    quota = None                        # cgroup v2 CPU quota of this job ("max 100000" = none): cores' worth of time per period
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, period = fh.read().split()[:2]
            quota = None if q == "max" else float(q) / float(period)
    except (OSError, ValueError):
        pass
    return {"host_cores": os.cpu_count(), "usable_cores": usable, "cgroup_cpu_quota_cores": quota, "cpu_model": model}


def cpu_baseline_block(params, n_threads: int, paths_per_thread: int, single_thread_paths: int, all_cores_seconds: float = 10.0):
    """`cpu_baseline` of the JSON line ("single thread and all cores"): the oracle on ONE thread, on
    `n_threads` threads (`few_threads`) and on EVERY core this process may run on (`all_cores`, the figure `value` reports: what
    the host could do), the box's core count, CPU model and cgroup CPU quota beside them.  The all-core sample is sized from
    a short probe run so that it takes about `all_cores_seconds` whatever the box (bare 256 hardware threads or a cgroup
    share of them)."""
    info = host_cpu_info()
    usable = int(info["usable_cores"])
    n_threads = max(1, min(int(n_threads), usable))
    v1, s1 = cpu_baseline(params, 1, single_thread_paths)
    vf, sf = cpu_baseline(params, n_threads, paths_per_thread)
    few = {"value": vf, "unit": "paths/s", "cores": n_threads,
           "sample": f"{n_threads} threads x {paths_per_thread} paths ({sf:.1f} s wall)"}
    if all_cores_seconds > 0 and usable > n_threads:
        probe_per = max(200, int(v1 * 0.25))                      # ~0.25 s per thread if every thread has a core of its own
        vp, _ = cpu_baseline(params, usable, probe_per)
        per = max(probe_per, int(vp * all_cores_seconds / usable))
        va, sa = cpu_baseline(params, usable, per)
        allc = {"value": va, "unit": "paths/s", "cores": usable,
                "sample": f"{usable} threads (every logical core this process may run on) x {per} paths ({sa:.1f} s wall), "
                          f"sized from a {usable} x {probe_per}-path probe"}
    else:
        allc = dict(few, sample=few["sample"] + " (= every usable core)" if usable <= n_threads else few["sample"] + " (all-core run disabled)")
    # A cgroup CPU quota below the core count (this pool's one-GPU boxes: 16 cores' worth of time on a 256-thread host) throttles
    # the all-core run to the quota — 256 runnable threads then finish FEWER paths per second than 16: `value` is the best the
    # host side of this job reached, and `quota_limited` says that "all cores" was not the whole machine.
    quota = info.get("cgroup_cpu_quota_cores")
    best = allc if allc["value"] >= few["value"] else few
    return {
        "value": best["value"],
        "unit": "paths/s",
        "cores": best["cores"],
        "kind": "port",
        "sample": best["sample"] + " of the same workload (oracle/mcr_oracle.c, scalar fp64, one path range per thread)",
        "quota_limited": bool(quota is not None and quota < usable),
        "all_cores": allc,
        "few_threads": few,
        "single_thread": {"value": v1, "unit": "paths/s", "cores": 1,
                          "sample": f"1 thread x {single_thread_paths} paths ({s1:.1f} s wall)"},
        **info,
        "reference_cpython": "113 paths/s on 1 core, 574 paths/s with 8 processes (BASELINE.md 2: the reference's own NumPy/CPython "
                             "path, survey container, 8 vCPU Xeon 2.1 GHz; its files cannot travel to this box)",
    }


def aux_hbm_kernels(torch, n, rho=0.3, allocations=3):
    """The HBM-side kernels of the path on the BASELINE configs[2] shape (jorge.json, wm=75: T=48 yearly samples,
    40 WR rows), n paths: K1 with full trajectory output (write efficiency), K3 row quantiles and K2 histogram (achieved
    algorithmic GB/s vs the 8 TB/s HBM peak).  SURVEY 8(d) B3 asks for the scenario twice: with the
    equity/inflation correlation of BASELINE's description (rho = 0.3: the `hbm_kernels` block) and AS SHIPPED (the file has
    no `equity_inflation_correlation` key: rho = None -> the Config default 0.0, backend/config.py:85-90: `hbm_kernels_rho0`).
    Not part of `value`; reported so the memory-bound side of the path has a measured roofline too."""
    from monte_carlo_retirement_amd import Config, params_from_config
    from monte_carlo_retirement_amd import aggregation as A
    from monte_carlo_retirement_amd import engine as E

    with open(os.path.join(REPO, "scenarios", "jorge.json")) as fh:
        cfg = Config(**dict(json.load(fh), seed=12345, **({} if rho is None else {"equity_inflation_correlation": rho})))
    p = params_from_config(cfg)
    b = E.DeviceBatch(p, 75, n, want="full")
    T, ry = b.sizes.trajectory_len, b.sizes.retirement_years

    def timed(fn, reps=15, warm=3):
        # steady state: after an idle gap the first calls of a ~2 ms kernel sequence run 10-20 % slow while the clocks ramp
        # (tools/k3_series.py: 2.72, 2.61, 2.48, 2.44, 2.34, then 2.27 +- 0.01 ms); the first call also allocates scratch
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return sorted(ts)[len(ts) // 2]

    b.launch(12345, 1, 0)
    ms_k1 = timed(lambda: b.launch(12345, 1, 0))
    bytes_k1 = n * (8 * (2 * T + ry + 6) + 1)
    # K3 depends on where the driver places the slab: the same kernels on the same data run the slab pass at 5.3 or at 6.0
    # TB/s from one allocation of the batch to the next (DESIGN.md 5 / LABNOTES.md, tools/k3_alloc_modes.py).  Three allocations, the
    # median reported, all three listed.
    k3_by_alloc = [timed(lambda: A.band_quantiles(b, n))]
    fallback_rows = A.last_fallback_rows()
    for _ in range(allocations - 1):
        del b
        torch.cuda.empty_cache()
        b = E.DeviceBatch(p, 75, n, want="full")
        b.launch(12345, 1, 0)
        k3_by_alloc.append(timed(lambda: A.band_quantiles(b, n)))
        fallback_rows = max(fallback_rows, A.last_fallback_rows())
    ms_k3 = sorted(k3_by_alloc)[len(k3_by_alloc) // 2]
    bytes_k3 = 8 * n * (2 * T + ry)      # algorithmic: every entry of the slab has to be read once
    ms_k2 = timed(lambda: A.success_histogram(b.summary["final_balance"], b.success, 100))
    bytes_k2 = 2 * 9 * n                 # min/max pass + bin pass, 8 B value + 1 B flag each
    profiled = None   # the slab pass alone, from the committed rocprofv3 summary of this same command (latest round)
    for rnd, kernel in (("r04", "mcr::rq_slab_kernel<16>"), ("r03", "mcr::rq_slab_kernel<16>"), ("r02", "mcr::rq_count_kernel<16, true>"), ("r01_final", "mcr::rq_bracket_kernel")):
        try:
            with open(os.path.join(REPO, "profiles", rnd, "pmc_summary.json")) as fh:
                dv = json.load(fh)["derived"]
            profiled = {"kernel": kernel,
                        "ms": dv["K3_bracket_avg_ms_from_kernel_stats"], "achieved_TBps": dv["K3_bracket_achieved_TBps"],
                        "frac_of_hbm_peak": dv["K3_bracket_achieved_TBps"] * 1e3 / HBM_PEAK_GBS,
                        "traffic_over_algorithmic": dv["K3_bracket_traffic_over_algorithmic"],
                        "valu_busy": dv.get("K3_valu_busy_slab_pass"),
                        "paths": 10_000_000 if rnd != "r01_final" else 4_000_000, "timing": "median over the launches" if rnd == "r04" else "mean over the launches",
                        # a STATIC figure read from a committed file, not measured in this run: this says which build it belongs to
                        "provenance": dv.get("provenance", {"round": rnd, "commit": None}),
                        "source": f"profiles/{rnd}/pmc_summary.json (rocprofv3 --kernel-trace --stats + FETCH_SIZE/WRITE_SIZE passes)"}
            break
        except (OSError, KeyError, ValueError):
            continue
    return {
        "workload": f"BASELINE configs[2] shape: jorge.json " + (f"rho={rho}" if rho is not None else "as shipped (rho = 0.0, the Config default)")
                    + f", wm=75 (555 months), {n} paths, T={T}, ry={ry}",
        "equity_inflation_correlation": cfg.equity_inflation_correlation,
        "K1_full_output": {"ms": ms_k1, "paths_per_s": n / ms_k1 * 1e3, "algorithmic_write_bytes": bytes_k1,
                           "write_GBps": bytes_k1 / ms_k1 / 1e6, "frac_of_hbm_peak": bytes_k1 / ms_k1 / 1e6 / HBM_PEAK_GBS,
                           "note": "compute-bound: the time-major trajectory stores hide under the fp64 VALU work"},
        "K3_row_quantiles": {"ms": ms_k3, "by_allocation_ms": k3_by_alloc, "rows": 2 * T + ry, "algorithmic_read_bytes": bytes_k3,
                             "GBps": bytes_k3 / ms_k3 / 1e6, "frac_of_hbm_peak": bytes_k3 / ms_k3 / 1e6 / HBM_PEAK_GBS,
                             "fallback_rows": fallback_rows, "slab_pass_profiled": profiled,
                             # STATIC, from a committed microbenchmark run (tools/ubench/hbm_read.hip), not measured here: what this part's
                             # memory system gives a kernel of the slab pass's shape
                             "memory_system_measured": {"read_only_sweep_TBps": [6.5, 7.0], "read_sweep_with_4pct_writes_TBps": [4.9, 5.6],
                                                        "slab_pass_without_candidate_output_TBps": [6.6, 6.8],
                                                        "source": "profiles/r03/hbm_read_gfx950.txt, profiles/r03/k3_slab_parts.txt"},
                             "note": "one call over the [2T+ry] slab (bands of all rows), steady state (3 warm-up calls, median of 15; `ms` = the "
                                     "median over three allocations of the batch, `by_allocation_ms` lists them): seven "
                                     "launches + one word read back, results written straight into pinned host memory. Algorithmic bytes = ONE read "
                                     "of the slab; the pass that does it (rq_slab_kernel) runs at 5.3-6.0 TB/s depending on where the driver placed "
                                     "this process's slab (DESIGN.md 5) - the rate a bare read sweep with the pass's 4 % of candidate writes reaches "
                                     "on this part (`memory_system_measured`; reading alone: 6.5-7.0) -, the rest is the two sampling steps before it and the per-row cell selection "
                                     "after it (profiles/). fallback_rows = rows that needed the 4-pass radix select (-1: rows too short for the "
                                     "bracketed route)"},
        "K2_histogram": {"ms": ms_k2, "algorithmic_bytes": bytes_k2, "GBps": bytes_k2 / ms_k2 / 1e6},
    }


def numpy_stream_block(torch, n, reps=7):
    """The literal-seed mode (`rng="numpy"`): the reference's OWN random stream reproduced on the device — SeedSequence ->
    PCG64 -> ziggurat (backend/simulation.py:148-149,195-197,457-458) — so that the same `seed` gives the reference's numbers.
    config.json scenario, wm=233, n paths, count-only: paths/s of that kernel variant next to the engine's Philox stream's
    (`value`).  Median of `reps` event-timed launches after two warm-up launches.  Not part of `value`."""
    from monte_carlo_retirement_amd import Config, params_from_config
    from monte_carlo_retirement_amd import _native as N
    from monte_carlo_retirement_amd import engine as E

    with open(os.path.join(REPO, "scenarios", "config.json")) as fh:
        cfg = Config(**dict(json.load(fh), seed=12345))
    b = E.DeviceBatch(params_from_config(cfg), WORKING_MONTHS, n, want="count")
    rng = N.numpy_rng(12345)
    ts = []
    for i in range(reps + 2):
        b.zero_counters()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(); b.launch(rng, 1, 0); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = sorted(ts[2:])[reps // 2]
    ok, cnt = (int(v) for v in b.counters.cpu().tolist())
    return {"workload": f"config.json scenario, working_months={WORKING_MONTHS}, {n} paths, success-count only, rng='numpy': "
                        "SeedSequence(main_seed).spawn -> PCG64 -> ziggurat per path, the reference's own stream (literal seed parity)",
            "paths": n, "ms": ms, "paths_per_s": n / ms * 1e3, "success_probability": ok / max(1, cnt), "paths_counted": cnt,
            "kernel": "mcr::path_kernel<0, 1, 3, false, ...> (NumPy-stream variant: sequential generator per lane, no staging)",
            "note": "the reference's native run of this scenario and seed gives the same per-path flags (tests/test_numpy_rng_gpu.py)"}


def class_api_block(torch, n, hbm=None, reps=3):
    """The drop-in class end to end at the BASELINE configs[2] size: `RetirementMonteCarloSimulator.run_monte_carlo_simulations(75, n)`
    on jorge.json (rho = 0.3) — the reference's 7-tuple (reference backend/simulation.py:952-1128): n-row summary frame, nominal /
    real trajectory bands, withdrawal-rate bands, 5 sampled paths, observation counts.  Wall seconds per call (median of `reps`
    after one untimed call), of which kernels = the path kernel with full output + the band selection as timed in `hbm_kernels`
    on the same shape; the rest is the 56 B/path summary crossing the host link (on a copy stream, under the band selection),
    pinned-buffer allocation and frame assembly.  Not part of `value`."""
    from monte_carlo_retirement_amd import Config
    from monte_carlo_retirement_amd.simulation import RetirementMonteCarloSimulator

    with open(os.path.join(REPO, "scenarios", "jorge.json")) as fh:
        cfg = Config(**dict(json.load(fh), equity_inflation_correlation=0.3, seed=12345))
    sim = RetirementMonteCarloSimulator(cfg)
    sim.use_final_seeds()
    res = sim.run_monte_carlo_simulations(75, n)          # untimed: first-use allocations (device slab, pinned host buffers)
    times = []
    for _ in range(reps):
        del res
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = sim.run_monte_carlo_simulations(75, n)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    sec = sorted(times)[len(times) // 2]
    kern = None
    if hbm and "error" not in hbm:
        kern = (hbm["K1_full_output"]["ms"] + hbm["K3_row_quantiles"]["ms"]) * 1e-3
    summary_bytes = 49 * n
    return {"workload": f"BASELINE configs[2] through the class API: jorge.json rho=0.3, run_monte_carlo_simulations(75, {n}) -> the reference's 7-tuple",
            "paths": n, "seconds": sec, "all_seconds": times, "paths_per_s": n / sec, "kernel_seconds": kern,
            "summary_rows": int(len(res[0])), "summary_bytes_device_to_host": summary_bytes,
            "host_link_GBps_if_the_rest_were_all_transfer": (summary_bytes / max(1e-9, sec - kern) / 1e9) if kern else None,
            "success_probability_pct": float(res[0]["Success"].mean() * 100.0), "band_rows": int(res[1].shape[0]),
            "note": "kernel_seconds = K1 full output + K3 bands of the `hbm_kernels` block (same shape, same box, event-timed); the summary "
                    "download runs on a copy stream under the band selection, the frame is assembled last"}


def accuracy_10k():
    """The second half of BASELINE's metric: |p_gpu - p_ref| on the 10k-path config (config.json, wm=233).  p_ref and
    the per-path Success flags are the REFERENCE's own output, recorded in tests/golden/metric_10k_config_json.*
    (generate_golden.py ran the reference on the engine's Philox shocks): data files, no oracle involved."""
    import numpy as np

    from monte_carlo_retirement_amd import Config, params_from_config
    from monte_carlo_retirement_amd import engine as E

    gdir = os.path.join(REPO, "tests", "golden")
    with open(os.path.join(gdir, "metric_10k_config_json.json")) as fh:
        meta = json.load(fh)
    z = np.load(os.path.join(gdir, "metric_10k_config_json.npz"))
    n = int(meta["n_paths"])
    stream = {"search": 0, "final": 1}[meta["stream"]]
    res = E.run_batch_host(params_from_config(Config(**meta["cfg"])), meta["seed"], stream, 0, n, meta["working_months"],
                           want_trajectories=False)
    flags = np.unpackbits(z["success_bits"])[:n]
    p_gpu, p_ref = float(res["counters"][0]) / n, meta["success_count"] / n
    return {"config": "BASELINE configs[0]: config.json, working_months=233, %d paths" % n, "p_gpu": p_gpu, "p_reference": p_ref,
            "abs_error": abs(p_gpu - p_ref), "flipped_success_flags": int((res["success"] != flags).sum()),
            "source": "tests/golden/metric_10k_config_json.{json,npz} (the reference's own flags on identical shocks)"}


S60_EDGES = (1.0, 1.0e12, 100)   # fixed log-spaced bins of the s60 block: np.geomspace(lo, hi, n + 1)


def s60_block(torch, dist, world, n_total, reps=3, grouped=None, fixed_edges=True):
    """North-star shape (BASELINE configs[3]): S60 = config.json with initial_balance=2e6,
    inv1 volatility 0.15, rho=0.3, wm=120 (720-month paths); success counts + 100-bin histogram of the
    successful final balances over `n_total` paths IN TOTAL, sharded by global path range over the ranks.
    fixed_edges (the `s60` block): `distributed.run_sharded_histogram(hist_edges=...)` — the COUNT-ONLY kernel bins
    every path's final balance itself on fixed log-spaced edges, nothing per path touches HBM, and the
    ranks exchange ONE all-reduce of the integer block [counters | year bins | histogram].
    Otherwise (`s60_data_ranged`): np.histogram's data-ranged edges — summary-output kernel (49 B/path), min/max +
    bins (K2), three collectives.  End to end per repetition incl. allocation, collectives, download.  Not part of `value`."""
    import numpy as np

    from monte_carlo_retirement_amd import Config, params_from_config
    from monte_carlo_retirement_amd import distributed as D

    with open(os.path.join(REPO, "scenarios", "config.json")) as fh:
        cfg = Config(**dict(json.load(fh), initial_balance=2.0e6, inv1_returns_volatility=0.15,
                            equity_inflation_correlation=0.3, seed=12345))
    p = params_from_config(cfg)
    grouped = world > 1 if grouped is None else grouped
    edges = np.geomspace(S60_EDGES[0], S60_EDGES[1], S60_EDGES[2] + 1) if fixed_edges else None
    times, r = [], None
    for _ in range(reps):
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = D.run_sharded_histogram(p, 12345, 1, n_total, 120, n_bins=100, hist_edges=edges)
        torch.cuda.synchronize()
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
        if grouped:
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        times.append(float(dt.item()))
        del dt
    sec = sorted(times)[len(times) // 2]
    c = r["counts"]
    return {
        "workload": f"BASELINE configs[3] / north-star shape: S60 (config.json, initial_balance=2e6, inv1 vol 0.15, rho=0.3), "
                    f"wm=120 (720 months/path), {n_total} paths in total over {world} GPU(s), success count + 100-bin "
                    "histogram of successful final balances"
                    + (f" on fixed edges np.geomspace({S60_EDGES[0]:g}, {S60_EDGES[1]:g}, 101), binned inside the count-only path kernel"
                       if fixed_edges else " on np.histogram's data-ranged edges (summary-output kernel + min/max + bin kernels)"),
        "n_paths_total": n_total, "n_gpus": world, "seconds": sec, "paths_per_s": n_total / sec,
        "success_probability": c.success / max(1, c.paths), "paths_counted": c.paths,
        "hist_total": int(r["hist_bins"].sum()), "hist_outside_edges": int(c.success - int(r["hist_bins"].sum())),
        "exchange": r["exchange"],
        "note": "end to end incl. allocation and download; median of %d repetitions, max over ranks" % reps,
    }


def search_block(torch, dist, world, device, grouped, reps=3):
    """BASELINE configs[4]: `find_minimum_working_months` — bracket, bisection, verification window;
    reference backend/simulation.py:1138-1342 — on the config.json scenario with num_simulations_search = 50 000 paths per
    probed month, then the final run of num_simulations_main = 10^6 paths, all through the drop-in class
    (`RetirementMonteCarloSimulator`).  Probes are count-only launches; the months of one round share their accumulation
    sweep.  Under a process group the probes of a round are split by candidate month over the ranks (one all-reduce per
    round) and the final run is sharded by path range; the curve must then equal the single-GPU search's, which every
    rank replays locally as a check (`distributed.local_only`).  Not part of `value`."""
    import hashlib

    from monte_carlo_retirement_amd import Config
    from monte_carlo_retirement_amd import distributed as D
    from monte_carlo_retirement_amd.simulation import RetirementMonteCarloSimulator

    with open(os.path.join(REPO, "scenarios", "config.json")) as fh:
        cfg = Config(**dict(json.load(fh), seed=12345, num_simulations_search=50_000, num_simulations_main=1_000_000))

    def fence():
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    def wall_max(dt):
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        if grouped:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def one_search():
        sim = RetirementMonteCarloSimulator(cfg, device=device)
        calls, inner = [], sim._probe_many          # one call = one round of launches (+ one all-reduce under a group)
        sim._probe_many = lambda months, n: (calls.append(len(months)), inner(months, n))[1]
        fence()
        t0 = time.perf_counter()
        months, prob, curve = sim.find_minimum_working_months(verbose=False)
        torch.cuda.synchronize()
        return sim, months, prob, curve, calls, wall_max(time.perf_counter() - t0)

    one_search()                                   # untimed: first-use allocations, module load of the probe kernels
    runs = [one_search() for _ in range(reps)]
    runs.sort(key=lambda r: r[-1])
    sim, months, prob, curve, calls, sec = runs[len(runs) // 2]
    assert all(r[3] == curve for r in runs), "the search is deterministic: every repetition must replay the same curve"
    probes = len(curve)
    out = {
        "workload": "BASELINE configs[4]: config.json, num_simulations_search=50000 paths per probed month, bracket + bisection + "
                    f"verification window, then the final run of 1000000 paths, over {world} GPU(s)",
        "n_gpus": world, "paths_per_probe": cfg.num_simulations_search,
        "months_found": months, "probability_pct": prob, "probes": probes,
        "probe_rounds": len(calls), "months_evaluated": sum(calls), "largest_round": max(calls) if calls else 0,
        "search_seconds": sec, "ms_per_probe": sec / max(1, probes) * 1e3,
        "probe_paths_per_s": probes * cfg.num_simulations_search / sec,
        "curve_sha16": hashlib.sha256(json.dumps(curve, sort_keys=True).encode()).hexdigest()[:16],
        "curve_head": curve[:3], "curve_tail": curve[-2:],
        "probe_split": ("by candidate month over the ranks, 1 all-reduce(sum) of the per-candidate counters per round"
                        if world > 1 else "none (1 GPU): the months of a round share one accumulation sweep"),
    }
    if D.is_active():
        with D.local_only():                       # the same search on this rank's GPU alone
            lsim = RetirementMonteCarloSimulator(cfg, device=device)
            lm, lp, lcurve = lsim.find_minimum_working_months(verbose=False)
        same = torch.tensor([1 if (lm, lp, lcurve) == (months, prob, curve) else 0], dtype=torch.int64, device="cuda")
        dist.all_reduce(same, op=dist.ReduceOp.MIN)
        out["equals_single_gpu_search"] = bool(int(same.item()))
    # the final run (simulation.py:1130-1136 of the caller's flow: use_final_seeds, run, success probability)
    if months >= 0:
        sim.use_final_seeds()
        sim.run_monte_carlo_simulations(months, cfg.num_simulations_main)      # untimed first call: allocations
        times = []
        res = None
        for _ in range(reps):
            fence()
            t0 = time.perf_counter()
            res = sim.run_monte_carlo_simulations(months, cfg.num_simulations_main)
            torch.cuda.synchronize()
            times.append(wall_max(time.perf_counter() - t0))
        fsec = sorted(times)[len(times) // 2]
        out.update({
            "final_run_paths": cfg.num_simulations_main, "final_run_seconds": fsec,
            "final_run_paths_per_s": cfg.num_simulations_main / fsec,
            "final_success_probability_pct": sim._success_probability(res[0]),
            "final_run_outputs": "the reference's 7-tuple: per-path summary frame (10^6 x 7), nominal / real trajectory bands, "
                                 "withdrawal-rate bands, 5 sampled paths, observation counts",
            "end_to_end_seconds": sec + fsec,
        })
    return out


def launch_ranks(n_ranks: int) -> int:
    """`python bench.py --gpus N` without an outer launcher: start N copies of this script, one rank per GPU
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment, as torch.distributed.run would), wait, and
    return non-zero if any rank failed.  The parent never imports torch or touches a GPU; children are plain
    child processes (nothing is exec'ed over a process that has initialised the GPU).  Rank 0 inherits stdout,
    so its JSON line is this command's JSON line.  Every rank's stderr goes to a file of its own; the parent
    replays rank 0's in full and the tail of every rank that failed or had to be killed.
    Nothing here can wait forever: an overall deadline (MCR_BENCH_DEADLINE_S, default 1200 s — ranks stuck in the
    RCCL rendezvous are the likeliest first-contact failure on an 8-GPU node) and, once one rank has failed, 30 s
    for the others; children that are still running then are killed by PID and the exit code is non-zero."""
    import socket
    import subprocess
    import tempfile

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    logdir = tempfile.mkdtemp(prefix="mcr_bench_ranks_")
    procs, logs = [], []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        # The host driver of this pool supports only dmabuf IPC: with the legacy IPC mode RCCL's (and torch's) cross-process
        # buffer sharing fails with `hipIpcGetMemHandle: invalid argument`.  The image exports 0 already; keep a caller's choice.
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        logs.append(open(os.path.join(logdir, f"rank{r}.stderr"), "w+"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL, stderr=logs[r]))

    def tail(r, n_lines=40):
        logs[r].flush()
        logs[r].seek(0)
        return logs[r].read().splitlines()[-n_lines:]

    overall = time.monotonic() + float(os.environ.get("MCR_BENCH_DEADLINE_S", "1200"))
    rc, pending, deadline, killed = 0, set(range(n_ranks)), None, []
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0:
                rc = rc or code
                print(f"bench.py: rank {r} exited with code {code}", file=sys.stderr, flush=True)
                if deadline is None:
                    deadline = time.monotonic() + 30.0  # the others are stuck in a collective: do not wait for them
        now = time.monotonic()
        if pending and ((deadline is not None and now > deadline) or now > overall):
            why = "a peer failed" if (deadline is not None and now > deadline) else "overall deadline reached"
            killed = sorted(pending)
            print(f"bench.py: killing ranks {killed} ({why})", file=sys.stderr, flush=True)
            for r in killed:
                procs[r].kill()  # exact PIDs this launcher started
            for r in killed:
                procs[r].wait()
            rc = rc or 124
            break
        time.sleep(0.05)
    logs[0].flush()
    logs[0].seek(0)
    sys.stderr.write(logs[0].read())
    for r in range(1, n_ranks):
        if procs[r].returncode != 0 or r in killed:
            for line in tail(r):
                print(f"[rank {r}] {line}", file=sys.stderr)
    sys.stderr.flush()
    for f in logs:
        f.close()
    import shutil

    shutil.rmtree(logdir, ignore_errors=True)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 200 timed steps of 1e6 paths = 1.4 s of GPU work (long enough for an SMI sampler to see the card busy)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--paths", type=int, default=1_000_000, help="paths per GPU per step")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse on one GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-aux", action="store_true", help="skip the HBM-side kernels (trajectory write / quantiles / histogram)")
    ap.add_argument("--aux-paths", type=int, default=10_000_000, help="paths of the BASELINE configs[2] block (hbm_kernels)")
    ap.add_argument("--no-s60", action="store_true", help="skip the north-star S60 block")
    ap.add_argument("--s60-paths", type=int, default=100_000_000, help="TOTAL paths of the BASELINE configs[3] block (s60)")
    ap.add_argument("--no-search", action="store_true", help="skip the BASELINE configs[4] block (search)")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--cpu-paths-per-thread", type=int, default=60_000)
    ap.add_argument("--cpu-single-thread-paths", type=int, default=20_000)
    ap.add_argument("--cpu-all-cores-seconds", type=float, default=10.0, help="target wall time of the all-core oracle run (0 = skip it)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))  # before torch is imported: the launcher never touches a GPU

    import torch
    import torch.distributed as dist

    from monte_carlo_retirement_amd import Config, params_from_config
    from monte_carlo_retirement_amd import engine as E

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    device = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(device)
    # `grouped`: the N > 1 code path (process group, per-step exchange, collectives in the s60 block).  MCR_BENCH_FORCE_GROUP=1
    # takes it with ONE rank too: the only way to run the RCCL calls of this file on a one-GPU box (tests/test_bench_gpu.py).
    grouped = world > 1 or os.environ.get("MCR_BENCH_FORCE_GROUP") == "1"
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        import datetime

        # a finite rendezvous / collective timeout: a rank that never arrives fails the job instead of hanging it
        tmo = datetime.timedelta(seconds=float(os.environ.get("MCR_BENCH_PG_TIMEOUT_S", "600")))
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device), timeout=tmo)
        else:
            dist.init_process_group(args.backend, timeout=tmo)
    comm_dev = torch.device("cuda", device) if (not grouped or args.backend == "nccl") else torch.device("cpu")
    # What the process group actually contains: an all-reduce of ones (how many ranks took part in a collective) and
    # every rank's device, gathered to all ranks.
    me = {"rank": rank, "local_rank": local_rank, "device_index": device, "device_name": torch.cuda.get_device_name(device),
          "visible_devices": torch.cuda.device_count(), "pid": os.getpid()}
    # (the marketing name needs /opt/amdgpu/share/libdrm/amdgpu.ids, which a box may lack: "AMD Radeon Graphics"; the
    #  architecture string, CU count and memory size identify the part either way)
    props = torch.cuda.get_device_properties(device)
    for key, attr in (("gcn_arch", "gcnArchName"), ("compute_units", "multi_processor_count"), ("total_memory", "total_memory"),
                      ("pci_bus_id", "pci_bus_id")):
        if hasattr(props, attr):
            me[key] = getattr(props, attr)
    if grouped:
        ones = torch.ones(1, dtype=torch.int64, device=comm_dev)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
        devices_seen = [None] * dist.get_world_size()
        dist.all_gather_object(devices_seen, me)
        backend_seen = dist.get_backend()
    else:
        ranks_seen, devices_seen, backend_seen = 1, [me], None

    with open(os.path.join(REPO, "scenarios", "config.json")) as fh:
        cfg = Config(**dict(json.load(fh), seed=12345))
    params = params_from_config(cfg)
    n = args.paths
    batch = E.DeviceBatch(params, WORKING_MONTHS, n, want="count", device=device)
    # The exchange step works on a COPY of the rank's running totals: all-reducing the accumulation vector in place
    # would feed every step's global sum back into the next step's local counters.
    # Two copies, used alternately, and the all-reduce is asynchronous: step k's exchange runs on the communication
    # stream while step k+1 computes (nothing on the launch stream waits for it until its buffer comes up again).
    exch2 = [torch.zeros(batch.reduce_vec.shape, dtype=batch.reduce_vec.dtype, device=comm_dev) for _ in range(2)]
    pending = [None, None]
    exch = exch2[0]

    def exchange(i):
        # the path's single exchange step: counters + year bins, summed over the ranks
        j = i & 1
        if pending[j] is not None:
            pending[j].wait()
        exch2[j].copy_(batch.reduce_vec, non_blocking=True)
        pending[j] = dist.all_reduce(exch2[j], async_op=True)

    def drain():
        for j in range(2):
            if pending[j] is not None:
                pending[j].wait()
                pending[j] = None

    def step(i):
        # global path index: step-major, then rank (every path of the job is distinct)
        batch.launch(12345, 1, (i * world + rank) * n)
        if grouped:
            exchange(i)

    def fence():
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    drain()
    fence()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    batch.zero_counters()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()
        batch.launch(12345, 1, ((args.warmup + i) * world + rank) * n)
        ev[i][1].record()
        if grouped:
            exchange(i)
    drain()
    fence()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
    if grouped:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    kern_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps  # HIP events on the launch stream
    # local running totals of the timed steps; with N ranks the last exchange holds the job's totals
    counters = (exch2[(args.steps - 1) & 1] if grouped else batch.reduce_vec)[:2].cpu().tolist()

    def guarded(fn, *a, **kw):
        try:
            return fn(*a, **kw)
        except Exception as exc:  # never lose the headline line to an auxiliary block
            return {"error": f"{type(exc).__name__}: {exc}"}

    s60 = s60_ranged = search = None
    if not args.no_s60:
        s60 = guarded(s60_block, torch, dist, world, args.s60_paths, grouped=grouped, fixed_edges=True)
        s60_ranged = guarded(s60_block, torch, dist, world, args.s60_paths, grouped=grouped, fixed_edges=False)
    if not args.no_search:
        search = guarded(search_block, torch, dist, world, device, grouped)

    if rank == 0:
        total_paths = n * world * args.steps
        value = total_paths / dt
        achieved_t = ALGO_OPS_PER_PATH * n / (kern_ms * 1e-3) / 1e12
        traffic, traffic_prov = None, None
        pmc = os.path.join(REPO, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            with open(pmc) as fh:
                tj = json.load(fh)
            traffic = tj.get("path_kernel_count_only_bytes_per_launch")
            traffic_prov = tj.get("provenance", {"round": tj.get("round"), "commit": None})
        out = {
            "metric": "paths/sec (whole node), config.json scenario, 833-month paths, success-count only",
            "value": value,
            "unit": "paths/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1]: config.json scenario, working_months=233 (833 months/path), "
                            f"{n} paths per GPU per step, success-count only (no trajectory writeback)",
                "paths_per_gpu_per_step": n,
                "rng": "Philox4x32-10 + Box-Muller, counter=(path,month,stream), key=seed",
                "parallelism": f"path-range sharding x{world}" + (
                    f" + 1 all-reduce(sum) of the {exch.numel()}-word counter/bin vector per step ({args.backend}), overlapped with the next step's compute" if grouped else ""),
                "ranks_seen": ranks_seen,          # all-reduce(sum) of one 1 per rank: the ranks that took part in a collective
                "backend": backend_seen,           # torch.distributed backend of the group ("nccl" = RCCL), None without a group
                "devices": devices_seen,           # every rank's own report (all_gather_object)
            },
            "roofline": {
                "kernel": "mcr::path_kernel<0, 0, 3, false, false, 3, false, false, false>  (MODE 0 count-only, Philox, realized-gains tax on both assets, no annual tax, "
                          "PHASE 3 = time-sliced path blocks: the form launches of 10^6 paths take; <..., 0, ...> = the plain launch of other sizes)",
                "bound": "valu_fp64",
                "achieved": achieved_t,
                "peak": FP64_LANE_OPS_PEAK_T,
                "unit": "TFLOP/s",
                "frac": achieved_t / FP64_LANE_OPS_PEAK_T,
                "measured_issue_ceiling": FP64_MEASURED_ISSUE_CEILING_T,
                "frac_of_measured_issue_ceiling": achieved_t / FP64_MEASURED_ISSUE_CEILING_T,
                "traffic": traffic,
                # (`traffic` is a STATIC figure from profiles/pmc_traffic.json — PMC passes cannot run inside this process; the
                #  provenance says which build and which profiled kernel time it belongs to; `kernel_ms` below is measured live)
                "traffic_provenance": traffic_prov,
                "kernel_ms": kern_ms,
                "algorithmic_ops_per_path": ALGO_OPS_PER_PATH,
                "note": "no dense contraction and ~0 HBM bytes/path in this variant: the bound is fp64 VALU issue (SURVEY 8d). "
                        "achieved = SURVEY 8d's SOURCE-LEVEL operation count of the reference's arithmetic (each + - * / min max compare "
                        "and each exp / log / sqrt / sin / cos call = 1) per second; peak = 78.6 TFLOP/s spec / 2 = one fp64 lane-operation "
                        "per lane and clock.  The kernel itself issues FMAs where the reference's roundings allow (growth factors, "
                        "Newton steps of the divisions), so this is a count of useful work against the issue peak, not an instruction count",
            },
            "success_probability": counters[0] / max(1, counters[1]),
            "paths_counted": counters[1],
        }
        try:
            out["accuracy_10k"] = accuracy_10k()
        except Exception as exc:  # never lose the headline line to an auxiliary block
            out["accuracy_10k"] = {"error": f"{type(exc).__name__}: {exc}"}
        if s60 is not None:
            out["s60"] = s60
            out["s60_data_ranged"] = s60_ranged
        if search is not None:
            out["search"] = search
        if not args.no_aux and world == 1:
            out["hbm_kernels"] = guarded(aux_hbm_kernels, torch, args.aux_paths)
            out["hbm_kernels_rho0"] = guarded(aux_hbm_kernels, torch, args.aux_paths, rho=None, allocations=1)
            out["numpy_stream"] = guarded(numpy_stream_block, torch, args.paths)
            out["class_api_1e7"] = guarded(class_api_block, torch, args.aux_paths, out["hbm_kernels"])
        if not args.no_cpu_baseline:
            # rank 0 only; with N > 1 the other ranks wait at the closing barrier (outside every timed region), so the
            # sample is a quarter of the single-GPU run's
            scale = 1 if world == 1 else 4
            out["cpu_baseline"] = guarded(cpu_baseline_block, params, args.cpu_threads, max(1, args.cpu_paths_per_thread // scale),
                                          max(1, args.cpu_single_thread_paths // scale), args.cpu_all_cores_seconds / scale)
        print(json.dumps(out), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
