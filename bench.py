#!/usr/bin/env python3
"""Headline benchmark: paths/sec of the per-path Monte Carlo kernel (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic paths: ONE launch of the path
kernel over `--paths` (default 1e6) paths per GPU of the `config.json` scenario at
working_months=233 (833 months/path), success-count only (BASELINE.json configs[1]), followed —
when N > 1 — by the single all-reduce of the counter vector (RCCL).  Independent path ranges
shard across ranks by GLOBAL path index (weak scaling: per-GPU work is fixed).

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel: algorithmic fp64 ops / measured
kernel time vs the fp64 vector-issue peak) and `cpu_baseline` (the CPU oracle timed on this box's
host cores on a bounded sample of the same workload).
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
# The path has no FMAs by construction (-ffp-contract=off, reference rounding), so one algorithmic
# op = one lane-instruction = 1 flop: the applicable issue ceiling is 39.3 T fp64 lane-ops/s.
FP64_LANE_OPS_PEAK_T = 39.3
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


def aux_hbm_kernels(torch, n):
    """The HBM-side kernels of the path on the BASELINE configs[2] shape (jorge.json + rho=0.3, wm=75:
    T=48 yearly samples, 40 WR rows), n paths: K1 with full trajectory output (write efficiency), K3
    row quantiles and K2 histogram (achieved algorithmic GB/s vs the 8 TB/s HBM peak).  Not part of
    `value`; reported so the memory-bound side of the path has a measured roofline too."""
    from monte_carlo_retirement_amd import Config, params_from_config
    from monte_carlo_retirement_amd import aggregation as A
    from monte_carlo_retirement_amd import engine as E

    with open(os.path.join(REPO, "scenarios", "jorge.json")) as fh:
        cfg = Config(**dict(json.load(fh), equity_inflation_correlation=0.3, seed=12345))
    p = params_from_config(cfg)
    b = E.DeviceBatch(p, 75, n, want="full")
    T, ry = b.sizes.trajectory_len, b.sizes.retirement_years

    def timed(fn, reps=3):
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
    ms_k3 = timed(lambda: A.band_quantiles(b, n))
    fallback_rows = A.last_fallback_rows()
    bytes_k3 = 8 * n * (2 * T + ry)      # algorithmic: every entry of the slab has to be read once
    ms_k2 = timed(lambda: A.success_histogram(b.summary["final_balance"], b.success, 100))
    bytes_k2 = 2 * 9 * n                 # min/max pass + bin pass, 8 B value + 1 B flag each
    profiled = None   # the slab pass alone, from the committed rocprofv3 summary of this same command
    try:
        with open(os.path.join(REPO, "profiles", "r01_final", "pmc_summary.json")) as fh:
            dv = json.load(fh)["derived"]
        profiled = {"kernel": "mcr::rq_bracket_kernel", "ms": dv["K3_bracket_avg_ms_from_kernel_stats"],
                    "achieved_TBps": dv["K3_bracket_achieved_TBps"], "frac_of_hbm_peak": dv["K3_bracket_achieved_TBps"] * 1e3 / HBM_PEAK_GBS,
                    "traffic_over_algorithmic": dv["K3_bracket_traffic_over_algorithmic"],
                    "source": "profiles/r01_final/pmc_summary.json (rocprofv3 --kernel-trace --stats + FETCH_SIZE/WRITE_SIZE passes)"}
    except (OSError, KeyError, ValueError):
        pass
    return {
        "workload": f"BASELINE configs[2] shape: jorge.json rho=0.3, wm=75 (555 months), {n} paths, T={T}, ry={ry}",
        "K1_full_output": {"ms": ms_k1, "paths_per_s": n / ms_k1 * 1e3, "algorithmic_write_bytes": bytes_k1,
                           "write_GBps": bytes_k1 / ms_k1 / 1e6, "frac_of_hbm_peak": bytes_k1 / ms_k1 / 1e6 / HBM_PEAK_GBS,
                           "note": "compute-bound: the time-major trajectory stores hide under the fp64 VALU work"},
        "K3_row_quantiles": {"ms": ms_k3, "rows": 2 * T + ry, "algorithmic_read_bytes": bytes_k3,
                             "GBps": bytes_k3 / ms_k3 / 1e6, "frac_of_hbm_peak": bytes_k3 / ms_k3 / 1e6 / HBM_PEAK_GBS,
                             "fallback_rows": fallback_rows, "slab_pass_profiled": profiled,
                             "note": "one call over the [2T+ry] slab (bands of all rows), incl. scratch allocation and result download. "
                                     "Algorithmic bytes = ONE read of the slab; the bracket pass that does it runs at ~5 TB/s, the rest of "
                                     "the time is the sample select before it and the candidate select after it (profiles/). "
                                     "fallback_rows = rows that needed the 4-pass radix select (-1: rows too short for the bracketed route)"},
        "K2_histogram": {"ms": ms_k2, "algorithmic_bytes": bytes_k2, "GBps": bytes_k2 / ms_k2 / 1e6},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--paths", type=int, default=1_000_000, help="paths per GPU per step")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse on one GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-aux", action="store_true", help="skip the HBM-side kernels (trajectory write / quantiles / histogram)")
    ap.add_argument("--aux-paths", type=int, default=4_000_000)
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--cpu-paths-per-thread", type=int, default=160_000)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from monte_carlo_retirement_amd import Config, params_from_config
    from monte_carlo_retirement_amd import engine as E

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    device = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(device)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(args.backend)

    with open(os.path.join(REPO, "scenarios", "config.json")) as fh:
        cfg = Config(**dict(json.load(fh), seed=12345))
    params = params_from_config(cfg)
    n = args.paths
    batch = E.DeviceBatch(params, WORKING_MONTHS, n, want="count", device=device)

    def step(i):
        # global path index: step-major, then rank (every path of the job is distinct)
        begin = (i * world + rank) * n
        batch.launch(12345, 1, begin)
        if world > 1:
            dist.all_reduce(batch.reduce_vec)  # the path's single exchange step: counters + year bins, summed

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        batch.zero_counters()
        step(i)
    fence()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    batch.zero_counters()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()
        batch.launch(12345, 1, ((args.warmup + i) * world + rank) * n)
        ev[i][1].record()
        if world > 1:
            dist.all_reduce(batch.reduce_vec)
    fence()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    kern_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps  # HIP events on the launch stream
    counters = batch.counters.cpu().tolist()

    if rank == 0:
        total_paths = n * world * args.steps
        value = total_paths / dt
        achieved_t = ALGO_OPS_PER_PATH * n / (kern_ms * 1e-3) / 1e12
        traffic = None
        pmc = os.path.join(REPO, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            with open(pmc) as fh:
                traffic = json.load(fh).get("path_kernel_count_only_bytes_per_launch")
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
                "parallelism": f"path-range sharding x{world}" + (" + 1 all-reduce(sum) of the counter/bin vector per step" if world > 1 else ""),
            },
            "roofline": {
                "kernel": "mcr::path_kernel<0, 0, true, false>  (MODE 0 count-only, Philox, realized-gains tax, no annual tax)",
                "bound": "valu_fp64",
                "achieved": achieved_t,
                "peak": FP64_LANE_OPS_PEAK_T,
                "unit": "TFLOP/s",
                "frac": achieved_t / FP64_LANE_OPS_PEAK_T,
                "traffic": traffic,
                "kernel_ms": kern_ms,
                "algorithmic_ops_per_path": ALGO_OPS_PER_PATH,
                "note": "no dense contraction and ~0 HBM bytes/path in this variant: the bound is fp64 VALU "
                        "issue (SURVEY 8d). peak = 78.6 TFLOP/s spec / 2 (the path has no FMAs: 1 op = 1 flop)",
            },
            "success_probability": counters[0] / max(1, counters[1]),
        }
        if not args.no_aux and world == 1:
            out["hbm_kernels"] = aux_hbm_kernels(torch, args.aux_paths)
        if not args.no_cpu_baseline and world == 1:
            v, secs = cpu_baseline(params, args.cpu_threads, args.cpu_paths_per_thread)
            out["cpu_baseline"] = {
                "value": v,
                "unit": "paths/s",
                "cores": args.cpu_threads,
                "kind": "port",
                "sample": f"{args.cpu_threads} threads x {args.cpu_paths_per_thread} paths of the same workload "
                          f"(oracle/mcr_oracle.c, scalar fp64, {secs:.1f} s wall)",
            }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
