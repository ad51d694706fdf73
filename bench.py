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

    def timed(fn, reps=5):
        fn()                              # first call: scratch allocation, cold caches
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
    profiled = None   # the slab pass alone, from the committed rocprofv3 summary of this same command (latest round)
    for rnd in ("r02", "r01_final"):
        try:
            with open(os.path.join(REPO, "profiles", rnd, "pmc_summary.json")) as fh:
                dv = json.load(fh)["derived"]
            profiled = {"kernel": "mcr::rq_count_kernel<16, true>" if rnd != "r01_final" else "mcr::rq_bracket_kernel",
                        "ms": dv["K3_bracket_avg_ms_from_kernel_stats"], "achieved_TBps": dv["K3_bracket_achieved_TBps"],
                        "frac_of_hbm_peak": dv["K3_bracket_achieved_TBps"] * 1e3 / HBM_PEAK_GBS,
                        "traffic_over_algorithmic": dv["K3_bracket_traffic_over_algorithmic"],
                        "paths": 10_000_000 if rnd != "r01_final" else 4_000_000,
                        "source": f"profiles/{rnd}/pmc_summary.json (rocprofv3 --kernel-trace --stats + FETCH_SIZE/WRITE_SIZE passes)"}
            break
        except (OSError, KeyError, ValueError):
            continue
    return {
        "workload": f"BASELINE configs[2] shape: jorge.json rho=0.3, wm=75 (555 months), {n} paths, T={T}, ry={ry}",
        "K1_full_output": {"ms": ms_k1, "paths_per_s": n / ms_k1 * 1e3, "algorithmic_write_bytes": bytes_k1,
                           "write_GBps": bytes_k1 / ms_k1 / 1e6, "frac_of_hbm_peak": bytes_k1 / ms_k1 / 1e6 / HBM_PEAK_GBS,
                           "note": "compute-bound: the time-major trajectory stores hide under the fp64 VALU work"},
        "K3_row_quantiles": {"ms": ms_k3, "rows": 2 * T + ry, "algorithmic_read_bytes": bytes_k3,
                             "GBps": bytes_k3 / ms_k3 / 1e6, "frac_of_hbm_peak": bytes_k3 / ms_k3 / 1e6 / HBM_PEAK_GBS,
                             "fallback_rows": fallback_rows, "slab_pass_profiled": profiled,
                             "note": "one call over the [2T+ry] slab (bands of all rows): six launches + one word read back, incl. the result "
                                     "download. Algorithmic bytes = ONE read of the slab; the counting pass that does it runs at ~5 TB/s, the "
                                     "rest is the two sampling steps before it and the per-row cell selection after it (profiles/). "
                                     "fallback_rows = rows that needed the 4-pass radix select (-1: rows too short for the bracketed route)"},
        "K2_histogram": {"ms": ms_k2, "algorithmic_bytes": bytes_k2, "GBps": bytes_k2 / ms_k2 / 1e6},
    }


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


def s60_block(torch, dist, world, n_total, reps=3, grouped=None):
    """North-star shape (BASELINE configs[3]): S60 = config.json with initial_balance=2e6,
    inv1 volatility 0.15, rho=0.3, wm=120 (720-month paths); success counts + 100-bin histogram of the
    successful final balances over `n_total` paths IN TOTAL, sharded by global path range over the ranks
    (`distributed.run_sharded_histogram`: counter/bin all-reduce + a min/max all-reduce for the range).
    End to end per repetition: buffer allocation, summary-output kernel (49 B/path), min/max + bins,
    collectives, result download.  Not part of `value`."""
    from monte_carlo_retirement_amd import Config, params_from_config
    from monte_carlo_retirement_amd import distributed as D

    with open(os.path.join(REPO, "scenarios", "config.json")) as fh:
        cfg = Config(**dict(json.load(fh), initial_balance=2.0e6, inv1_returns_volatility=0.15,
                            equity_inflation_correlation=0.3, seed=12345))
    p = params_from_config(cfg)
    grouped = world > 1 if grouped is None else grouped
    times, r = [], None
    for _ in range(reps):
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = D.run_sharded_histogram(p, 12345, 1, n_total, 120, n_bins=100)
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
                    "histogram of successful final balances",
        "n_paths_total": n_total, "n_gpus": world, "seconds": sec, "paths_per_s": n_total / sec,
        "success_probability": c.success / max(1, c.paths), "paths_counted": c.paths,
        "hist_total": int(r["hist_bins"].sum()),
        "exchange": "none (1 GPU)" if world == 1 else "all-reduce(sum) of the counter/bin vectors + all-reduce(min,max) of the histogram range",
        "note": "end to end incl. allocation and download; median of %d repetitions, max over ranks" % reps,
    }


def launch_ranks(n_ranks: int) -> int:
    """`python bench.py --gpus N` without an outer launcher: start N copies of this script, one rank per GPU
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment, as torch.distributed.run would), wait, and
    return non-zero if any rank failed.  The parent never imports torch or touches a GPU; children are plain
    child processes (nothing is exec'ed over a process that has initialised the GPU).  Rank 0 inherits stdout,
    so its JSON line is this command's JSON line."""
    import socket
    import subprocess

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc, pending, deadline = 0, set(range(n_ranks)), None
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
        if deadline is not None and time.monotonic() > deadline:
            for r in pending:
                procs[r].kill()  # exact PIDs this launcher started
            for r in pending:
                procs[r].wait()
            break
        time.sleep(0.05)
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
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--cpu-paths-per-thread", type=int, default=160_000)
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
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(args.backend)
    comm_dev = torch.device("cuda", device) if (not grouped or args.backend == "nccl") else torch.device("cpu")

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

    s60 = None
    if not args.no_s60:
        try:
            s60 = s60_block(torch, dist, world, args.s60_paths, grouped=grouped)
        except Exception as exc:  # never lose the headline line to the auxiliary block
            s60 = {"error": f"{type(exc).__name__}: {exc}"}

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
                "parallelism": f"path-range sharding x{world}" + (
                    f" + 1 all-reduce(sum) of the {exch.numel()}-word counter/bin vector per step ({args.backend}), overlapped with the next step's compute" if grouped else ""),
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
            "paths_counted": counters[1],
        }
        try:
            out["accuracy_10k"] = accuracy_10k()
        except Exception as exc:  # never lose the headline line to an auxiliary block
            out["accuracy_10k"] = {"error": f"{type(exc).__name__}: {exc}"}
        if s60 is not None:
            out["s60"] = s60
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
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
