#!/usr/bin/env python3
"""BASELINE configs[3] shape: the synthetic 60-year scenario S60 (config.json with
initial_balance 2.0e6, inv1 volatility 0.15, rho 0.3; wm=120 -> 720 months), success counts + histogram of
successful final balances, sharded by path range over the ranks of a process group (one all-reduce of the counter
block, one of the bins, plus the min/max pair when the edges are data-ranged).

    python tools/bench_b4.py --paths 100000000                       # 1 GPU, chunks of --chunk paths
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/bench_b4.py --paths 100000000
"""
import argparse, json, os, sys, time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--paths", type=int, default=10_000_000, help="global number of paths")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--bins", type=int, default=100)
    args = ap.parse_args()
    import torch
    import torch.distributed as dist

    from monte_carlo_retirement_amd import Config, params_from_config
    from monte_carlo_retirement_amd import distributed as D

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count()))
        dist.init_process_group(os.environ.get("MCR_BACKEND", "nccl"))
    with open(os.path.join(os.path.dirname(__file__), "..", "scenarios", "config.json")) as fh:
        cfg = Config(**dict(json.load(fh), initial_balance=2.0e6, inv1_returns_volatility=0.15,
                            equity_inflation_correlation=0.3, seed=12345))
    p = params_from_config(cfg)
    ts = []
    for _ in range(args.reps + 1):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = D.run_sharded_histogram(p, 12345, 1, args.paths, 120, n_bins=args.bins)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    best = min(ts[1:])
    if int(os.environ.get("RANK", "0")) == 0:
        c = out["counts"]
        print(json.dumps({
            "workload": "S60: 720-month paths, success counts + %d-bin histogram of successful final balances" % args.bins,
            "n_gpus": world, "paths": args.paths, "seconds": best, "paths_per_s": args.paths / best,
            "success_probability": c.success / c.paths, "hist_total": int(out["hist_bins"].sum()),
            "per_path_hbm_bytes": 49, "note": "includes buffer allocation, both all-reduces and the result download",
        }))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
