#!/usr/bin/env python3
"""BASELINE configs[2] shape on one GPU: jorge.json (+ rho=0.3), N paths with trajectory bands +
histogram output.  Times each stage with HIP events on the launch stream and reports algorithmic
HBM bytes / time for the HBM-bound kernels (K1 trajectory writes, K3 quantiles, K2 histogram).

    python tools/bench_b3.py [--paths 10000000] [--reps 3]
"""

from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--paths", type=int, default=10_000_000)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--rho", type=float, default=0.3)
    args = ap.parse_args()
    import torch

    from monte_carlo_retirement_amd import Config, params_from_config
    from monte_carlo_retirement_amd import aggregation as A
    from monte_carlo_retirement_amd import engine as E

    cfg = Config(**dict(json.load(open(os.path.join(REPO, "scenarios", "jorge.json"))), equity_inflation_correlation=args.rho))
    p = params_from_config(cfg)
    n, wm = args.paths, 75
    b = E.DeviceBatch(p, wm, n, want="full")
    T, ry = b.sizes.trajectory_len, b.sizes.retirement_years

    def timed(fn):
        fn()                                  # first call: scratch allocation (15 GB at 1e8 paths)
        ts = []
        for _ in range(args.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record(); r = fn(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return float(np.median(ts)), r

    def k1():
        b.zero_counters()
        b.launch(12345, 1, 0)

    k1()
    ms_k1, _ = timed(k1)
    bytes_k1 = n * (8 * (2 * T + ry + 6) + 1)
    ms_q, (tq, _rq, wq, wc) = timed(lambda: A.band_quantiles(b, n))
    ms_qr = ms_qw = 0.0
    bytes_q_pass = 8 * n  # per row per pass
    ms_h, (bins, edges) = timed(lambda: A.success_histogram(b.summary["final_balance"], b.success, 100))
    ms_h60, _ = timed(lambda: A.success_histogram(b.summary["final_balance"], b.success, 60))
    out = {
        "workload": f"jorge.json rho={args.rho}, wm=75 (555 months), {n} paths, T={T}, ry={ry}",
        "K1_full_ms": ms_k1, "K1_paths_per_s": n / ms_k1 * 1e3,
        "K1_algorithmic_write_bytes": bytes_k1, "K1_write_GBps": bytes_k1 / ms_k1 / 1e6,
        "K3_traj_ms": ms_q, "K3_real_ms": ms_qr, "K3_wr_ms": ms_qw,
        "K3_rows": 2 * T + ry, "K3_total_ms": ms_q + ms_qr + ms_qw,
        "K3_slab_bytes": bytes_q_pass * (2 * T + ry),
        "K3_slab_reads_per_second_GBps": bytes_q_pass * (2 * T + ry) / (ms_q + ms_qr + ms_qw) / 1e6,
        "K3_fallback_rows": A.last_fallback_rows(),
        "K2_hist100_ms": ms_h, "K2_hist60_ms": ms_h60, "K2_algorithmic_bytes": 2 * 9 * n,
        "K2_GBps": 2 * 9 * n / ms_h / 1e6,
        "success_probability": float(b.counters[0].item()) / n,
        "median_final_nominal": float(tq[-1, 3]), "wr_year0_median": float(wq[0, 2]), "hist_total": int(bins.sum()),
        "end_to_end_ms": ms_k1 + ms_q + ms_qr + ms_qw + ms_h,
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
