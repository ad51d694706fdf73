#!/usr/bin/env python3
"""Kernel iteration tool (GPU box): time the path kernel and check it against the oracle.

    python tests/kbench.py [--paths 1000000] [--reps 5] [--check 20000]
Prints ms/launch for the count-only and full-output variants on the config.json (C1) and
jorge.json (C3) workloads, the success counts, and the parity of a `--check`-path batch vs the
CPU oracle (flipped flags, max relative error of the summary fields)."""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))  # tests/ -> repo root
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--paths", type=int, default=1_000_000)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--check", type=int, default=20_000)
    ap.add_argument("--full", action="store_true", help="also time the trajectory-writing variant")
    args = ap.parse_args()
    import torch

    from monte_carlo_retirement_amd import Config, params_from_config
    from monte_carlo_retirement_amd import engine as E
    from oracle import oracle as O

    out = {}
    for name, fname, wm, over in [("C1", "config.json", 233, {}), ("C3", "jorge.json", 75, {"equity_inflation_correlation": 0.3})]:
        cfg = Config(**dict(json.load(open(os.path.join(REPO, "scenarios", fname))), **over))
        p = params_from_config(cfg)
        for want in (["count", "full"] if args.full else ["count"]):
            n = args.paths if want == "count" else min(args.paths, 2_000_000)
            b = E.DeviceBatch(p, wm, n, want=want)
            b.launch(12345, 1, 0)
            torch.cuda.synchronize()
            ts = []
            for r in range(args.reps):
                b.zero_counters()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); b.launch(12345, 1, 0); e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            ms = float(np.median(ts))
            out[f"{name}_{want}"] = {"ms": ms, "paths_per_s": n / ms * 1e3, "success": int(b.counters[0].item()), "n": n}
        if args.check:
            g = E.run_batch_host(p, 12345, 1, 0, args.check, wm)
            c = O.run_batch(p, 12345, 1, 0, args.check, wm)
            flips = int((g["success"] != c["success"]).sum())
            worst = 0.0
            for k in E.SUMMARY_FIELDS + ("trajectory", "real_trajectory", "withdrawal_rate_trajectory"):
                a, bb = np.asarray(g[k], dtype=float), np.asarray(c[k], dtype=float)
                m = ~(np.isnan(a) & np.isnan(bb))
                worst = max(worst, float(np.max(np.abs(a[m] - bb[m]) / (np.abs(bb[m]) + 1e-6))) if m.any() else 0.0)
            out[f"{name}_check"] = {"n": args.check, "flips": flips, "max_rel_err": worst}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
