#!/usr/bin/env python3
"""CLI-shaped caller of the drop-in simulator (the call sequence of the reference's
backend/main.py:54-133 and backend/server.py:231-266,423-434): load a scenario JSON, search the
minimum working months, switch to the final seed stream, run the final batch, print a JSON summary.

    python examples/run_scenario.py scenarios/jorge.json --seed 12345 --rng numpy
    python examples/run_scenario.py scenarios/config.json --paths 1000000 --working-months 233

`--rng numpy` uses the reference's own NumPy stream (same seed -> the reference's numbers);
`--rng philox` (default) the engine's counter-based stream.  For large `--paths` the per-path
lists of the reference's API payload are replaced by device-side histograms."""

from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO)

from monte_carlo_retirement_amd import Config, load_config_from_json  # noqa: E402
from monte_carlo_retirement_amd.constants import MONTHS_PER_YEAR  # noqa: E402
from monte_carlo_retirement_amd.simulation import (  # noqa: E402
    RetirementMonteCarloSimulator,
    median_first_year_withdrawal_rate,
    retirement_age,
    trajectory_time_points,
)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--rng", choices=["philox", "numpy"], default="philox")
    ap.add_argument("--paths", type=int, default=None, help="override num_simulations_main")
    ap.add_argument("--search-paths", type=int, default=None, help="override num_simulations_search")
    ap.add_argument("--working-months", type=int, default=None, help="skip the search")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:  # one process per GPU (python -m torch.distributed.run --nproc-per-node N examples/run_scenario.py ...)
        import torch
        import torch.distributed as dist

        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count()))
        dist.init_process_group(os.environ.get("MCR_BACKEND", "nccl"))
    raw = load_config_from_json(args.config)
    if args.paths:
        raw["num_simulations_main"] = args.paths
    if args.search_paths:
        raw["num_simulations_search"] = args.search_paths
    config = Config(**raw)
    sim = RetirementMonteCarloSimulator(config, main_seed_override=args.seed, rng=args.rng)

    t0 = time.perf_counter()
    curve = []
    if args.working_months is None:
        months, achieved, curve = sim.find_minimum_working_months(verbose=False)
        if months == -1:
            print(json.dumps({"error": f"target {config.target_probability:.2f}% not reachable; best {achieved:.2f}%"}))
            return 1
    else:
        months = args.working_months
    t_search = time.perf_counter() - t0

    sim.use_final_seeds()
    t0 = time.perf_counter()
    summary_df, traj_pct, samples, wr_pct, real_pct, real_samples, wr_counts = sim.run_monte_carlo_simulations(
        working_months=months, num_simulations=config.num_simulations_main
    )
    t_final = time.perf_counter() - t0

    ok = summary_df["Success"].astype(bool)
    successful = summary_df.loc[ok, "Final Balance"]
    years = trajectory_time_points(months, config.retirement_years)
    out = {
        "scenario": config.Nickname,
        "rng": args.rng,
        "main_seed": sim.main_seed,
        "required_working_months": months,
        "required_working_years": round(months / MONTHS_PER_YEAR, 1),
        "retirement_age": round(retirement_age(config.current_age, months), 1),
        "search_probes": len(curve),
        "success_probability": round(float(sim._success_probability(summary_df)), 2),
        "median_start_balance": round(float(summary_df["Start Balance"].median()), 2),
        "median_final_balance_successful": round(float(successful.median()), 2) if not successful.empty else 0.0,
        "swr": None if math.isnan(median_first_year_withdrawal_rate(summary_df)) else round(median_first_year_withdrawal_rate(summary_df), 3),
        "final_balance_percentiles": {f"p{int(q * 100)}": round(max(0.0, float(v)), 2) for q, v in
                                      summary_df["Final Balance"].quantile([0.05, 0.25, 0.5, 0.75, 0.95]).items()},
        "trajectory_years": len(years),
        "median_trajectory_end": round(float(traj_pct[0.50].iloc[-1]), 2),
        "median_real_trajectory_end": round(float(real_pct[0.50].iloc[-1]), 2),
        "wr_median_year0": round(float(wr_pct[0.50].iloc[0]), 3),
        "wr_observation_counts_first_last": [wr_counts[0], wr_counts[-1]],
        "failure_count": int((~ok).sum()),
        "sample_paths": len(samples or []),
        "seconds": {"search": round(t_search, 3), "final_run": round(t_final, 3)},
    }
    assert len(years) == len(traj_pct) == len(real_pct)
    if int(os.environ.get("RANK", "0")) == 0:
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
