"""Worker for tests/test_distributed_cpu.py: one rank of a gloo process group on the CPU.

The shard of each rank is simulated by the CPU ORACLE (test infrastructure standing in for the
HIP kernel, which needs a GPU); what is under test is the product's sharding + all-reduce layer
(monte_carlo_retirement_amd/distributed.py)."""

from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO)

from monte_carlo_retirement_amd import Config, params_from_config  # noqa: E402
from monte_carlo_retirement_amd import distributed as D  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    out_path, n_total, wm = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    with open(os.path.join(REPO, "scenarios", "jorge.json")) as fh:
        cfg = Config(**dict(json.load(fh), equity_inflation_correlation=0.3))
    params = params_from_config(cfg)
    shards = []

    def runner(begin, count):
        shards.append((begin, count))
        return O.run_batch(params, 777, 1, begin, count, wm, want_summary=False, want_trajectories=False)

    red = D.run_sharded_counts(n_total, cfg.retirement_years, runner)
    # histogram range exchange: rank-local (min, max) -> global
    mm = torch.tensor([10.0 * (rank + 1), 100.0 * (rank + 1)], dtype=torch.float64)
    D.all_reduce_minmax_(mm)
    # no seed configured: every rank must end up with rank 0's time-derived seed (one Philox key, one set of
    # sampled columns for the whole group)
    from monte_carlo_retirement_amd.simulation import RetirementMonteCarloSimulator

    import time
    time.sleep(0.01 * rank)   # different timestamps per rank
    unseeded = RetirementMonteCarloSimulator(Config(**dict(cfg.model_dump(by_alias=True), seed=None)))
    own = RetirementMonteCarloSimulator(Config(**dict(cfg.model_dump(by_alias=True), seed=99 + rank)))
    # distributed.local_only(): inside it this thread computes as if no group existed (bench.py's cross-check of the
    # candidate-split search); the state nests and is restored, other threads are unaffected
    import threading
    seen = {}
    with D.local_only():
        seen["inside"] = D.is_active()
        seen["broadcast_inside"] = D.broadcast_int(1000 + rank)          # no collective: every rank keeps its own value
        with D.local_only():
            seen["nested"] = D.is_active()
        seen["after_nested"] = D.is_active()
        t = threading.Thread(target=lambda: seen.__setitem__("other_thread", D.is_active()))
        t.start(); t.join()
    seen["after"] = D.is_active()
    seen["broadcast_after"] = D.broadcast_int(1000 + rank)               # rank 0's value everywhere again
    # the search's probes under a group (distributed.probe_candidates, what RetirementMonteCarloSimulator._probe_many runs):
    # small batches split by CANDIDATE month over the ranks (fewer candidates than ranks: the rest contribute zeros), large
    # ones by path range; the local "kernel" is the oracle here
    evaluated = []

    def oracle_probe(begin, count, months):
        evaluated.append((int(begin), int(count), [int(m) for m in months]))
        return np.array([[int(O.run_batch(params, 777, 0, begin, count, m, want_summary=False, want_trajectories=False)["counters"][0]), count]
                         for m in months], dtype=np.int64)

    probe_few = D.probe_candidates([20, 21, 22], 64, 10**6, oracle_probe).tolist()                 # 3 candidates: ranks 3.. idle
    probe_many = D.probe_candidates(list(range(10, 27)), 48, 10**6, oracle_probe).tolist()         # 17 candidates (the bench's search)
    probe_range = D.probe_candidates([20, 25], n_total, 50, oracle_probe).tolist()                 # by path range, ragged tail
    with open(f"{out_path}.{rank}", "w") as fh:
        json.dump({"main_seed": unseeded.main_seed, "own_seed": own.main_seed, "rank": rank, "world": world, "shards": shards, "success": red.success, "paths": red.paths,
                   "wr": red.wr_obs_counts.tolist(), "ruin": red.ruin_year_bins.tolist(),
                   "prob": red.success_probability_pct, "minmax": mm.tolist(), "active": D.is_active(), "local_only": seen,
                   "probe_few": probe_few, "probe_many": probe_many, "probe_range": probe_range, "probe_calls": evaluated}, fh)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
