"""N > 1 path of the engine on CPU: world_size-2 (and 3) gloo process groups.  Shards tile the
global path range, the reduced counters equal a single-process run of the whole range, every
rank ends with the same numbers."""

from __future__ import annotations

import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO
from monte_carlo_retirement_amd import Config, params_from_config
from monte_carlo_retirement_amd import distributed as D


def test_shard_range_tiles_the_path_range():
    for n in (0, 1, 7, 64, 1000, 10**8 + 3):
        for world in (1, 2, 3, 8):
            shards = [D.shard_range(n, r, world) for r in range(world)]
            assert sum(c for _, c in shards) == n
            pos = 0
            for b, c in shards:
                assert b == pos or c == 0
                pos += c
            assert max(c for _, c in shards) - min(c for _, c in shards if True) <= -(-n // world)
    with pytest.raises(ValueError):
        D.shard_range(10, 2, 2)


def test_pack_unpack_roundtrip():
    v = D.pack_counts([5, 9], np.arange(3), np.arange(5) * 2)
    r = D.unpack_counts(v, 3)
    assert (r.success, r.paths) == (5, 9) and r.wr_obs_counts.tolist() == [0, 1, 2] and r.ruin_year_bins.tolist() == [0, 2, 4, 6, 8]
    assert r.success_probability_pct == pytest.approx(5 / 9 * 100)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,n_total", [(2, 301), (3, 64), (8, 1003)])
def test_sharded_counts_equal_single_process(tmp_path, oracle, world, n_total):
    """(8 ranks = the world size the driver's scaling run uses: ragged shards 126 x 7 + 121, candidates split 8 ways.)"""
    wm = 30
    out = str(tmp_path / "res")
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                   WORLD_SIZE=str(world), LOCAL_RANK=str(rank), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "dist_worker.py"), out, str(n_total), str(wm)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        stdout, _ = p.communicate(timeout=240)
        assert p.returncode == 0, stdout.decode()[-2000:]
    res = [json.load(open(f"{out}.{r}")) for r in range(world)]
    with open(os.path.join(REPO, "scenarios", "jorge.json")) as fh:
        cfg = Config(**dict(json.load(fh), equity_inflation_correlation=0.3))
    whole = oracle.run_batch(params_from_config(cfg), 777, 1, 0, n_total, wm, want_summary=False, want_trajectories=False)
    covered = sorted((b, c) for r in res for b, c in r["shards"])
    assert sum(c for _, c in covered) == n_total and covered[0][0] == 0
    for r in res:
        assert r["active"] and r["world"] == world
        assert r["success"] == int(whole["counters"][0]) and r["paths"] == n_total
        assert r["wr"] == whole["wr_obs_counts"].tolist() and r["ruin"] == whole["ruin_year_bins"].tolist()
        assert r["prob"] == res[0]["prob"]
        assert r["minmax"] == [10.0, 100.0 * world]
        assert r["main_seed"] == res[0]["main_seed"] and r["own_seed"] == 99 + r["rank"]   # seed=None: rank 0's seed everywhere
        lo = r["local_only"]
        assert (lo["inside"], lo["nested"], lo["after_nested"], lo["other_thread"], lo["after"]) == (False, False, False, True, True), lo
        assert lo["broadcast_inside"] == 1000 + r["rank"] and lo["broadcast_after"] == 1000
    # the search's probes: every rank ends with the single-process counts, whatever the split
    p = params_from_config(cfg)

    def single(months, n):
        return [[int(oracle.run_batch(p, 777, 0, 0, n, m, want_summary=False, want_trajectories=False)["counters"][0]), n] for m in months]

    few, many, ranged = single([20, 21, 22], 64), single(list(range(10, 27)), 48), single([20, 25], n_total)
    for r in res:
        assert r["probe_few"] == few and r["probe_many"] == many and r["probe_range"] == ranged, r["rank"]
        calls = r["probe_calls"]
        # by candidate: rank r evaluates months[r::world] over the WHOLE range, or nothing at all
        mine_few, mine_many = [20, 21, 22][r["rank"]::world], list(range(10, 27))[r["rank"]::world]
        expected = ([[0, 64, mine_few]] if mine_few else []) + ([[0, 48, mine_many]] if mine_many else [])
        begin, count = D.shard_range(n_total, r["rank"], world)
        expected += [[begin, count, [20, 25]]] if count else []
        assert calls == expected, (r["rank"], calls, expected)
    if world == 8:
        assert sum(1 for r in res if not [20, 21, 22][r["rank"]::world]) == 5      # five ranks sat the 3-candidate round out
