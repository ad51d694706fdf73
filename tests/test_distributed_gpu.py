"""Multi-rank paths on real kernels: 2 and 3 ranks of a gloo group share the box's single GPU
(the 8-GPU RCCL run is the driver's).  Distributed exact quantiles and sharded bands must be
bit-identical to the unsharded computation."""

from __future__ import annotations

import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_quantiles_and_bands_equal_unsharded(tmp_path, world):
    out = str(tmp_path / "res")
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                   LOCAL_RANK=str(rank), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "dist_gpu_worker.py"), out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        stdout, _ = p.communicate(timeout=600)
        assert p.returncode == 0, stdout.decode()[-3000:]
    res = [json.load(open(f"{out}.{r}")) for r in range(world)]
    for r in res:
        assert r["quantiles_equal"] and r["counts_equal"], r
        assert r["bracket_route_taken"] and r["bracket_quantiles_equal"] and r["bracket_counts_equal"], {k: v for k, v in r.items() if k.startswith("bracket")}
        assert r["bracket_at_threshold"] and r["radix_below_threshold"], r
        assert r["bracket_mid_equal"] and r["bracket_mid_counts"] and 1 <= r["bracket_fallback_rows"] <= 5, {k: v for k, v in r.items() if k.startswith("bracket")}
        assert r["bracket_fallback_rows"] == res[0]["bracket_fallback_rows"]
        assert r["fuzz_bad"] == [] and r["fuzz_bracket_calls"] >= 1, (r["fuzz_bad"], r["fuzz_bracket_calls"])
        assert r["bands_equal"] and r["hist_equal"] and r["counts_ok"], r
        assert r["success"] == res[0]["success"]
        assert r["hist_only_equal"] and r["hist_fixed_equal"], r
        assert r["class_summary_equal"] and r["class_bands_equal"] and r["class_samples_equal"] and r["class_probe_equal"], r
        assert r["class_probe_many_by_candidate"] and r["class_probe_many_by_range"] and r["class_speculation_slots"], r
        assert r["class_search_replays_reference"] and r["class_search_batched"], r
        assert r["class_replicated_equals_sharded"], r
        # large-n guard of the sharded class API: no n-sized host frame on the ranks other than 0
        assert r["guard_frame_rows"] == (5003 if r["rank"] == 0 else 0) and r["guard_frame_columns_ok"], r
        assert r["guard_rank0_frame_equal"] and r["guard_rest_equal"], r
        assert r["unseeded_samples_equal"] and r["unseeded_seed"] == res[0]["unseeded_seed"], (r["unseeded_seed"], res[0]["unseeded_seed"])
    # the compact document built from sharded batches equals the single-process one (built here, same scenario)
    from monte_carlo_retirement_amd import Config
    from monte_carlo_retirement_amd import results as R
    from monte_carlo_retirement_amd.simulation import RetirementMonteCarloSimulator

    with open(os.path.join(REPO, "scenarios", "jorge.json")) as fh:
        cfg = Config(**dict(json.load(fh), equity_inflation_correlation=0.3, initial_balance=20000.0, monthly_contribution=3000.0))
    sim = RetirementMonteCarloSimulator(cfg, main_seed_override=2024)
    sim.use_final_seeds()
    whole = json.loads(json.dumps(R.compact_result(cfg, sim, 40, num_simulations=5003)))
    for r in res:
        assert r["compact_doc"] == whole, [k for k in whole if r["compact_doc"].get(k) != whole[k]]
