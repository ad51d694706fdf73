"""rng="numpy": the reference's OWN random stream on the device.  Literal seed parity: the same
`seed` reproduces the reference's native (un-injected) results — shocks bit-for-bit, per-path
results within the path tolerance, Success flags and the search identical."""

from __future__ import annotations

import os

import numpy as np
import pytest

from conftest import compare_batch_to_golden, load_golden
from monte_carlo_retirement_amd import Config, params_from_config
from monte_carlo_retirement_amd import _native as N
from monte_carlo_retirement_amd import engine as E
from monte_carlo_retirement_amd.simulation import RetirementMonteCarloSimulator

pytestmark = pytest.mark.gpu
REL, ABS = 1e-9, 1e-6


def test_shock_rows_equal_numpy_bit_for_bit():
    """default_rng(seed).standard_normal((n, 3)) + rho mix (simulation.py:452-466), on the device."""
    seeds = np.array([0, 1, 42, 3735928559, 2**32 - 1, 777], dtype=np.uint32)
    extra = int(os.environ.get("MCR_NPRNG_FUZZ_SEEDS", "0"))    # soak by hand: that many more random seeds (1.2e4 draws each)
    if extra:
        seeds = np.concatenate([seeds, np.random.default_rng(int(os.environ.get("MCR_NPRNG_FUZZ_SEED", "1"))).integers(0, 2**32, extra, dtype=np.uint64).astype(np.uint32)])
    n_months, rho = 2000, 0.3
    got = E.draw_shocks_host(N.numpy_rng(0), 1, 0, len(seeds), n_months, rho, path_seeds=seeds)
    exact = total = 0
    for i, s in enumerate(seeds):
        ind = np.random.default_rng(int(s)).standard_normal((n_months, 3))
        exp = np.column_stack((ind[:, 0], rho * ind[:, 0] + np.sqrt(max(0.0, 1.0 - rho * rho)) * ind[:, 1], ind[:, 2]))
        np.testing.assert_allclose(got[i], exp, rtol=0, atol=2e-15)   # tail draws go through device log1p: <= 1-2 ulp
        exact += int((got[i].view(np.uint64) == exp.view(np.uint64)).sum())
        total += exp.size
    assert exact / total > 0.9995, exact / total                          # everything but a few tail values is bit-identical


def test_seed_derivation_matches_seedsequence_spawn():
    """Kernel-side SeedSequence(main).spawn(2)[s].spawn(..)[j].generate_state(1) == NumPy's."""
    for main_seed in (12345, 0, 2**40 + 17, 2**100 + 3):
        for stream, off in ((0, 0), (1, 7), (1, 2**32 - 2)):
            n = 5
            derived = E.draw_shocks_host(N.numpy_rng(main_seed, child_offset=off), stream, 0, n, 6, 0.0)
            kids = [np.random.SeedSequence(main_seed, spawn_key=(stream, off + j)) for j in range(n)]
            seeds = np.array([int(k.generate_state(1)[0]) for k in kids], dtype=np.uint32)
            explicit = E.draw_shocks_host(N.numpy_rng(0), stream, 0, n, 6, 0.0, path_seeds=seeds)
            assert np.array_equal(derived, explicit), (main_seed, stream, off)


def test_native_paths_match_reference():
    """Per-path results of the reference run with its own RNG (no injection), reproduced from the
    seed alone — including the spawn-offset rule of _path_seeds (an earlier batch of another size)."""
    for g in load_golden("numpy_native_batch.json")["paths"]:
        sim = RetirementMonteCarloSimulator(Config(**g["cfg"]), main_seed_override=g["main_seed"], rng="numpy")
        getattr(sim, f"use_{g['stream']}_seeds")()
        assert sim._path_seeds(g["earlier_batch"]) == g["earlier_seeds"]
        n = len(g["results"])
        assert sim._path_seeds(n) == g["numpy_path_seeds"]
        # (a) single-path API with the reference's uint32 seeds
        for s, exp in list(zip(g["numpy_path_seeds"], g["results"]))[:4]:
            r = sim._run_single_simulation_path(g["working_months"], s)
            assert r["Success"] == exp["Success"]
            np.testing.assert_allclose(r["Trajectory"], exp["Trajectory"], rtol=REL, atol=ABS)
        # (b) batch API: seeds derived in-kernel from (main_seed, stream, spawn offset)
        res = E.run_batch_host(params_from_config(Config(**g["cfg"])), sim._batch_rng(n), sim._stream_id, 0, n, g["working_months"])
        try:
            compare_batch_to_golden(res, g["results"], exact=False, rel=REL, abs_tol=ABS)
        except AssertionError as e:
            raise AssertionError(f"{g['name']}/{g['stream']}: {e}") from e


def test_native_search_and_final_run_match_reference():
    """find_minimum_working_months + the final run with literally the reference's seed."""
    g = load_golden("numpy_native_batch.json")["search"][0]
    sim = RetirementMonteCarloSimulator(Config(**g["cfg"]), main_seed_override=g["seed"], rng="numpy")
    events = []
    months, prob, curve = sim.find_minimum_working_months(verbose=False, progress_callback=events.append)
    assert (months, prob) == (g["months"], g["probability"])
    assert curve == g["search_curve"] and events == g["events"]
    sim.use_final_seeds()
    summary, traj, _, _, _, _, wr_counts = sim.run_monte_carlo_simulations(months, g["cfg"]["num_simulations_main"])
    assert summary["Success"].tolist() == g["final_success"]
    assert sim._success_probability(summary) == g["final_success_probability"]
    np.testing.assert_allclose(traj.to_numpy(), np.array(g["final_trajectory_percentiles"]["values"]), rtol=REL, atol=ABS)
    assert wr_counts == g["final_wr_observation_counts"]


def test_native_success_rate_equals_reference_at_4000_paths():
    """numpy_native_stats.json: the reference's native 4000-path runs; same seed -> same count."""
    for g in load_golden("numpy_native_stats.json"):
        n = g["n_paths"]
        p = params_from_config(Config(**g["cfg"]))
        res = E.run_batch_host(p, N.numpy_rng(g["main_seed"], child_offset=0), N.MCR_STREAM_FINAL, 0, n, g["working_months"],
                               want_trajectories=False)
        assert int(res["counters"][0]) == round(g["success_probability_pct"] * n / 100.0), g["name"]
        assert float(np.median(res["final_balance"])) == pytest.approx(g["median_final_balance"], rel=REL)
        assert float(np.median(res["start_balance"])) == pytest.approx(g["median_start_balance"], rel=REL)


def test_cli_flow_reproduces_the_references_surveyed_numbers():
    """BASELINE.md §2 / SURVEY §6, measured on the REFERENCE with seed 12345: jorge.json ->
    33 probes, 75 months @ 99.0 %, final N=1000 98.5 %; config.json (num_simulations_search=300) ->
    29 probes, 233 months @ 98.0 %, final 97.9 %.  The CLI-shaped caller with rng=numpy must print
    exactly those numbers."""
    import json
    import os
    import subprocess
    import sys

    from conftest import REPO

    def run(cfg, *extra):
        out = subprocess.run([sys.executable, os.path.join(REPO, "examples", "run_scenario.py"),
                              os.path.join(REPO, "scenarios", cfg), "--seed", "12345", "--rng", "numpy", *extra],
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads(out.stdout.strip().splitlines()[-1])

    j = run("jorge.json")
    assert (j["summary"]["required_working_months"], j["search_probes"], j["summary"]["success_probability"]) == (75, 33, 98.5)
    c = run("config.json")
    assert (c["summary"]["required_working_months"], c["search_probes"], c["summary"]["success_probability"]) == (233, 29, 97.9)
    assert c["trajectory_points"] == 71 and c["sample_paths"] == 5
    # the device-aggregated document reports the same summary block
    k = run("config.json", "--compact")
    assert k["summary"] == c["summary"] and k["search_probes"] == 29 and k["document_bytes"] < c["document_bytes"]
