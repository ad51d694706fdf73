"""The CPU oracle (oracle/mcr_oracle.c) against the golden vectors produced by the REFERENCE.

These are the tests that pin the oracle: every comparison here is BIT-EXACT (the oracle is
compiled without FMA contraction and uses the same glibc exp as CPython)."""

from __future__ import annotations

import os

import numpy as np
import pytest

from conftest import GOLDEN, STREAM_ID, assert_same_float, compare_batch_to_golden, load_golden
from monte_carlo_retirement_amd import Config, params_from_config


def _params(cfgd):
    return params_from_config(Config(**cfgd))


# ---- Philox known-answer vectors (Random123 kat_vectors, philox4x32-10) ------------------
@pytest.mark.parametrize(
    "ctr,key,exp",
    [
        ([0, 0, 0, 0], [0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
        ([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
        ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0],
         [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]),
    ],
)
def test_philox_known_answers(oracle, ctr, key, exp):
    assert oracle.philox4x32_10(ctr, key) == exp


def test_shock_rows_are_prefix_stable_and_standard_normal(oracle):
    """Row k does not depend on n (common random numbers, simulation.py:452-466 / SURVEY a5)."""
    a = oracle.draw_shocks(12345, 1, 7, 50, 0.3)
    b = oracle.draw_shocks(12345, 1, 7, 900, 0.3)
    assert np.array_equal(a, b[:50])
    big = np.concatenate([oracle.draw_shocks(99, 0, p, 1200, 0.0) for p in range(40)])
    assert abs(big.mean()) < 0.02 and abs(big.std() - 1.0) < 0.02
    c = np.corrcoef(big.T)
    assert np.abs(c - np.eye(3)).max() < 0.02
    pos = oracle.draw_shocks(5, 1, 4, 100, 1.0)
    neg = oracle.draw_shocks(5, 1, 4, 100, -1.0)
    assert np.array_equal(pos[:, 1], pos[:, 0]) and np.array_equal(neg[:, 1], -neg[:, 0])
    mid = np.concatenate([oracle.draw_shocks(3, 1, p, 1200, 0.3) for p in range(40)])
    assert abs(np.corrcoef(mid[:, 0], mid[:, 1])[0, 1] - 0.3) < 0.02


# ---- scalar helpers ----------------------------------------------------------------------
def test_helpers_match_reference_exactly(oracle):
    g = load_golden("helpers.json")
    for r in g["arithmetic_to_log_params"]:
        mu, sg = oracle.log_params(r["mean"], r["vol"])
        assert_same_float(mu, r["mu_log"], "mu_log")
        assert_same_float(sg, r["sigma_log"], "sigma_log")
    for r in g["stream_start_month_index"]:
        assert oracle.stream_start_month_index(r["current_age"], r["working_months"], r["start_at_age"]) == r["start_month"]
    for r in g["trajectory_time_points"]:
        assert oracle.trajectory_time_points(r["working_months"], r["retirement_years"]) == r["points"]
    for r in g["monthly_gross"]:
        assert_same_float(oracle.monthly_gross(r["mu_log"], r["sigma_log"], r["z"]), r["gross"], "gross")
    for r in g["withdraw"]:
        b, c, t, u, rate = r["in"]
        got = oracle.withdraw(b, c, t, bool(u), rate)
        for x, y in zip(got, r["out"]):
            assert_same_float(x, y, f"withdraw{r['in']}")
    for r in g["nlv"]:
        b, c, u, rate = r["in"]
        assert_same_float(oracle.nlv(b, c, bool(u), rate), r["out"], f"nlv{r['in']}")
    params = [_params(c) for c in g["tax_cfgs"]]
    for r in g["rebalance"]:
        got = oracle.rebalance(params[r["cfg"]], *r["in"])
        for x, y in zip(got, r["out"]):
            assert_same_float(x, y, f"rebalance cfg{r['cfg']} {r['in']}")
    for r in g["annual_tax"]:
        got = oracle.annual_tax(params[r["cfg"]], *r["in"])
        for x, y in zip(got[:4], r["out"][:4]):
            assert_same_float(x, y, f"annual_tax cfg{r['cfg']} {r['in']}")
        assert got[4] == r["out"][4]


def test_log_params_errors(oracle):
    with pytest.raises(ValueError):
        oracle.log_params(-1.0, 0.1)
    with pytest.raises(ValueError):
        oracle.log_params(0.1, -0.1)


# ---- whole paths -------------------------------------------------------------------------
def test_deterministic_paths_match_reference_exactly(oracle):
    for case in load_golden("paths_deterministic.json"):
        p = _params(case["cfg"])
        res = oracle.run_batch(p, 0, 1, 0, 1, case["working_months"])
        try:
            compare_batch_to_golden(res, [case["result"]], exact=True)
        except AssertionError as e:
            raise AssertionError(f"{case['name']}: {e}") from e


@pytest.mark.parametrize("fname", ["paths_injected.json", "paths_fuzz.json"])
def test_stochastic_paths_match_reference_exactly(oracle, fname):
    """Same Philox shocks in the reference loop (injected) and in the oracle (own RNG)."""
    for g in load_golden(fname):
        p = _params(g["cfg"])
        res = oracle.run_batch(p, g["seed"], STREAM_ID[g["stream"]], g["path_begin"], g["n_paths"], g["working_months"])
        try:
            compare_batch_to_golden(res, g["results"], exact=True)
        except AssertionError as e:
            raise AssertionError(f"{g['name']}: {e}") from e
        assert int(res["counters"][0]) == sum(r["Success"] for r in g["results"])
        assert int(res["counters"][1]) == g["n_paths"]
        wr = np.array([r["WithdrawalRateTrajectory"] for r in g["results"]])
        assert res["wr_obs_counts"].tolist() == (~np.isnan(wr)).sum(axis=0).tolist()
        assert int(res["ruin_year_bins"].sum()) == sum(not r["Success"] for r in g["results"])


def test_numpy_native_shock_replay_matches_reference_exactly(oracle):
    """The reference's OWN NumPy shocks, stored verbatim, replayed through the oracle."""
    meta = load_golden("numpy_native_paths.json")
    arrays = np.load(os.path.join(GOLDEN, "numpy_native_shocks.npz"))
    for g in meta:
        p = _params(g["cfg"])
        sh = arrays[g["name"]]
        res = oracle.run_batch(p, 0, 1, 0, sh.shape[0], g["working_months"], injected_shocks=sh)
        compare_batch_to_golden(res, g["results"], exact=True)


def test_metric_10k_fixture(oracle):
    """BASELINE metric config: config.json, wm=233, 10k paths — per-path flags and summary."""
    meta = load_golden("metric_10k_config_json.json")
    z = np.load(os.path.join(GOLDEN, "metric_10k_config_json.npz"))
    n = meta["n_paths"]
    p = _params(meta["cfg"])
    res = oracle.run_batch(p, meta["seed"], STREAM_ID[meta["stream"]], 0, n, meta["working_months"],
                           want_trajectories=False)
    flags = np.unpackbits(z["success_bits"])[:n]
    assert np.array_equal(res["success"], flags)
    assert int(res["counters"][0]) == meta["success_count"]
    for key, field in (("Start_Balance", "start_balance"), ("Final_Balance", "final_balance"),
                       ("YearsToRuin", "years_to_ruin"), ("First_Year_Gross_Withdrawal", "first_year_gross_withdrawal"),
                       ("First_Year_Real_Gross_Withdrawal", "first_year_real_gross_withdrawal"),
                       ("Inflation_At_Retirement", "inflation_at_retirement")):
        assert np.array_equal(res[field], z[key], equal_nan=True), key
