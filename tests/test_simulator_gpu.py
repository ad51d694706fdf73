"""The drop-in simulator class on a real MI355X: the reference's own test scenarios
(tests/test_simulation_correctness.py) re-stated against this engine, the aggregation block vs
pandas goldens, and the full search vs the reference's recorded search."""

from __future__ import annotations

import math
import os

import numpy as np
import pandas as pd
import pytest

from conftest import STREAM_ID, load_golden
from monte_carlo_retirement_amd import Config
from monte_carlo_retirement_amd.constants import MONTHS_PER_YEAR, SMALL_EPSILON
from monte_carlo_retirement_amd.simulation import (
    SUMMARY_COLUMNS,
    RetirementMonteCarloSimulator,
    median_first_year_withdrawal_rate,
    trajectory_time_points,
)

pytestmark = pytest.mark.gpu


def _det(name):
    return [c for c in load_golden("paths_deterministic.json") if c["name"] == name][0]


def _sim(case, **kw):
    return RetirementMonteCarloSimulator(Config(**case["cfg"]), **kw)


def test_partial_year_inflation_accrual():  # reference :84-107
    c = _det("partial_year_inflation_accrual")
    r = _sim(c)._run_single_simulation_path(13, path_seed=99)
    assert abs(r["Inflation At Retirement"] - 1.06 ** (13 / MONTHS_PER_YEAR)) < 1e-9
    pts = trajectory_time_points(13, 1)
    assert pts == pytest.approx([0.0, 1.0, 13 / 12, 25 / 12]) and len(pts) == len(r["Trajectory"])


def test_partial_year_trajectory_keeps_equal_retirement_balance():  # :110-134
    r = _sim(_det("partial_year_equal_retirement_balance"))._run_single_simulation_path(working_months=13, path_seed=1)
    assert r["Trajectory"] == pytest.approx([100_000.0, 100_000.0, 100_000.0, 88_000.0])
    assert r["RealTrajectory"] == pytest.approx(r["Trajectory"])


def test_allocation_weights_conserve_every_dollar():  # :198-217
    r = _sim(_det("allocation_weights_conserve"))._run_single_simulation_path(working_months=0, path_seed=1)
    assert r["Start Balance"] == pytest.approx(100_000.0) and r["Trajectory"][0] == pytest.approx(100_000.0)


def test_perfect_equity_inflation_correlation_is_preserved():  # :185-195
    base = _det("years_to_ruin")["cfg"]
    pos = RetirementMonteCarloSimulator(Config(**dict(base, equity_inflation_correlation=1.0)))._draw_shock_path(100, path_seed=4)
    assert pos.shape == (100, 3) and pos[:, 1] == pytest.approx(pos[:, 0])
    neg = RetirementMonteCarloSimulator(Config(**dict(base, equity_inflation_correlation=-1.0)))._draw_shock_path(100, path_seed=4)
    assert neg[:, 1] == pytest.approx(-neg[:, 0])


def test_income_stream_cases():  # :363-404, :407-441, :444-493
    with_p = _sim(_det("income_stream_starts_at_age"))._run_single_simulation_path(240, 1)
    without = _sim(_det("income_stream_starts_at_age_no_pension"))._run_single_simulation_path(240, 1)
    assert with_p["Final Balance"] > 0 and with_p["Final Balance"] > without["Final Balance"]
    r = _sim(_det("income_stream_fractional_age"))._run_single_simulation_path(0, 4)
    assert r["Success"] is True and r["Final Balance"] == pytest.approx(0.0, abs=1e-6)
    assert r["First Year Gross Withdrawal"] == pytest.approx(6_000.0)
    sim = _sim(_det("pension_covers_after_depletion"))
    r = sim._run_single_simulation_path(0, 1)
    assert r["Success"] is True and r["Final Balance"] == pytest.approx(0.0, abs=1e-6)
    assert _sim(_det("pension_covers_after_depletion_no_pension"))._run_single_simulation_path(0, 1)["Success"] is False
    sim.use_final_seeds()
    summary = sim.run_monte_carlo_simulations(0, 5)[0]
    assert sim._success_probability(summary) == pytest.approx(100.0)
    assert (summary["Final Balance"] <= SMALL_EPSILON).all()


def test_withdrawal_rate_cases():  # :220-256, :496-564
    c = _det("withdrawal_rate_first_year")
    sim = _sim(c)
    r = sim._run_single_simulation_path(0, 1)
    wr = r["WithdrawalRateTrajectory"]
    expected = (r["First Year Gross Withdrawal"] / r["Start Balance"]) * 100.0
    assert len(wr) == 5 and wr[0] == pytest.approx(expected, abs=1e-6) and wr[1] == pytest.approx(wr[0], abs=1e-6)
    summary, _, _, wr_pct, _, _, wr_counts = sim.run_monte_carlo_simulations(0, 10)
    assert wr_pct is not None and not wr_pct.empty and wr_counts == [10] * 5
    assert abs(wr_pct.iloc[0][0.50] - expected) < 0.5
    assert abs(median_first_year_withdrawal_rate(summary) - wr_pct.iloc[0][0.50]) < 0.5
    sim.use_final_seeds()
    summary = sim.run_monte_carlo_simulations(working_months=0, num_simulations=20)[0]  # keyword call (server.py:431-434)
    assert abs(median_first_year_withdrawal_rate(summary) - 6.0) < 0.5
    for _, row in summary.iterrows():
        assert abs(row["First Year Gross Withdrawal"] - 12_000.0) < 1.0
    r = _sim(_det("real_wr_flat_deterministic_inflation"))._run_single_simulation_path(0, 3)
    assert r["Success"] is True and r["WithdrawalRateTrajectory"][0] == pytest.approx(5.0, abs=0.05)
    for rate in r["WithdrawalRateTrajectory"]:
        assert rate == pytest.approx(r["WithdrawalRateTrajectory"][0], abs=1e-4)


def test_years_to_ruin_and_real_trajectory():  # :567-602
    sim = _sim(_det("years_to_ruin"))
    r = sim._run_single_simulation_path(0, 1)
    assert r["Success"] is False and r["YearsToRuin"] == pytest.approx(3 / 12)
    assert len(r["RealTrajectory"]) == len(r["Trajectory"])
    for nom, real in zip(r["Trajectory"], r["RealTrajectory"]):
        assert real == pytest.approx(nom, abs=1e-6)
    summary, traj, _, _, real_traj, _, wr_counts = sim.run_monte_carlo_simulations(0, 20)
    assert (summary["Success"] == False).all() and summary["YearsToRuin"].notna().all()  # noqa: E712
    assert real_traj is not None and traj is not None and len(real_traj) == len(traj)
    assert wr_counts == [0] * 10


def test_helper_methods_unit_pins():  # :605-662
    sim = RetirementMonteCarloSimulator(Config(**load_golden("helpers.json")["tax_cfgs"][0]))
    assert sim._calculate_withdrawal_and_update(100.0, 0.0, 90.0, True, 0.20) == pytest.approx((0.0, 0.0, 100.0, 80.0))
    assert sim._calculate_withdrawal_and_update(80.0, 100.0, 40.0, True, 0.20) == pytest.approx((40.0, 50.0, 40.0, 40.0))
    b1, c1, b2, c2 = sim._rebalance_portfolio(bal_inv1=70.0, cb_inv1=50.0, bal_inv2=30.0, cb_inv2=30.0)
    assert b1 / (b1 + b2) == pytest.approx(0.60, abs=1e-10) and b1 + b2 < 100.0
    assert sim._net_liquidation_value(100.0, 0.0, True, 0.2) == pytest.approx(80.0)
    assert sim._monthly_gross_from_shock(0.12, 0.0, 1.0) == pytest.approx(math.exp(0.01))
    out = sim._apply_annual_gain_taxes(60.0, 60.0, 40.0, 40.0, 0.0, 0.0)
    assert out[:4] == pytest.approx((60.0, 60.0, 40.0, 40.0)) and out[4] is False


def test_annual_tax_cases():  # :665-734
    a = _sim(_det("annual_tax_excludes_transfers_no_tax"))._run_single_simulation_path(12, 1)
    b = _sim(_det("annual_tax_excludes_transfers_full_tax"))._run_single_simulation_path(12, 1)
    assert b["Start Balance"] == pytest.approx(a["Start Balance"], rel=1e-10)
    assert b["Final Balance"] == pytest.approx(a["Final Balance"], rel=1e-10)
    r = _sim(_det("retirement_does_not_split_tax_period"))._run_single_simulation_path(13, 1)
    assert r["Start Balance"] == pytest.approx((112.0 - 6.0) * 1.12 ** (1 / 12), rel=1e-10)


def test_success_probability_non_decreasing_in_working_months():  # :55-81
    g = load_golden("search.json")[1]
    sim = RetirementMonteCarloSimulator(Config(**dict(g["cfg"], num_simulations_main=80, seed=123)))
    sim.use_search_seeds()
    probs = [sim._success_probability(sim.run_monte_carlo_simulations(m, 80)[0]) for m in range(0, 61, 6)]
    assert all(b + 1e-9 >= a for a, b in zip(probs, probs[1:])), probs


@pytest.mark.parametrize("idx", [0, 1])
def test_aggregation_matches_pandas_on_reference_batch(idx):
    """a12: the 7-tuple for a 200-path batch vs the reference (same shocks).  Quantile frames use
    the same order statistics + NumPy-linear arithmetic: rel 1e-9 (path tolerance)."""
    g = load_golden("aggregation.json")[idx]
    sim = RetirementMonteCarloSimulator(Config(**g["cfg"]), main_seed_override=g["seed"])
    getattr(sim, f"use_{g['stream']}_seeds")()
    summary, traj, samples, wr, real, real_samples, wr_counts = sim.run_monte_carlo_simulations(g["working_months"], g["n_paths"])
    assert list(summary.columns) == SUMMARY_COLUMNS and summary["Success"].dtype == bool
    assert summary["Success"].tolist() == g["summary"]["Success"]
    for k in SUMMARY_COLUMNS:
        if k != "Success":
            np.testing.assert_allclose(summary[k].to_numpy(), np.array(g["summary"][k]), rtol=1e-9, atol=1e-6, equal_nan=True, err_msg=k)
    assert sim._success_probability(summary) == g["success_probability"]
    assert median_first_year_withdrawal_rate(summary) == pytest.approx(g["median_first_year_withdrawal_rate"], rel=1e-9)
    for frame, key in ((traj, "trajectory_percentiles"), (real, "real_trajectory_percentiles"), (wr, "wr_percentiles")):
        exp = g[key]
        assert [float(c) for c in frame.columns] == exp["columns"]
        assert list(frame.index) == list(range(len(exp["values"])))
        np.testing.assert_allclose(frame.to_numpy(), np.array(exp["values"], dtype=float), rtol=1e-9, atol=1e-6, equal_nan=True, err_msg=key)
    assert wr_counts == g["wr_observation_counts"]
    # the 5 sampled columns are the ones pandas picks for this main_seed
    np.testing.assert_allclose(np.array(samples), np.array(g["sample_trajectories"]), rtol=1e-9, atol=1e-6)
    np.testing.assert_allclose(np.array(real_samples), np.array(g["sample_real_trajectories"]), rtol=1e-9, atol=1e-6)


def test_row_quantiles_match_pandas_exactly_on_random_rows():
    """K3 against pandas on synthetic rows with NaNs, ties, infinities, ragged strides: BIT-exact
    (order statistics are exact; interpolation uses NumPy's own arithmetic)."""
    import torch

    from monte_carlo_retirement_amd import aggregation as A

    rng = np.random.default_rng(5)
    for n, stride in [(1, 64), (2, 64), (7, 64), (200, 256), (1001, 1024), (50_000, 50_048)]:
        rows = np.full((9, stride), 123.0)
        rows[0, :n] = rng.normal(1e6, 3e5, n)
        rows[1, :n] = rng.integers(0, 5, n).astype(float)          # heavy ties
        rows[2, :n] = np.where(rng.random(n) < 0.3, np.nan, rng.lognormal(0, 2, n))
        rows[3, :n] = np.nan                                        # all-NaN row
        rows[4, :n] = -rng.lognormal(3, 1, n)                       # negatives
        rows[5, :n] = 0.0
        rows[6, :n] = rng.choice([-np.inf, np.inf, 0.0, -0.0, 1e-300, -1e-300, 5.0], n)
        rows[7, :n] = np.arange(n, dtype=float)
        rows[8, :n] = rng.normal(0, 1, n) * 10.0 ** rng.integers(-8, 9, n)
        got, counts = A.row_quantiles(torch.as_tensor(rows, device="cuda"), n, A.TRAJECTORY_QUANTILES)
        exp = pd.DataFrame(rows[:, :n].T).quantile(list(A.TRAJECTORY_QUANTILES), axis=0).T.to_numpy()
        assert np.array_equal(got, exp, equal_nan=True), (n, got, exp)
        assert counts.tolist() == (~np.isnan(rows[:, :n])).sum(axis=1).tolist()


def test_success_histogram_matches_numpy():
    import torch

    from monte_carlo_retirement_amd import aggregation as A

    rng = np.random.default_rng(11)
    for n in (1, 5, 1000, 300_000):
        v = rng.lognormal(14, 1, n)
        ok = (rng.random(n) < 0.8).astype(np.uint8)
        if n == 5:
            v[:] = 7.0
        got, edges = A.success_histogram(torch.as_tensor(v, device="cuda"), torch.as_tensor(ok, device="cuda"), 100)
        sel = v[ok.astype(bool)]
        if sel.size == 0:
            assert got.sum() == 0
            continue
        exp, exp_edges = np.histogram(sel, bins=100)
        assert got.tolist() == exp.tolist()
        np.testing.assert_allclose(edges, exp_edges, rtol=0, atol=0)


@pytest.mark.parametrize("idx", [0, 1, 2, 3, 4, 5])
def test_full_search_matches_reference(idx):
    """Search driver end-to-end on the GPU (count-only probes) vs the reference's recorded search on
    the same shocks: same probes, same curve, same result."""
    g = load_golden("search.json")[idx]
    sim = RetirementMonteCarloSimulator(Config(**g["cfg"]), main_seed_override=g["seed"])
    events = []
    months, prob, curve = sim.find_minimum_working_months(verbose=False, progress_callback=events.append)
    assert (months, prob) == (g["months"], g["probability"])
    assert curve == g["search_curve"]
    assert events == g["events"]
    # the count-only probe equals the full run's success probability bit-for-bit
    if months >= 0:     # (-1: the target was out of reach, nothing to re-run)
        full = sim._success_probability(sim.run_monte_carlo_simulations(months, sim.params_model.num_simulations_search)[0])
        assert full == prob


def test_integration_md_ctypes_stub_runs_verbatim():
    """the ctypes stub a reference maintainer would paste is executed as written
    (against an object with the reference simulator's attributes) and agrees with the drop-in class."""
    import os
    import re

    from conftest import REPO
    from monte_carlo_retirement_amd import _native as N

    text = open(os.path.join(REPO, "INTEGRATION.md")).read()
    code = re.findall(r"```python\n(.*?)```", text, flags=re.S)[1]
    code = code.replace("/path/to/monte_carlo_retirement_amd/csrc/libmcr_hip.so", N.library_path())
    ns: dict = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    g = load_golden("paths_injected.json")[3]
    sim = RetirementMonteCarloSimulator(Config(**g["cfg"]), main_seed_override=g["seed"])
    sim.use_final_seeds()
    res = ns["run_paths_on_gpu"](sim, g["working_months"], g["n_paths"])
    assert len(res) == g["n_paths"]
    for got, exp in zip(res, g["results"]):
        assert got["Success"] == exp["Success"]
        for k in ("Start Balance", "Final Balance", "Inflation At Retirement", "First Year Real Gross Withdrawal"):
            assert got[k] == pytest.approx(exp[k], rel=1e-9, abs=1e-6)
        np.testing.assert_allclose(got["Trajectory"], exp["Trajectory"], rtol=1e-9, atol=1e-6)
        np.testing.assert_allclose(got["WithdrawalRateTrajectory"], exp["WithdrawalRateTrajectory"], rtol=1e-9, equal_nan=True)


def test_search_at_50k_paths_per_probe_has_its_defining_properties():
    """BASELINE configs[4] size (num_simulations_search = 50 000): the month found meets the target, every
    probed earlier month misses it, probes are memoised (no month twice), and the count-only probe of the
    chosen month equals a full run's success probability."""
    import json
    import os

    from conftest import REPO

    with open(os.path.join(REPO, "scenarios", "config.json")) as fh:
        cfg = Config(**dict(json.load(fh), num_simulations_search=50_000, seed=12345))
    sim = RetirementMonteCarloSimulator(cfg)
    months, prob, curve = sim.find_minimum_working_months(verbose=False)
    assert months > 0 and prob >= cfg.target_probability
    probed = [p["working_months"] for p in curve]
    assert len(probed) == len(set(probed))
    assert all(p["probability"] < cfg.target_probability for p in curve if p["working_months"] < months)
    assert months - 1 in probed  # the verification sweep tests the month just before the answer
    full = sim._success_probability(sim.run_monte_carlo_simulations(months, 50_000)[0])
    assert full == prob


@pytest.mark.parametrize("rng", ["philox", "numpy"])
def test_probe_many_equals_one_launch_per_candidate(rng):
    """mcr_probe_months_rng (candidates forked onto side streams) counts exactly what one count-only
    launch per candidate counts — also with more candidates than side streams, duplicates of a month,
    and on a non-default torch stream."""
    import torch

    from monte_carlo_retirement_amd import engine as E

    g = load_golden("search.json")[0]
    sim = RetirementMonteCarloSimulator(Config(**g["cfg"]), main_seed_override=g["seed"], rng=rng)
    sim.use_search_seeds()
    n = 4000
    months = [0, 1, 7, 12, 13, 24, 36, 37, 59, 60, 61, 120, 12]   # 13 candidates > 8 side streams; 12 twice
    params, r = sim._current_params(), sim._batch_rng(n)
    expect = []
    for m in months:
        b = E.DeviceBatch(params, m, n, want="count")
        b.launch(r, sim._stream_id, 0)
        expect.append(b.counters.cpu().tolist())
    got = E.probe_months(params, r, sim._stream_id, 0, n, months).cpu().tolist()
    assert got == expect
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        got2 = E.probe_months(params, r, sim._stream_id, 0, n, months)
    side.synchronize()
    assert got2.cpu().tolist() == expect
    probs = sim._probe_many(months[:-1], n)
    assert probs == {m: float(np.float64(e[0]) / np.float64(n) * 100.0) for m, e in zip(months[:-1], expect)}
    assert E.probe_months(params, r, sim._stream_id, 0, n, []).shape == (0, 2)
    with pytest.raises(RuntimeError, match="working_months must be >= 0"):
        E.probe_months(params, r, sim._stream_id, 0, n, [5, -1])
    # the failed call enqueued nothing: the next one is unaffected
    assert E.probe_months(params, r, sim._stream_id, 0, n, months[:3]).cpu().tolist() == expect[:3]


@pytest.mark.parametrize("idx", [0, 6, 7, 8, 9])
def test_shared_accumulation_probes_equal_separate_launches(idx):
    """Philox stream, 2..32 distinct candidates: mcr_probe_months_rng runs ONE accumulation sweep that stores the state at
    the end of every candidate month and ONE launch that resumes all decumulations (path_kernel PHASE 1 / 2).  The counts
    must equal one full launch per candidate — across tax systems (annual-tax scenarios carry the pre-retirement failure
    flag and the gain accumulators through the snapshot), stream start months that depend on the candidate, odd and even
    months (half-used Philox pairs), candidate 0, unsorted order — and more than 32 candidates take the plain route."""
    from monte_carlo_retirement_amd import engine as E

    g = load_golden("paths_injected.json")[idx]
    sim = RetirementMonteCarloSimulator(Config(**g["cfg"]), main_seed_override=4711)
    sim.use_search_seeds()
    n = 3000
    params, r = sim._current_params(), sim._batch_rng(n)

    def separately(months):
        out = []
        for m in months:
            b = E.DeviceBatch(params, m, n, want="count")
            b.launch(r, sim._stream_id, 5)
            out.append(b.counters.cpu().tolist())
        return out

    for months in ([37, 0, 14, 13, 36, 1, 2, 3, 50, 49, 12, 24], list(range(20, 52)), [7, 9], list(range(0, 66, 2))):
        got = E.probe_months(params, r, sim._stream_id, 5, n, months).cpu().tolist()
        assert got == separately(months), (g["name"], months)
    # the probabilities move with the candidate (the snapshots are not all the same state)
    few = E.probe_months(params, r, sim._stream_id, 5, n, [0, 30, 60, 240])[:, 0].cpu().tolist()
    assert all(c == n for c in [int(x) for x in E.probe_months(params, r, sim._stream_id, 5, n, [0, 30, 60, 240])[:, 1].cpu().tolist()])
    assert len(set(few)) > 1 or g["name"].startswith("PRETAX")


def test_batched_search_equals_one_probe_at_a_time():
    """find_minimum_working_months with candidate batches (the verification window in one call; bracket and
    bisection points evaluated ahead when they are free) returns exactly what strictly one launch per
    probe returns."""
    import json
    import os

    from conftest import REPO

    with open(os.path.join(REPO, "scenarios", "jorge.json")) as fh:
        cfg = Config(**dict(json.load(fh), num_simulations_search=20_000, seed=99))
    ea, eb, calls_a, calls_b = [], [], [], []
    a = RetirementMonteCarloSimulator(cfg)
    a._speculation_slots = lambda n: 4          # as under a 4-rank process group
    many_a = a._probe_many
    a._probe_many = lambda months, n: (calls_a.append(len(months)), many_a(months, n))[1]
    ra = a.find_minimum_working_months(verbose=False, progress_callback=ea.append)
    b = RetirementMonteCarloSimulator(cfg)
    assert b._speculation_slots(20_000) == 3 and b._speculation_slots(1_000_000) == 1   # one GPU: small probes only
    b._speculation_slots = lambda n: 1          # strictly the reference's sequence: one month per round, one launch per month
    many_b = b._probe_many
    b._probe_many = lambda months, n: {m: (calls_b.append(1), many_b([m], n))[1][m] for m in months}
    rb = b.find_minimum_working_months(verbose=False, progress_callback=eb.append)
    assert ra == rb and ea == eb
    assert max(calls_a) > 1 and len(calls_a) < len(calls_b) <= len(rb[2])   # fewer, wider calls


def _rows_for_bracket_test(rng, n, stride):
    rows = np.full((14, stride), 123.0)
    rows[0, :n] = rng.normal(1e6, 3e5, n)
    rows[1, :n] = rng.integers(0, 5, n).astype(float)                      # 5 giant ties: candidates overflow -> fallback
    rows[2, :n] = np.where(rng.random(n) < 0.3, np.nan, rng.lognormal(0, 2, n))
    rows[3, :n] = np.nan                                                    # all NaN
    rows[4, :n] = -rng.lognormal(3, 1, n)
    rows[5, :n] = 0.0                                                       # constant
    rows[6, :n] = rng.choice([-np.inf, np.inf, 0.0, -0.0, 1e-300, -1e-300, 5.0], n)
    rows[7, :n] = np.arange(n, dtype=float)                                 # sorted: the prefix sample is useless -> fallback
    rows[8, :n] = rng.normal(0, 1, n) * 10.0 ** rng.integers(-8, 9, n)
    rows[9, :n] = np.where(np.arange(n) < n // 16, np.nan, rng.lognormal(12, 1, n))   # the sample is mostly NaN
    rows[10, :n] = np.where(rng.random(n) < 0.65, 0.0, rng.lognormal(10, 2, n))       # 65 % zeros + a continuous tail
    rows[11, :n] = np.sort(rng.normal(0, 1, n))[::-1]                       # descending
    rows[12, :n] = rng.normal(5e5, 1.0, n).round(0)                         # few distinct values around the median
    rows[13, :n] = np.where(np.arange(n) < 70_000, np.nan, rng.normal(0, 1, n))       # sample has NO valid entry
    return rows


@pytest.mark.parametrize("n,stride", [(4096, 4096), (65_536, 65_600), (300_001, 300_032), (300_000, 300_001)])
def test_bracketed_row_quantiles_match_pandas_exactly(n, stride, monkeypatch):
    """The sample-bracketed single pass (forced here for short rows through MCR_RQ_BRACKET_MIN_N) returns bit for
    bit what the full radix passes and pandas return — on rows its brackets decide and on rows that must fall
    back (giant ties, sorted input, NaN-only sample, constant rows) — for the trajectory and the WR quantile sets
    and for the extreme quantiles 0 and 1."""
    import torch

    from monte_carlo_retirement_amd import aggregation as A

    rng = np.random.default_rng(n)
    rows = _rows_for_bracket_test(rng, n, stride)
    dev_rows = torch.as_tensor(rows, device="cuda")
    nine = (0.01, 0.05, 0.10, 0.25, 0.50, 0.75, 0.90, 0.95, 0.99)       # the final-balance percentiles: 32-entry bound table
    for qs in (A.TRAJECTORY_QUANTILES, A.WR_QUANTILES, (0.0, 1.0, 0.5, 0.999, 0.001), nine, tuple(np.linspace(0.02, 0.98, 15)),
               tuple(np.linspace(0.03, 0.97, 16))):
        exp = pd.DataFrame(rows[:, :n].T).quantile(list(qs), axis=0).T.to_numpy()
        monkeypatch.setenv("MCR_RQ_BRACKET_MIN_N", "1")
        got, counts = A.row_quantiles(dev_rows, n, qs)
        n_fb = A.last_fallback_rows()
        # rows 3, 9 and 13 have no valid entry in their first sample (the leading 4096 entries); the sorted rows 7 and
        # 11 defeat a prefix sample unless the sample is the whole row; a bracket that straddles two giant ties (rows
        # 1, 6, 12) overflows the candidate buffer, and so does an OPEN bracket (q = 0 or 1) that ends on a giant tie
        # (rows 5, 10).  A bracket INSIDE one tie is a one-key interval, not a fallback; the continuous rows
        # (0, 2, 4, 8) never fall back.
        if len(qs) <= 7:
            assert 2 <= n_fb <= 10, n_fb
        elif len(qs) < 16:
            # many quantiles on a short row: with the minimum 65 536-entry sample the brackets together can hold more
            # than the candidate buffer (n/8) — those rows fall back too (still exact); the bracketed route is taken
            assert 2 <= n_fb <= rows.shape[0], n_fb
        else:
            assert n_fb == -1            # 16 quantiles: 32 bounds do not fit the bound table, plain radix route
        monkeypatch.setenv("MCR_RQ_BRACKET_MIN_N", str(2**40))
        plain, counts_plain = A.row_quantiles(dev_rows, n, qs)
        assert A.last_fallback_rows() == -1
        assert np.array_equal(plain, exp, equal_nan=True)
        bad = [(r, got[r], exp[r]) for r in range(rows.shape[0]) if not np.array_equal(got[r], exp[r], equal_nan=True)]
        assert not bad, (n, qs, bad[:3])
        assert counts.tolist() == counts_plain.tolist() == (~np.isnan(rows[:, :n])).sum(axis=1).tolist()


def test_bracketed_row_quantiles_at_the_default_threshold():
    """3 x 2^22 entries: above the default threshold, so the default route is the bracketed one; equals numpy."""
    import torch

    from monte_carlo_retirement_amd import aggregation as A

    n = 1 << 22
    rng = np.random.default_rng(3)
    rows = np.empty((3, n))
    rows[0] = rng.lognormal(14, 1.2, n)
    rows[1] = np.where(rng.random(n) < 0.2, np.nan, rng.normal(4.0, 1.5, n))
    rows[2] = 1.0e6
    got, counts = A.row_quantiles(torch.as_tensor(rows, device="cuda"), n, A.TRAJECTORY_QUANTILES)
    assert A.last_fallback_rows() == 0       # bracketed route, every row decided on its candidates
    for r in range(3):
        v = rows[r][~np.isnan(rows[r])]
        assert np.array_equal(got[r], np.quantile(v, A.TRAJECTORY_QUANTILES)), r
        assert counts[r] == v.size
    # nine quantiles (the response document's final-balance percentiles): the 32-entry bound table, still no fallback
    nine = (0.01, 0.05, 0.10, 0.25, 0.50, 0.75, 0.90, 0.95, 0.99)
    got9, _ = A.row_quantiles(torch.as_tensor(rows, device="cuda"), n, nine)
    assert A.last_fallback_rows() == 0
    for r in range(3):
        assert np.array_equal(got9[r], np.quantile(rows[r][~np.isnan(rows[r])], nine)), r


def _random_row(rng, n):
    """One row from a menu of shapes that stress the bracketed route differently (scale, sign, ties, NaNs, order)."""
    kind = int(rng.integers(0, 12))
    scale = 10.0 ** rng.uniform(-6, 12)
    if kind == 0:
        x = rng.normal(rng.normal(0, 3) * scale, scale, n)
    elif kind == 1:
        x = rng.lognormal(rng.uniform(-3, 15), rng.uniform(0.1, 3), n) * (-1.0 if rng.random() < 0.3 else 1.0)
    elif kind == 2:
        x = rng.integers(0, int(rng.integers(2, 2000)), n).astype(float) * scale          # ties, few to many distinct values
    elif kind == 3:
        x = np.where(rng.random(n) < rng.uniform(0.05, 0.95), 0.0, rng.lognormal(5, 2, n))  # a spike at zero + a tail
    elif kind == 4:
        x = np.sort(rng.normal(0, scale, n))[:: (1 if rng.random() < 0.5 else -1)]         # monotone
    elif kind == 5:
        x = rng.normal(0, 1, n) * 10.0 ** rng.integers(-12, 13, n)                        # every magnitude, both signs
    elif kind == 6:
        x = rng.choice([-np.inf, np.inf, 0.0, -0.0, 5e-324, -5e-324, 1.0, -1.0, scale], n)
    elif kind == 7:
        x = np.full(n, rng.normal(0, scale))                                              # constant
    elif kind == 8:
        x = np.concatenate([np.full(n // 3, -scale), rng.normal(0, scale, n - 2 * (n // 3)), np.full(n // 3, scale)])
        rng.shuffle(x)                                                                    # two giant ties around a continuum
    elif kind == 9:
        x = rng.standard_t(2, n) * scale                                                  # heavy tails
    elif kind == 10:
        x = rng.uniform(-scale, scale, n).round(int(rng.integers(0, 3)))
    else:
        x = np.where(np.arange(n) < int(rng.integers(1, n)), rng.normal(-5 * scale, scale, n), rng.normal(5 * scale, scale, n))  # regime change
    if rng.random() < 0.35:                                                               # NaNs: scattered, leading, or nearly all
        mode = int(rng.integers(0, 3))
        if mode == 0:
            x = np.where(rng.random(n) < rng.uniform(0.01, 0.6), np.nan, x)
        elif mode == 1:
            x[: int(rng.integers(1, n))] = np.nan
        else:
            keep = rng.integers(0, n, int(rng.integers(1, 50)))
            y = np.full(n, np.nan); y[keep] = x[keep]; x = y
    return x


def test_bracketed_row_quantiles_random_shapes(monkeypatch):
    """Randomised shapes for the bracketed route (forced through MCR_RQ_BRACKET_MIN_N): row length, stride, row count,
    row contents and quantile set all drawn; every entry must equal pandas' bit for bit whether the row was decided on
    its brackets or fell back.  (pandas, which the reference calls — simulation.py:1059-1113 —, not np.quantile: for a q
    that is not a short binary fraction NumPy's own virtual index n q + (1 - q) - 1 rounds differently from the
    (n - 1) q pandas ends up with, and the last bits of the interpolation follow.)
    MCR_RQ_FUZZ_SEED / MCR_RQ_FUZZ_ROUNDS lengthen it for soaks by hand."""
    import torch

    from monte_carlo_retirement_amd import aggregation as A

    rng = np.random.default_rng(int(os.environ.get("MCR_RQ_FUZZ_SEED", "20260102")))
    monkeypatch.setenv("MCR_RQ_BRACKET_MIN_N", "1")
    decided = fell_back = 0
    for _ in range(int(os.environ.get("MCR_RQ_FUZZ_ROUNDS", "8"))):
        n = int(rng.choice([4096, 4097, 5000, 70_000, int(rng.integers(4096, 400_000)), int(rng.integers(400_000, 3_000_000))]))
        stride = n + int(rng.choice([0, 1, 2, 7, 64]))
        n_rows = int(rng.integers(1, 12)) if n < 1_000_000 else int(rng.integers(1, 5))
        rows = np.full((n_rows, stride), -7.0)
        for r in range(n_rows):
            rows[r, :n] = _random_row(rng, n)
        k = int(rng.integers(1, 16))
        qs = tuple(sorted(set(np.round(rng.uniform(0, 1, k), int(rng.integers(1, 6))).tolist() + ([0.0] if rng.random() < 0.2 else []) + ([1.0] if rng.random() < 0.2 else []))))[:15]
        got, counts = A.row_quantiles(torch.as_tensor(rows, device="cuda"), n, qs)
        n_fb = A.last_fallback_rows()
        assert n_fb >= 0, (n, qs)                     # the bracketed route was taken
        fell_back += n_fb
        decided += n_rows - n_fb
        exp = pd.DataFrame(rows[:, :n].T).quantile(list(qs), axis=0).T.to_numpy()
        for r in range(n_rows):
            assert np.array_equal(got[r], exp[r], equal_nan=True), (n, stride, r, qs, got[r].tolist(), exp[r].tolist())
            assert counts[r] == int((~np.isnan(rows[r, :n])).sum())
    assert decided > 0
    print("bracketed route, random shapes: rows decided on their brackets", decided, "fell back", fell_back)


def test_shared_accumulation_probes_random_scenarios():
    """Randomised: scenario (the differential test's generator: every tax system, up to 6 income streams, extreme
    rates), candidate set (1-40 months from 0 to 300, unsorted, with duplicates), path range and batch size all drawn.
    mcr_probe_months_rng must count exactly what one full launch per candidate counts.
    MCR_PROBE_FUZZ_SEED / MCR_PROBE_FUZZ_ROUNDS lengthen it for soaks by hand."""
    from monte_carlo_retirement_amd import params_from_config
    from monte_carlo_retirement_amd import engine as E
    from test_gpu_differential import _random_config

    rng = np.random.default_rng(int(os.environ.get("MCR_PROBE_FUZZ_SEED", "20260103")))
    shared = 0
    for _ in range(int(os.environ.get("MCR_PROBE_FUZZ_ROUNDS", "12"))):
        cfgd = _random_config(rng)
        params = params_from_config(Config(**cfgd))
        n = int(rng.choice([1, 63, 64, 65, 257, 1000, int(rng.integers(1, 6000))]))
        begin = int(rng.choice([0, 7, 2**32 - 100, 2**40]))
        seed, stream = int(rng.integers(0, 2**63)), int(rng.integers(2))
        k = int(rng.integers(1, 41))
        top = int(rng.choice([12, 40, 130, 300]))
        months = [int(m) for m in rng.integers(0, top + 1, k)]
        if rng.random() < 0.3:
            months = sorted(set(months))
        got = E.probe_months(params, seed, stream, begin, n, months).cpu().tolist()
        exp = []
        for m in months:
            b = E.DeviceBatch(params, m, n, want="count")
            b.launch(seed, stream, begin)
            exp.append(b.counters.cpu().tolist())
        assert got == exp, (cfgd, seed, stream, begin, n, months)
        shared += 2 <= len(set(months)) <= 32
    assert shared > 0


def test_success_histogram_random_inputs():
    """np.histogram(values[success], bins) on randomised inputs: sizes from 1 to 3e6, value scales from 1e-9 to 1e13,
    heavy ties on the edges (zeros — the final balance of a ruined path — and the maximum), cohort fractions from none
    to all, 1 to 400 bins, fixed and derived ranges.  Counts must be identical, edges bit-equal.
    MCR_HIST_FUZZ_SEED / MCR_HIST_FUZZ_ROUNDS lengthen it for soaks by hand."""
    import torch

    from monte_carlo_retirement_amd import aggregation as A

    rng = np.random.default_rng(int(os.environ.get("MCR_HIST_FUZZ_SEED", "20260105")))
    for _ in range(int(os.environ.get("MCR_HIST_FUZZ_ROUNDS", "25"))):
        n = int(rng.choice([1, 2, 63, 64, 65, 1000, int(rng.integers(1, 100_000)), int(rng.integers(100_000, 3_000_000))]))
        scale = 10.0 ** rng.uniform(-9, 13)
        kind = int(rng.integers(0, 5))
        if kind == 0:
            v = rng.lognormal(0, rng.uniform(0.1, 2.5), n) * scale
        elif kind == 1:
            v = np.where(rng.random(n) < rng.uniform(0, 0.9), 0.0, rng.lognormal(0, 1, n) * scale)   # ruined paths: exact zeros
        elif kind == 2:
            v = rng.integers(0, int(rng.integers(1, 50)), n).astype(float) * scale                    # values ON bin edges
        elif kind == 3:
            v = np.full(n, scale)                                                                     # degenerate range
        else:
            v = rng.uniform(0, scale, n)
            v[rng.integers(0, n, max(1, n // 10))] = v.max()                                           # ties at the right edge
        ok = (rng.random(n) < rng.choice([0.0, 0.02, 0.5, 0.97, 1.0])).astype(np.uint8)
        bins = int(rng.choice([1, 2, 7, 60, 100, 400]))
        sel = v[ok.astype(bool)]
        rng_arg = None
        if rng.random() < 0.3 and sel.size:
            lo = float(sel.min() - rng.uniform(0, 1) * scale)
            rng_arg = (lo, float(sel.max() + rng.uniform(0, 1) * scale))
        got, edges = A.success_histogram(torch.as_tensor(v, device="cuda"), torch.as_tensor(ok, device="cuda"), bins, value_range=rng_arg)
        if sel.size == 0:
            assert got.sum() == 0
            continue
        try:
            exp, exp_edges = np.histogram(sel, bins=bins, range=rng_arg)
        except ValueError:            # "Too many bins for data range": numpy (hence the reference) refuses; nothing to match
            continue
        assert got.tolist() == exp.tolist(), (n, kind, bins, rng_arg, scale)
        assert np.array_equal(edges, exp_edges), (n, kind, bins, rng_arg)


def test_batched_search_equals_one_probe_at_a_time_on_random_scenarios():
    """The search on randomised scenarios (the differential test's generator + random target probability, starting
    month and paths per probe), with 1, 2, 4 or 8 speculation slots: result, probe curve and progress events must equal
    those of strictly one launch per probe — including searches that hit the reference's upper bound or find the
    target at month 0.  MCR_SEARCH_FUZZ_SEED / MCR_SEARCH_FUZZ_ROUNDS lengthen it for soaks by hand."""
    from test_gpu_differential import _random_config

    rng = np.random.default_rng(int(os.environ.get("MCR_SEARCH_FUZZ_SEED", "20260107")))
    found = 0
    for _ in range(int(os.environ.get("MCR_SEARCH_FUZZ_ROUNDS", "10"))):
        cfgd = _random_config(rng)
        cfgd.update(num_simulations_search=int(rng.choice([1, 50, 300, 2000])), num_simulations_main=10,
                    target_probability=float(rng.choice([1.0, 50.0, 90.0, 97.0, 99.9, 100.0])),
                    starting_working_months_search=int(rng.choice([0, 0, 12, 100, 400])), seed=int(rng.integers(0, 2**31)))
        cfg = Config(**cfgd)
        slots = int(rng.choice([1, 2, 4, 8]))
        ea, eb = [], []
        a = RetirementMonteCarloSimulator(cfg)
        a._speculation_slots = lambda n, s=slots: s
        b = RetirementMonteCarloSimulator(cfg)
        many_b = b._probe_many
        b._probe_many = lambda months, n, f=many_b: {m: f([m], n)[m] for m in months}
        ra = a.find_minimum_working_months(verbose=False, progress_callback=ea.append)
        rb = b.find_minimum_working_months(verbose=False, progress_callback=eb.append)
        assert ra == rb and ea == eb, (cfgd, slots, ra, rb)
        found += ra[0] is not None
    assert found > 0
