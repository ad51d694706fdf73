"""Host logic of the drop-in simulator that needs no GPU: config validation, module helpers,
and the search driver driven by a fake backend — including a replay of the REFERENCE's own
recorded searches (tests/golden/search.json: curve, events, result)."""

from __future__ import annotations

import math

import pandas as pd
import pytest

from conftest import load_golden
from monte_carlo_retirement_amd import Config
from monte_carlo_retirement_amd.simulation import (
    RetirementMonteCarloSimulator,
    age_at_retirement_year,
    arithmetic_to_log_params,
    median_first_year_withdrawal_rate,
    retirement_age,
    stream_payment_start_age,
    stream_payment_start_month_index,
    trajectory_time_points,
)


def _base_config(**overrides) -> Config:
    data = load_golden("helpers.json")["tax_cfgs"][5].copy()  # the reference tests' base scenario values
    data.update({"allocation_inv1_pct": 0.6})
    data.update(overrides)
    return Config(**data)


def _fake_frame(n, ok_count):
    flags = [True] * ok_count + [False] * (n - ok_count)
    return pd.DataFrame({
        "Start Balance": [100.0] * n,
        "Final Balance": [1.0 if f else 0.0 for f in flags],
        "Success": flags,
        "First Year Gross Withdrawal": [1.0] * n,
        "Inflation At Retirement": [1.0] * n,
    })


def test_config_rejects_impossible_means_and_empty_search():
    """Mirror of the reference's validation test (:167-182)."""
    for bad in (dict(inv1_returns_mean=-1.0), dict(inflation_rate_mean=-1.0),
                dict(inv2_premium_over_inflation_mean=-1.0), dict(num_simulations_search=0), dict(seed=-1)):
        with pytest.raises(ValueError):
            _base_config(**bad)
    with pytest.raises(ValueError):
        RetirementMonteCarloSimulator(_base_config(seed=0), main_seed_override=-1)


def test_config_surface_defaults_and_alias():
    c = _base_config()
    assert c.Nickname == "test" and c.model_dump(by_alias=True)["scenario"] == "test"
    assert c.allocation_inv1_pct + c.allocation_inv2_pct == pytest.approx(1.0)
    c.retirement_years = 5  # validate_assignment
    with pytest.raises(ValueError):
        c.retirement_years = 0
    d = Config(scenario="x", initial_balance=1, monthly_contribution=0, monthly_expenses=0, current_age=30,
               retirement_years=1, allocation_inv1_pct=0.5, inv1_returns_mean=0.1, inv1_returns_volatility=0.1,
               inv1_annual_tax_on_gains_rate=0, inv2_premium_over_inflation_mean=0, inv2_premium_over_inflation_volatility=0,
               inv2_annual_tax_on_gains_rate=0, inflation_rate_mean=0, inflation_rate_volatility=0,
               num_simulations_main=1, num_simulations_search=1, target_probability=50, starting_working_months_search=0)
    assert (d.inv1_use_realized_gains_tax_system, d.inv2_use_realized_gains_tax_system) == (False, True)
    assert d.equity_inflation_correlation == 0.0 and d.seed is None and d.num_processes == 1 and d.other_income_streams == []


def test_module_helpers_match_reference():
    g = load_golden("helpers.json")
    for r in g["arithmetic_to_log_params"]:
        assert arithmetic_to_log_params(r["mean"], r["vol"]) == (r["mu_log"], r["sigma_log"])
    for r in g["stream_start_month_index"]:
        assert stream_payment_start_month_index(r["current_age"], r["working_months"], r["start_at_age"]) == r["start_month"]
    for r in g["trajectory_time_points"]:
        assert trajectory_time_points(r["working_months"], r["retirement_years"]) == r["points"]
    # the reference's own pins (:344-361)
    assert retirement_age(40.0, 240) == pytest.approx(60.0)
    assert stream_payment_start_age(40.0, 240, 65.0) == pytest.approx(65.0)
    assert stream_payment_start_age(40.0, 240, 55.0) == pytest.approx(60.0)
    assert age_at_retirement_year(40.0, 240, 5) == pytest.approx(65.0)
    assert stream_payment_start_month_index(60.0, 0, 60.51) == 7
    with pytest.raises(ValueError):
        arithmetic_to_log_params(-1.0, 0.1)
    with pytest.raises(ValueError):
        arithmetic_to_log_params(0.1, -0.1)


def test_median_first_year_withdrawal_rate():
    df = pd.DataFrame({"Start Balance": [100.0, 200.0, 0.0], "First Year Real Gross Withdrawal": [4.0, 10.0, 1.0],
                       "First Year Gross Withdrawal": [9.0, 9.0, 9.0]})
    assert median_first_year_withdrawal_rate(df) == pytest.approx(4.5)
    assert math.isnan(median_first_year_withdrawal_rate(pd.DataFrame()))


def test_bisection_finds_true_minimum():
    """Reference test :259-293: step function at 37 months through a patched backend."""
    sim = RetirementMonteCarloSimulator(_base_config(target_probability=90.0, num_simulations_search=10, seed=0))
    sim.run_monte_carlo_simulations = lambda wm, n: (_fake_frame(n, n if wm >= 37 else 0),) + (None,) * 6
    months, prob, curve = sim.find_minimum_working_months(verbose=False)
    assert months == 37 and prob >= 90.0 and len(curve) >= 1
    assert all("working_months" in p and "probability" in p for p in curve)


def test_search_verification_handles_non_monotone_probabilities():
    """Reference test :296-332: an isolated earlier pass at month 4 must be found."""
    sim = RetirementMonteCarloSimulator(_base_config(target_probability=50.0, num_simulations_search=400, seed=0))

    def fake(wm, n):
        ok = 201 if wm == 4 else (213 if wm >= 24 else 199)
        return (_fake_frame(n, ok),) + (None,) * 6

    sim.run_monte_carlo_simulations = fake
    months, probability, _ = sim.find_minimum_working_months(verbose=False)
    assert months == 4 and probability == pytest.approx(50.25)


def test_search_goes_through_an_overridden_batch_driver():
    """The reference's search always calls self.run_monte_carlo_simulations (simulation.py:1190): a subclass
    override or a class-level patch must be honoured exactly like a per-instance replacement."""
    from unittest import mock

    calls = []

    class Sub(RetirementMonteCarloSimulator):
        def run_monte_carlo_simulations(self, working_months, num_simulations):
            calls.append(working_months)
            return (_fake_frame(num_simulations, num_simulations if working_months >= 37 else 0),) + (None,) * 6

    sim = Sub(_base_config(target_probability=90.0, num_simulations_search=10, seed=0))
    assert sim.find_minimum_working_months(verbose=False)[0] == 37 and len(calls) > 3

    seen = []

    def fake(self, working_months, num_simulations):
        seen.append(working_months)
        return (_fake_frame(num_simulations, num_simulations if working_months >= 14 else 0),) + (None,) * 6

    with mock.patch.object(RetirementMonteCarloSimulator, "run_monte_carlo_simulations", fake):
        sim = RetirementMonteCarloSimulator(_base_config(target_probability=90.0, num_simulations_search=10, seed=0))
        assert sim.find_minimum_working_months(verbose=False)[0] == 14 and len(seen) > 3


def test_search_unreachable_target_returns_minus_one():
    sim = RetirementMonteCarloSimulator(_base_config(target_probability=99.0, num_simulations_search=10, seed=0))
    sim.run_monte_carlo_simulations = lambda wm, n: (_fake_frame(n, min(n - 1, wm // 100)),) + (None,) * 6
    months, prob, curve = sim.find_minimum_working_months(verbose=False)
    assert months == -1 and prob == pytest.approx(80.0) and curve[-1]["working_months"] == 70 * 12


@pytest.mark.parametrize("idx", [0, 1, 2, 3, 4, 5])
def test_search_replays_reference_run(idx):
    """Feed the probabilities the REFERENCE measured (search.json) through a fake backend: the
    driver must probe the same months in the same order, emit the same events and result."""
    g = load_golden("search.json")[idx]
    cfg = Config(**g["cfg"])
    n = cfg.num_simulations_search
    table = {e["working_months"]: int(round(e["probability"] * n / 100.0)) for e in g["events"] if e["type"] == "search_iter"}
    sim = RetirementMonteCarloSimulator(cfg, main_seed_override=g["seed"])
    sim.run_monte_carlo_simulations = lambda wm, k: (_fake_frame(k, table[wm]),) + (None,) * 6
    events = []
    months, prob, curve = sim.find_minimum_working_months(verbose=False, progress_callback=events.append)
    assert months == g["months"]
    assert prob == g["probability"]
    assert curve == g["search_curve"]
    assert events == g["events"]
    assert sim._stream_name == "search"


def test_config_schema_equals_the_references():
    """Field names, types, bounds, defaults, aliases and the required set are the reference's
    (tests/golden/config_schema.json = RefConfig.model_json_schema() without prose)."""
    import json

    def strip(o):
        if isinstance(o, dict):
            o.pop("description", None)
            o.pop("title", None)
            for v in o.values():
                strip(v)
        elif isinstance(o, list):
            for v in o:
                strip(v)
        return o

    mine = strip(json.loads(json.dumps(Config.model_json_schema())))
    assert mine == load_golden("config_schema.json")


@pytest.mark.parametrize("idx", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("slots", [1, 3, 5, 16])
def test_search_batches_candidates_without_changing_the_replay(idx, slots):
    """The GPU route evaluates candidates ahead of time, several per call (`_probe_many`).  With the
    probabilities the REFERENCE measured behind a fake `_probe_many`, months/probability/curve/events
    must still equal the reference's run for any speculation width; months the reference never probed
    may be evaluated but must not be reported, and no month is evaluated twice."""
    g = load_golden("search.json")[idx]
    cfg = Config(**g["cfg"])
    n = cfg.num_simulations_search
    table = {e["working_months"]: e["probability"] for e in g["events"] if e["type"] == "search_iter"}
    exact = {e["working_months"]: int(round(e["probability"] * n / 100.0)) / n * 100.0 for e in g["events"]
             if e["type"] == "search_iter"}
    sim = RetirementMonteCarloSimulator(cfg, main_seed_override=g["seed"])
    calls = []

    def fake_probe_many(months, k):
        assert k == n and len(set(months)) == len(months)
        calls.append(list(months))
        # a month outside the reference's run gets a neutral value between its probed neighbours
        known = sorted(exact)
        return {m: exact.get(m, exact[min(known, key=lambda x: abs(x - m))]) for m in months}

    sim._probe_many = fake_probe_many
    sim._speculation_slots = lambda k: slots
    events = []
    months, prob, curve = sim.find_minimum_working_months(verbose=False, progress_callback=events.append)
    assert months == g["months"]
    assert prob == pytest.approx(g["probability"], abs=1e-9)
    assert [c["working_months"] for c in curve] == [c["working_months"] for c in g["search_curve"]]
    assert [c["probability"] for c in curve] == [round(table[c["working_months"]], 2) for c in g["search_curve"]]
    assert [e for e in events if e["type"] != "search_iter"] == [e for e in g["events"] if e["type"] != "search_iter"]
    assert [(e["iteration"], e["working_months"], e["lo"], e["hi"]) for e in events if e["type"] == "search_iter"] == \
           [(e["iteration"], e["working_months"], e["lo"], e["hi"]) for e in g["events"] if e["type"] == "search_iter"]
    flat = [m for c in calls for m in c]
    assert len(flat) == len(set(flat))                      # nothing evaluated twice
    if slots == 1:
        # without speculation only the verification window is batched
        assert set(flat) == set(table)
    else:
        assert len(calls) < len(table) or len(table) == 1   # fewer launches than probes (a one-probe search has nothing to batch)
