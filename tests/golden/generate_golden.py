#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE implementation.

Runs ONLY in the build container (needs /root/reference, which never travels to the GPU
box).  It imports the reference's own backend (pure Python; `loguru` is absent in this
image so a no-op stub module is put on sys.path from a temp dir), drives
`RetirementMonteCarloSimulator._run_single_simulation_path` / `run_monte_carlo_simulations`
/ `find_minimum_working_months` and the scalar helpers, and writes inputs + expected
outputs as JSON / NPZ data files.  No reference source text is stored.

Stochastic paths use SHOCK INJECTION: the engine's own Philox/Box-Muller shock rows
(oracle/mcr_oracle.c: orc_draw_shocks) are assigned to `sim._draw_shock_path`
(the method is looked up on `self`, backend/simulation.py:488), so reference, oracle and
HIP kernel consume identical random numbers.  A few vectors also carry NumPy-native shock
arrays (the reference's own RNG) stored verbatim.

Usage:  python tests/golden/generate_golden.py [--skip-10k]
"""

from __future__ import annotations

import argparse
import json
import math
import os
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"
sys.path.insert(0, REPO)

_stub_dir = tempfile.mkdtemp(prefix="loguru_stub_")
with open(os.path.join(_stub_dir, "loguru.py"), "w") as fh:
    fh.write(
        "class _L:\n"
        "    def __getattr__(self, name):\n"
        "        return lambda *a, **k: None\n"
        "logger = _L()\n"
    )
sys.path.insert(0, _stub_dir)
sys.path.insert(0, os.path.join(REF, "backend"))

import simulation as ref_sim  # noqa: E402  (the reference)
from config import Config as RefConfig  # noqa: E402

from oracle import oracle as O  # noqa: E402

SUMMARY_KEYS = [
    "Start Balance", "Final Balance", "Success", "YearsToRuin",
    "First Year Gross Withdrawal", "First Year Real Gross Withdrawal",
    "Inflation At Retirement",
]
STREAM_ID = {"search": 0, "final": 1}


def load_json(name):
    with open(os.path.join(REF, name)) as fh:
        return json.load(fh)


def base_test_config(**over):
    """The reference tests' `_base_config` values (tests/test_simulation_correctness.py:20-52), as data."""
    d = {
        "scenario": "test", "initial_balance": 500_000.0, "monthly_contribution": 0.0,
        "contribution_growth_rate_annual": 0.0, "monthly_expenses": 2_000.0, "current_age": 40.0,
        "retirement_years": 10, "allocation_inv1_pct": 0.6, "inv1_returns_mean": 0.08,
        "inv1_returns_volatility": 0.15, "inv1_annual_tax_on_gains_rate": 0.0,
        "inv1_realized_gains_tax_rate": 0.0, "inv1_use_realized_gains_tax_system": False,
        "inv2_premium_over_inflation_mean": 0.02, "inv2_premium_over_inflation_volatility": 0.01,
        "inv2_annual_tax_on_gains_rate": 0.0, "inv2_realized_gains_tax_rate": 0.0,
        "inv2_use_realized_gains_tax_system": False, "inflation_rate_mean": 0.03,
        "inflation_rate_volatility": 0.01, "equity_inflation_correlation": 0.0,
        "num_simulations_main": 50, "num_simulations_search": 40, "target_probability": 80.0,
        "starting_working_months_search": 0, "seed": 42, "num_processes": 1,
        "other_income_streams": [],
    }
    d.update(over)
    return d


def make_sim(cfg_dict, seed=None):
    cfg = RefConfig(**cfg_dict)
    sim = ref_sim.RetirementMonteCarloSimulator(cfg, main_seed_override=seed)
    return sim


def inject_engine_shocks(sim, seed):
    """Route the reference's shock draw to the engine's Philox stream; path_seed == path index."""
    rho = sim._equity_inflation_rho

    def draw(n_months, path_seed):
        return O.draw_shocks(seed, STREAM_ID[sim._stream_name], int(path_seed), int(n_months), rho)

    sim._draw_shock_path = draw
    sim._path_seeds = lambda n: list(range(n))


def result_to_jsonable(r):
    out = {}
    for k, v in r.items():
        if isinstance(v, list):
            out[k] = [float(x) for x in v]
        elif isinstance(v, bool):
            out[k] = v
        else:
            out[k] = float(v)
    return out


# ---------------------------------------------------------------------------------------
def gen_helpers():
    rng = np.random.default_rng(2024)
    out = {}
    # a1
    rows = []
    for mean, vol in [(0.12, 0.02), (0.062, 0.0235), (0.05, 0.02), (0.08, 0.15), (0.0, 0.0),
                      (0.06, 0.0), (-0.5, 0.3), (1.0, 0.0), (0.03, 0.01), (0.02, 0.01), (0.10, 0.12)]:
        mu, sg = ref_sim.arithmetic_to_log_params(mean, vol)
        rows.append({"mean": mean, "vol": vol, "mu_log": mu, "sigma_log": sg})
    out["arithmetic_to_log_params"] = rows
    # a2
    rows = []
    for ca, wm, sa in [(40.0, 240, 65.0), (40.0, 240, 55.0), (60.0, 0, 60.51), (60.0, 0, 60.5),
                       (35.0, 75, 65.0), (40.0, 233, 65.0), (40.0, 233, 40.0), (40.0, 13, 41.1),
                       (59.99, 1, 60.0), (30.5, 7, 31.0), (0.0, 0, 120.0)]:
        rows.append({"current_age": ca, "working_months": wm, "start_at_age": sa,
                     "start_month": ref_sim.stream_payment_start_month_index(ca, wm, sa)})
    out["stream_start_month_index"] = rows
    # a14
    rows = []
    for wm, ry in [(0, 1), (13, 1), (12, 3), (233, 50), (75, 40), (120, 50), (1, 2), (24, 2)]:
        rows.append({"working_months": wm, "retirement_years": ry,
                     "points": ref_sim.trajectory_time_points(wm, ry)})
    out["trajectory_time_points"] = rows
    # a6
    rows = []
    for _ in range(24):
        mu, sg, z = float(rng.uniform(-0.2, 0.2)), float(rng.uniform(0, 0.4)), float(rng.normal())
        sim0 = make_sim(base_test_config())
        rows.append({"mu_log": mu, "sigma_log": sg, "z": z,
                     "gross": sim0._monthly_gross_from_shock(mu, sg, z)})
    out["monthly_gross"] = rows

    # a8 / a9: grid incl. the reference tests' exact cases (:605-631)
    sim0 = make_sim(base_test_config())
    cases = [(100.0, 0.0, 90.0, True, 0.20), (80.0, 100.0, 40.0, True, 0.20),
             (0.0, 0.0, 10.0, True, 0.1), (1e-7, 5.0, 10.0, True, 0.1), (50.0, 20.0, 0.0, True, 0.1),
             (50.0, 20.0, -1.0, True, 0.1), (50.0, 20.0, 10.0, False, 0.1), (50.0, 20.0, 10.0, True, 0.0),
             (50.0, 20.0, 1e9, True, 0.3), (50.0, 60.0, 49.9999995, True, 0.3), (1e6, 0.0, 5e5, True, 1.0)]
    for _ in range(40):
        bal = float(rng.choice([0.0, rng.uniform(0, 10), rng.uniform(0, 1e6)]))
        cb = float(rng.uniform(0, 1.5) * bal)
        tgt = float(rng.choice([0.0, rng.uniform(0, 2) * bal, rng.uniform(0, 1e3)]))
        cases.append((bal, cb, tgt, bool(rng.integers(2)), float(rng.choice([0.0, rng.uniform(0, 1)]))))
    out["withdraw"] = [
        {"in": [b, c, t, float(u), r], "out": list(sim0._calculate_withdrawal_and_update(b, c, t, u, r))}
        for (b, c, t, u, r) in cases
    ]
    out["nlv"] = [
        {"in": [b, c, float(u), r], "out": sim0._net_liquidation_value(b, c, u, r)}
        for (b, c, _t, u, r) in cases
    ]

    # a7 / a10 under several tax configurations
    tax_cfgs = [
        dict(allocation_inv1_pct=0.60, inv1_use_realized_gains_tax_system=True, inv1_realized_gains_tax_rate=0.10,
             inv2_use_realized_gains_tax_system=True, inv2_realized_gains_tax_rate=0.10),  # tests :634-662
        dict(allocation_inv1_pct=0.50, inv1_use_realized_gains_tax_system=False, inv1_annual_tax_on_gains_rate=0.15,
             inv2_use_realized_gains_tax_system=False, inv2_annual_tax_on_gains_rate=0.25),
        dict(allocation_inv1_pct=0.30, inv1_use_realized_gains_tax_system=False, inv1_annual_tax_on_gains_rate=1.0,
             inv2_use_realized_gains_tax_system=True, inv2_realized_gains_tax_rate=0.9),
        dict(allocation_inv1_pct=1.0, inv1_use_realized_gains_tax_system=True, inv1_realized_gains_tax_rate=0.2),
        dict(allocation_inv1_pct=0.0, inv2_use_realized_gains_tax_system=True, inv2_realized_gains_tax_rate=0.2),
        dict(allocation_inv1_pct=0.333333),
    ]
    reb, tax = [], []
    out["tax_cfgs"] = [base_test_config(**tc) for tc in tax_cfgs]
    for ci, tc in enumerate(tax_cfgs):
        cfgd = base_test_config(**tc)
        sim = make_sim(cfgd)
        pts = [(70.0, 50.0, 30.0, 30.0), (30.0, 30.0, 70.0, 20.0), (0.0, 0.0, 0.0, 0.0), (60.0, 10.0, 40.0, 10.0),
               (60.0000001, 10.0, 39.9999999, 10.0), (1e-7, 1e-7, 0.0, 0.0), (100.0, 150.0, 0.0, 0.0), (0.0, 0.0, 100.0, 10.0)]
        for _ in range(16):
            b1, b2 = float(rng.uniform(0, 1e6)), float(rng.uniform(0, 1e6))
            pts.append((b1, float(rng.uniform(0, 1.3) * b1), b2, float(rng.uniform(0, 1.3) * b2)))
        for q in pts:
            reb.append({"cfg": ci, "in": list(q), "out": list(sim._rebalance_portfolio(*q))})
            g1 = float(rng.choice([0.0, rng.uniform(-1, 1) * (q[0] + 1.0), 5.0 * (q[0] + q[2] + 1.0)]))
            g2 = float(rng.choice([0.0, rng.uniform(-1, 1) * (q[2] + 1.0)]))
            r = sim._apply_annual_gain_taxes(q[0], q[1], q[2], q[3], g1, g2)
            tax.append({"cfg": ci, "in": list(q) + [g1, g2], "out": [float(x) for x in r[:4]] + [bool(r[4])]})
    out["rebalance"] = reb
    out["annual_tax"] = tax
    return out


# ---------------------------------------------------------------------------------------
ZERO = dict(inflation_rate_mean=0.0, inflation_rate_volatility=0.0, inv1_returns_mean=0.0,
            inv1_returns_volatility=0.0, inv2_premium_over_inflation_mean=0.0,
            inv2_premium_over_inflation_volatility=0.0)
NOTAX = dict(inv1_use_realized_gains_tax_system=False, inv1_annual_tax_on_gains_rate=0.0,
             inv2_use_realized_gains_tax_system=False, inv2_annual_tax_on_gains_rate=0.0)


def pension(amount, age, indexed=True, tax=0.0, dur=None, name="Pension"):
    return {"name": name, "monthly_amount_today": amount, "start_at_age": age,
            "duration_years": dur, "inflation_indexed": indexed, "tax_rate": tax}


def gen_deterministic():
    """sigma = 0 scenarios (RNG-independent).  Scenario VALUES are those the reference's own
    tests use (tests/test_simulation_correctness.py:84-133, :198-217, :363-404, :407-493,
    :496-602, :665-734); expected outputs come from running the reference here."""
    cases = []

    def add(name, cfgd, wm):
        sim = make_sim(cfgd)
        r = sim._run_single_simulation_path(wm, path_seed=1)
        cases.append({"name": name, "cfg": cfgd, "working_months": wm, "result": result_to_jsonable(r)})

    add("partial_year_inflation_accrual", base_test_config(
        inflation_rate_mean=0.06, inflation_rate_volatility=0.0, inv1_returns_volatility=0.0,
        inv2_premium_over_inflation_volatility=0.0, inv1_returns_mean=0.0,
        inv2_premium_over_inflation_mean=0.0, monthly_expenses=0.0, retirement_years=1, seed=7), 13)
    add("partial_year_equal_retirement_balance", base_test_config(
        initial_balance=100_000.0, monthly_expenses=1_000.0, retirement_years=1, **ZERO, **NOTAX), 13)
    add("allocation_weights_conserve", base_test_config(
        initial_balance=100_000.0, allocation_inv1_pct=0.333333, monthly_expenses=0.0,
        retirement_years=1, **ZERO), 0)
    pens = base_test_config(current_age=40.0, initial_balance=80_000.0, monthly_expenses=1000.0,
                            retirement_years=10, other_income_streams=[pension(1000.0, 65.0)],
                            seed=1, num_simulations_main=5, **ZERO, **NOTAX)
    add("income_stream_starts_at_age", pens, 240)
    add("income_stream_starts_at_age_no_pension", dict(pens, other_income_streams=[]), 240)
    add("income_stream_fractional_age", base_test_config(
        current_age=60.0, initial_balance=6_000.0, monthly_expenses=1_000.0, retirement_years=2,
        other_income_streams=[pension(1_000.0, 60.5, name="Midyear pension")], seed=3, **ZERO, **NOTAX), 0)
    dep = base_test_config(current_age=60.0, initial_balance=12_000.0, monthly_expenses=1_000.0,
                           retirement_years=10, other_income_streams=[pension(1_000.0, 61.0)],
                           seed=1, **ZERO, **NOTAX)
    add("pension_covers_after_depletion", dep, 0)
    add("pension_covers_after_depletion_no_pension", dict(dep, other_income_streams=[]), 0)
    add("withdrawal_rate_first_year", base_test_config(
        initial_balance=200_000.0, monthly_expenses=1_000.0, retirement_years=5, seed=1, **ZERO, **NOTAX), 0)
    add("real_wr_flat_deterministic_inflation", base_test_config(
        initial_balance=240_000.0, monthly_expenses=1_000.0, retirement_years=8,
        inflation_rate_mean=0.06, inflation_rate_volatility=0.0, inv1_returns_mean=0.06,
        inv1_returns_volatility=0.0, inv2_premium_over_inflation_mean=0.0,
        inv2_premium_over_inflation_volatility=0.0, seed=2, **NOTAX), 0)
    add("years_to_ruin", base_test_config(
        initial_balance=5_000.0, monthly_expenses=2_000.0, retirement_years=10, seed=9, **ZERO, **NOTAX), 0)
    common = dict(initial_balance=100_000.0, monthly_contribution=0.0, monthly_expenses=0.0,
                  retirement_years=1, allocation_inv1_pct=0.50, inv1_returns_mean=0.0,
                  inv1_returns_volatility=0.0, inv1_use_realized_gains_tax_system=False,
                  inv1_realized_gains_tax_rate=0.0, inv2_premium_over_inflation_mean=1.0,
                  inv2_premium_over_inflation_volatility=0.0, inv2_use_realized_gains_tax_system=True,
                  inv2_realized_gains_tax_rate=0.0, inflation_rate_mean=0.0,
                  inflation_rate_volatility=0.0, seed=11)
    add("annual_tax_excludes_transfers_no_tax", base_test_config(**common, inv1_annual_tax_on_gains_rate=0.0), 12)
    add("annual_tax_excludes_transfers_full_tax", base_test_config(**common, inv1_annual_tax_on_gains_rate=1.0), 12)
    add("retirement_does_not_split_tax_period", base_test_config(
        initial_balance=100.0, monthly_expenses=0.0, retirement_years=1, allocation_inv1_pct=1.0,
        inv1_returns_mean=0.12, inv1_returns_volatility=0.0, inv1_use_realized_gains_tax_system=False,
        inv1_annual_tax_on_gains_rate=0.50, inv2_premium_over_inflation_mean=0.0,
        inv2_premium_over_inflation_volatility=0.0, inv2_use_realized_gains_tax_system=False,
        inv2_annual_tax_on_gains_rate=0.0, inflation_rate_mean=0.0, inflation_rate_volatility=0.0, seed=12), 13)
    add("pre_retirement_tax_failure", base_test_config(
        initial_balance=100_000.0, monthly_expenses=500.0, retirement_years=3, allocation_inv1_pct=0.5,
        inv1_returns_mean=3.0, inv1_returns_volatility=0.0, inv1_use_realized_gains_tax_system=False,
        inv1_annual_tax_on_gains_rate=1.0, inv2_premium_over_inflation_mean=-0.99,
        inv2_premium_over_inflation_volatility=0.0, inv2_use_realized_gains_tax_system=True,
        inv2_realized_gains_tax_rate=0.5, inflation_rate_mean=0.0, inflation_rate_volatility=0.0), 14)
    # extra deterministic edge cases (not in the reference's tests): non-indexed stream lock,
    # finite duration, terminal partial tax period with annual taxes, contribution growth.
    add("nonindexed_stream_finite_duration", base_test_config(
        current_age=50.0, initial_balance=300_000.0, monthly_contribution=1500.0,
        contribution_growth_rate_annual=0.03, monthly_expenses=3_000.0, retirement_years=12,
        inflation_rate_mean=0.04, inflation_rate_volatility=0.0, inv1_returns_mean=0.07,
        inv1_returns_volatility=0.0, inv2_premium_over_inflation_mean=0.01,
        inv2_premium_over_inflation_volatility=0.0,
        inv1_use_realized_gains_tax_system=False, inv1_annual_tax_on_gains_rate=0.15,
        inv2_use_realized_gains_tax_system=True, inv2_realized_gains_tax_rate=0.2,
        other_income_streams=[pension(900.0, 55.25, indexed=False, tax=0.1, dur=4, name="Annuity"),
                              pension(1200.0, 60.0, indexed=True, tax=0.2, name="State")]), 31)
    return cases


# ---------------------------------------------------------------------------------------
def scenario_table():
    c1 = load_json("config.json")
    c3 = load_json("jorge.json")
    s60 = dict(c1, initial_balance=2.0e6, inv1_returns_volatility=0.15, equity_inflation_correlation=0.3)
    annual = base_test_config(
        initial_balance=1_200_000.0, monthly_contribution=2_000.0, contribution_growth_rate_annual=0.02,
        monthly_expenses=4_500.0, retirement_years=25, allocation_inv1_pct=0.7,
        inv1_returns_mean=0.09, inv1_returns_volatility=0.18, inv1_use_realized_gains_tax_system=False,
        inv1_annual_tax_on_gains_rate=0.15, inv2_premium_over_inflation_mean=0.015,
        inv2_premium_over_inflation_volatility=0.03, inv2_use_realized_gains_tax_system=False,
        inv2_annual_tax_on_gains_rate=0.10, inflation_rate_mean=0.035, inflation_rate_volatility=0.02,
        equity_inflation_correlation=-0.25,
        other_income_streams=[pension(1500.0, 67.0, indexed=True, tax=0.15),
                              pension(800.0, 50.0, indexed=False, tax=0.25, dur=10, name="Rent")])
    mixed = dict(annual, inv2_use_realized_gains_tax_system=True, inv2_realized_gains_tax_rate=0.15,
                 inv1_annual_tax_on_gains_rate=0.2, equity_inflation_correlation=0.6)
    failing = base_test_config(
        initial_balance=800_000.0, monthly_contribution=500.0, monthly_expenses=3_500.0,
        retirement_years=30, inv1_returns_volatility=0.2, inv1_use_realized_gains_tax_system=True,
        inv1_realized_gains_tax_rate=0.15, inv2_use_realized_gains_tax_system=True,
        inv2_realized_gains_tax_rate=0.15,
        other_income_streams=[pension(1000.0, 70.0, indexed=False, tax=0.1, dur=8)])
    pretax = base_test_config(
        initial_balance=100_000.0, monthly_expenses=500.0, retirement_years=3, allocation_inv1_pct=0.5,
        inv1_returns_mean=1.5, inv1_returns_volatility=0.5, inv1_use_realized_gains_tax_system=False,
        inv1_annual_tax_on_gains_rate=1.0, inv2_premium_over_inflation_mean=-0.9,
        inv2_premium_over_inflation_volatility=0.3, inv2_use_realized_gains_tax_system=True,
        inv2_realized_gains_tax_rate=0.5, inflation_rate_mean=0.0, inflation_rate_volatility=0.0)
    return [
        ("C1_config_json_wm233", c1, 233, "final", 12345, 12),
        ("C1_config_json_wm0", c1, 0, "search", 12345, 6),
        ("C1_config_json_wm13", c1, 13, "search", 12345, 6),
        ("C3_jorge_wm75", c3, 75, "final", 12345, 12),
        ("C3_jorge_wm75_rho03", dict(c3, equity_inflation_correlation=0.3), 75, "final", 12345, 8),
        ("S60_wm120", s60, 120, "final", 12345, 12),
        ("ANNUAL_wm50", annual, 50, "final", 777, 12),
        ("MIXED_wm36", mixed, 36, "search", 777, 12),
        ("FAILING_wm24", failing, 24, "final", 99, 16),
        ("PRETAX_wm14", pretax, 14, "final", 4242, 16),
        ("RHO_plus1", base_test_config(equity_inflation_correlation=1.0, retirement_years=15), 18, "final", 5, 6),
        ("RHO_minus1", base_test_config(equity_inflation_correlation=-1.0, retirement_years=15), 18, "final", 5, 6),
    ]


def gen_injected_paths():
    groups = []
    for name, cfgd, wm, stream, seed, n in scenario_table():
        sim = make_sim(cfgd, seed=seed)
        getattr(sim, f"use_{stream}_seeds")()
        inject_engine_shocks(sim, seed)
        paths = []
        for i in range(n):
            paths.append(result_to_jsonable(sim._run_single_simulation_path(wm, i)))
        groups.append({"name": name, "cfg": cfgd, "working_months": wm, "stream": stream,
                       "seed": seed, "path_begin": 0, "n_paths": n, "results": paths})
        print(f"  injected {name}: {n} paths, success={sum(p['Success'] for p in paths)}/{n}", flush=True)
    return groups


def gen_fuzz():
    """Random small scenarios that reach rarely-taken branches (zero allocations, 100 % tax,
    failing annual-tax payments, many streams).  Engine shocks injected."""
    rng = np.random.default_rng(31337)
    groups = []
    for k in range(48):
        n_streams = int(rng.integers(0, 5))
        streams = []
        cur_age = float(rng.choice([30.0, 45.5, 58.25, 64.0]))
        for s in range(n_streams):
            streams.append(pension(
                float(rng.choice([0.0, rng.uniform(100, 4000)])), float(cur_age + rng.uniform(-2, 12)),
                indexed=bool(rng.integers(2)), tax=float(rng.choice([0.0, rng.uniform(0, 0.5), 1.0])),
                dur=(None if rng.random() < 0.4 else int(rng.integers(0, 6))), name=f"s{s}"))
        cfgd = base_test_config(
            initial_balance=float(rng.choice([0.0, rng.uniform(1e3, 1e5), rng.uniform(1e5, 2e6)])),
            monthly_contribution=float(rng.choice([0.0, rng.uniform(0, 8000)])),
            contribution_growth_rate_annual=float(rng.choice([0.0, rng.uniform(0, 0.1)])),
            monthly_expenses=float(rng.choice([0.0, rng.uniform(500, 9000)])),
            current_age=cur_age,
            retirement_years=int(rng.integers(1, 9)),
            allocation_inv1_pct=float(rng.choice([0.0, 1.0, rng.uniform(0, 1)])),
            inv1_returns_mean=float(rng.uniform(-0.3, 0.5)),
            inv1_returns_volatility=float(rng.choice([0.0, rng.uniform(0, 0.6)])),
            inv1_annual_tax_on_gains_rate=float(rng.choice([0.0, rng.uniform(0, 1), 1.0])),
            inv1_realized_gains_tax_rate=float(rng.choice([0.0, rng.uniform(0, 1), 1.0])),
            inv1_use_realized_gains_tax_system=bool(rng.integers(2)),
            inv2_premium_over_inflation_mean=float(rng.uniform(-0.2, 0.4)),
            inv2_premium_over_inflation_volatility=float(rng.choice([0.0, rng.uniform(0, 0.3)])),
            inv2_annual_tax_on_gains_rate=float(rng.choice([0.0, rng.uniform(0, 1), 1.0])),
            inv2_realized_gains_tax_rate=float(rng.choice([0.0, rng.uniform(0, 1), 1.0])),
            inv2_use_realized_gains_tax_system=bool(rng.integers(2)),
            inflation_rate_mean=float(rng.uniform(-0.05, 0.25)),
            inflation_rate_volatility=float(rng.choice([0.0, rng.uniform(0, 0.1)])),
            equity_inflation_correlation=float(rng.choice([0.0, -1.0, 1.0, rng.uniform(-1, 1)])),
            other_income_streams=streams,
        )
        wm = int(rng.choice([0, 1, 11, 12, 13, 24, int(rng.integers(0, 50))]))
        seed = int(rng.integers(0, 2**40))
        stream = "search" if k % 2 else "final"
        sim = make_sim(cfgd, seed=seed)
        getattr(sim, f"use_{stream}_seeds")()
        inject_engine_shocks(sim, seed)
        pb = int(rng.choice([0, 2**32 - 2, 2**33 + 5]))  # exercises the 64-bit path counter
        n = 4
        paths = [result_to_jsonable(sim._run_single_simulation_path(wm, pb + i)) for i in range(n)]
        groups.append({"name": f"fuzz{k:02d}", "cfg": cfgd, "working_months": wm, "stream": stream,
                       "seed": seed, "path_begin": pb, "n_paths": n, "results": paths})
    nfail = sum(1 for g in groups for p in g["results"] if not p["Success"])
    pre = sum(1 for g in groups for p in g["results"] if p["YearsToRuin"] == 0.0)
    print(f"  fuzz: {len(groups)} scenarios, {nfail} failing paths, {pre} pre-retirement tax failures", flush=True)
    return groups + gen_many_streams()


def gen_many_streams():
    """Scenarios with MORE other_income_streams than the engine's by-value block holds (16): the reference takes any list
    (backend/config.py:99; loops backend/simulation.py:602-621, 649-677).  Indexed / frozen, finite / endless / zero-length,
    overlapping windows, starts before / at / after retirement.  Appended to paths_fuzz.json behind the 48 random
    scenarios (own generator state: those stay bit-identical)."""
    rng = np.random.default_rng(20261005)

    def streams(n, frozen_share, cur_age, horizon):
        out = []
        for s in range(n):
            indexed = bool(rng.random() >= frozen_share)
            dur = None if rng.random() < 0.3 else int(rng.integers(0, max(2, horizon // 2)))
            out.append(pension(float(rng.uniform(20.0, 600.0)), float(cur_age + rng.uniform(-3.0, horizon * 0.8)),
                               indexed=indexed, tax=float(rng.choice([0.0, 0.15, rng.uniform(0, 0.45)])), dur=dur, name=f"s{s}"))
        return out

    c3 = load_json("jorge.json")
    table = [
        # name, base cfg, n_streams, share of non-indexed, working months, stream, seed, paths
        ("streams17_boundary", dict(c3, equity_inflation_correlation=0.3), 17, 0.5, 75, "final", 4711, 6),
        ("streams20_mixed", base_test_config(
            initial_balance=650_000.0, monthly_contribution=800.0, monthly_expenses=4_600.0, current_age=52.0,
            retirement_years=30, inv1_returns_volatility=0.18, inv1_use_realized_gains_tax_system=True,
            inv1_realized_gains_tax_rate=0.15, inv2_use_realized_gains_tax_system=True, inv2_realized_gains_tax_rate=0.2,
            inflation_rate_volatility=0.02, equity_inflation_correlation=-0.2), 20, 0.5, 30, "search", 20, 12),
        ("streams33_all_frozen", base_test_config(
            initial_balance=520_000.0, monthly_expenses=4_000.0, current_age=60.0, retirement_years=25,
            inv1_returns_volatility=0.2, inflation_rate_mean=0.05, inflation_rate_volatility=0.03), 33, 1.0, 0, "final", 33, 8),
        ("streams40_mixed_annual_tax", base_test_config(
            initial_balance=600_000.0, monthly_contribution=1_500.0, contribution_growth_rate_annual=0.03,
            monthly_expenses=7_500.0, current_age=45.5, retirement_years=35, allocation_inv1_pct=0.7,
            inv1_returns_mean=0.09, inv1_returns_volatility=0.17, inv1_use_realized_gains_tax_system=False,
            inv1_annual_tax_on_gains_rate=0.15, inv2_use_realized_gains_tax_system=True, inv2_realized_gains_tax_rate=0.15,
            inflation_rate_mean=0.035, inflation_rate_volatility=0.015, equity_inflation_correlation=0.4), 40, 0.7, 49, "final", 40, 12),
    ]
    groups = []
    for name, cfgd, n_streams, frozen_share, wm, stream, seed, n in table:
        cfgd = dict(cfgd, other_income_streams=streams(n_streams, frozen_share, cfgd["current_age"], cfgd["retirement_years"]))
        sim = make_sim(cfgd, seed=seed)
        getattr(sim, f"use_{stream}_seeds")()
        inject_engine_shocks(sim, seed)
        paths = [result_to_jsonable(sim._run_single_simulation_path(wm, i)) for i in range(n)]
        groups.append({"name": name, "cfg": cfgd, "working_months": wm, "stream": stream,
                       "seed": seed, "path_begin": 0, "n_paths": n, "results": paths})
        print(f"  {name}: {n_streams} streams ({sum(not s['inflation_indexed'] for s in cfgd['other_income_streams'])} frozen), "
              f"success={sum(p['Success'] for p in paths)}/{n}", flush=True)
    return groups


def gen_numpy_native(outdir):
    """Reference with its OWN NumPy RNG; shock arrays stored so the oracle/kernel can replay them."""
    arrays = {}
    meta = []
    for name, cfgd, wm, seed, n in [("C1_config_json_wm233", load_json("config.json"), 233, 12345, 4),
                                    ("C3_jorge_wm75_rho03", dict(load_json("jorge.json"), equity_inflation_correlation=0.3), 75, 2024, 4)]:
        sim = make_sim(cfgd, seed=seed)
        sim.use_final_seeds()
        seeds = sim._path_seeds(n)
        total = wm + cfgd["retirement_years"] * 12
        sh = np.stack([sim._draw_shock_path(max(total, 1), s) for s in seeds])
        res = [result_to_jsonable(sim._run_single_simulation_path(wm, s)) for s in seeds]
        arrays[name] = sh
        meta.append({"name": name, "cfg": cfgd, "working_months": wm, "main_seed": seed,
                     "numpy_path_seeds": [int(s) for s in seeds], "results": res})
    np.savez_compressed(os.path.join(outdir, "numpy_native_shocks.npz"), **arrays)
    return meta


def gen_numpy_native_batch():
    """Reference driven by its OWN NumPy RNG, no injection: the literal-seed fixtures for the
    engine's rng="numpy" mode.  Per-path results with the uint32 path seeds, and a native search."""
    out = {"paths": [], "search": []}
    tab = {t[0]: t for t in scenario_table()}
    for name, seed, n in [("C1_config_json_wm233", 12345, 24), ("C3_jorge_wm75_rho03", 2024, 24),
                          ("FAILING_wm24", 99, 32), ("ANNUAL_wm50", 777, 24), ("PRETAX_wm14", 4242, 16)]:
        _, cfgd, wm, _stream, _seed, _n = tab[name]
        for stream in ("final", "search"):
            sim = make_sim(dict(cfgd, num_processes=1), seed=seed)
            getattr(sim, f"use_{stream}_seeds")()
            first = sim._path_seeds(7)          # an earlier batch of another size shifts the spawn offset (:192-199)
            seeds = sim._path_seeds(n)
            res = [result_to_jsonable(sim._run_single_simulation_path(wm, s)) for s in seeds]
            out["paths"].append({"name": name, "cfg": cfgd, "working_months": wm, "main_seed": seed, "stream": stream,
                                 "earlier_batch": 7, "earlier_seeds": [int(x) for x in first],
                                 "numpy_path_seeds": [int(x) for x in seeds], "results": res})
        print(f"  native batch {name}: ok", flush=True)
    for name, cfgd, seed in [("C3_jorge_search_native", dict(load_json("jorge.json"), num_processes=1), 12345)]:
        sim = make_sim(cfgd, seed=seed)
        events = []
        months, prob, curve = sim.find_minimum_working_months(verbose=False, progress_callback=events.append)
        sim.use_final_seeds()
        t = sim.run_monte_carlo_simulations(months, cfgd["num_simulations_main"])
        out["search"].append({"name": name, "cfg": cfgd, "seed": seed, "months": months, "probability": prob,
                              "search_curve": curve, "events": events,
                              "final_success_probability": sim._success_probability(t[0]),
                              "final_success": [bool(x) for x in t[0]["Success"].tolist()],
                              "final_trajectory_percentiles": frame_to_jsonable(t[1]),
                              "final_wr_observation_counts": t[6]})
        print(f"  native search {name}: {months} months @ {prob:.2f}%, final {out['search'][-1]['final_success_probability']:.2f}%", flush=True)
    return out


def gen_config_schema():
    """The reference's pydantic schema (field names, bounds, defaults, aliases, required set) as data."""
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
    return strip(json.loads(json.dumps(RefConfig.model_json_schema())))


def frame_to_jsonable(df):
    if df is None:
        return None
    return {"columns": [float(c) for c in df.columns], "values": df.values.tolist()}


def gen_aggregation():
    """a12: the 7-tuple of run_monte_carlo_simulations on a 200-path batch (engine shocks)."""
    out = []
    for name, cfgd, wm, stream, seed, n in [
        ("C3_jorge_wm75", load_json("jorge.json"), 75, "final", 12345, 200),
        ("FAILING_wm24", [t for t in scenario_table() if t[0] == "FAILING_wm24"][0][1], 24, "final", 99, 200),
    ]:
        sim = make_sim(cfgd, seed=seed)
        getattr(sim, f"use_{stream}_seeds")()
        inject_engine_shocks(sim, seed)
        t = sim.run_monte_carlo_simulations(wm, n)
        summary = t[0]
        import pandas as pd
        # which columns did pandas sample? (trajectory_df.sample(n=5, axis=1, random_state=main_seed), :1068-1072)
        cols = pd.DataFrame(np.zeros((2, n))).sample(n=min(n, 5), axis=1, random_state=sim.main_seed).columns.tolist()
        out.append({
            "name": name, "cfg": cfgd, "working_months": wm, "stream": stream, "seed": seed, "n_paths": n,
            "success_probability": sim._success_probability(summary),
            "median_first_year_withdrawal_rate": ref_sim.median_first_year_withdrawal_rate(summary),
            "summary": {k: [float(x) if k != "Success" else bool(x) for x in summary[k].tolist()] for k in SUMMARY_KEYS},
            "trajectory_percentiles": frame_to_jsonable(t[1]),
            "sample_trajectories": t[2],
            "wr_percentiles": frame_to_jsonable(t[3]),
            "real_trajectory_percentiles": frame_to_jsonable(t[4]),
            "sample_real_trajectories": t[5],
            "wr_observation_counts": t[6],
            "sampled_columns": [int(c) for c in cols],
        })
        print(f"  aggregation {name}: p={out[-1]['success_probability']:.2f}%", flush=True)
    return out


def gen_search():
    """Search driver on the reference with engine shocks (jorge.json, N=100/probe)."""
    out = []
    for name, cfgd, seed in [("C3_jorge_search", dict(load_json("jorge.json"), num_processes=1), 12345),
                             ("BASE_search", base_test_config(initial_balance=100_000.0, monthly_contribution=3_000.0,
                                                              monthly_expenses=5_000.0, retirement_years=30,
                                                              inv1_returns_mean=0.10, inv1_returns_volatility=0.12,
                                                              inflation_rate_mean=0.04, inflation_rate_volatility=0.015,
                                                              num_simulations_search=60, target_probability=85.0), 123),
                             # round 2: more of the search's branches on the reference itself —
                             # annual-gains tax on both assets + income streams, a late starting month
                             ("ANNUAL_TAX_search", base_test_config(initial_balance=150_000.0, monthly_contribution=2_500.0,
                                                                    contribution_growth_rate_annual=0.03, monthly_expenses=6_000.0,
                                                                    retirement_years=25, inv1_annual_tax_on_gains_rate=0.15,
                                                                    inv2_annual_tax_on_gains_rate=0.15, inv1_returns_mean=0.09,
                                                                    inv1_returns_volatility=0.16, inflation_rate_mean=0.035,
                                                                    inflation_rate_volatility=0.012, equity_inflation_correlation=-0.4,
                                                                    num_simulations_search=40, target_probability=90.0,
                                                                    starting_working_months_search=60,
                                                                    other_income_streams=[
                                                                        {"name": "pension", "monthly_amount_today": 1500.0, "start_at_age": 67.0,
                                                                         "duration_years": None, "inflation_indexed": True, "tax_rate": 0.1},
                                                                        {"name": "rent", "monthly_amount_today": 800.0, "start_at_age": 45.5,
                                                                         "duration_years": 12, "inflation_indexed": False, "tax_rate": 0.2}]), 2024),
                             # the target is met without working at all
                             ("RICH_search", base_test_config(initial_balance=5_000_000.0, monthly_expenses=3_000.0, retirement_years=20,
                                                              num_simulations_search=30, target_probability=95.0), 7),
                             # the target is out of reach: the search runs into its upper bound
                             ("HOPELESS_search", base_test_config(initial_balance=1_000.0, monthly_contribution=10.0, monthly_expenses=9_000.0,
                                                                  retirement_years=30, inv1_returns_mean=0.01, inv1_returns_volatility=0.05,
                                                                  num_simulations_search=12, target_probability=99.0), 11),
                             # few paths per probe and a high target: the probabilities are coarse and not monotone,
                             # the verification window below the bisection's answer matters
                             ("COARSE_search", base_test_config(initial_balance=80_000.0, monthly_contribution=1_800.0, monthly_expenses=3_500.0,
                                                                retirement_years=28, inv1_returns_volatility=0.22, allocation_inv1_pct=0.85,
                                                                inv1_realized_gains_tax_rate=0.15, inv1_use_realized_gains_tax_system=True,
                                                                num_simulations_search=16, target_probability=93.0), 31337)]:
        sim = make_sim(cfgd, seed=seed)
        inject_engine_shocks(sim, seed)
        events = []
        months, prob, curve = sim.find_minimum_working_months(verbose=False, progress_callback=events.append)
        out.append({"name": name, "cfg": cfgd, "seed": seed, "months": months, "probability": prob,
                    "search_curve": curve, "events": events})
        print(f"  search {name}: {months} months @ {prob:.2f}% in {len(curve)} probes", flush=True)
    return out


def gen_server_stream():
    """(f)-3: the reference's own server flows, driven through FastAPI's TestClient with the simulator class
    swapped for a subclass that consumes the engine's shocks: POST /api/simulate/stream (SSE events incl.
    the final `result` payload, server.py:322-413) and POST /api/simulate (server.py:301-319), plus the
    7-tuple the final run_monte_carlo_simulations returned (the input of _build_result, server.py:423-434)."""
    import server as ref_server
    from fastapi.testclient import TestClient

    captured = {}

    class InjectedSim(ref_sim.RetirementMonteCarloSimulator):
        def __init__(self, params_model, main_seed_override=None):
            super().__init__(params_model, main_seed_override)
            inject_engine_shocks(self, self.main_seed)

        def run_monte_carlo_simulations(self, working_months, num_simulations):
            t = super().run_monte_carlo_simulations(working_months, num_simulations)
            if self._stream_name == "final":
                captured["tuple"] = t
                captured["args"] = (int(working_months), int(num_simulations))
            return t

    ref_server.RetirementMonteCarloSimulator = InjectedSim
    client = TestClient(ref_server.app)
    failing = [t for t in scenario_table() if t[0] == "FAILING_wm24"][0][1]
    cases = [
        ("jorge_search_then_final", dict(load_json("jorge.json"), num_processes=1, seed=12345, num_simulations_search=100,
                                         num_simulations_main=300), None),
        ("failing_override_wm24", dict(failing, num_processes=1, seed=99, num_simulations_main=200), 24),
        ("unreachable_target", base_test_config(initial_balance=1_000.0, monthly_expenses=50_000.0, target_probability=99.0,
                                                num_simulations_search=20, num_simulations_main=20), None),
    ]
    out = []
    for name, cfgd, override in cases:
        body = {"config": cfgd}
        if override is not None:
            body["working_months_override"] = override
        captured.clear()
        r = client.post("/api/simulate/stream", json=body)
        assert r.status_code == 200, r.text
        events = [json.loads(line[len("data: "):]) for line in r.text.split("\n") if line.startswith("data: ")]
        t = captured.get("tuple")
        entry = {"name": name, "cfg": cfgd, "working_months_override": override, "events": events}
        if t is not None:
            entry["final_args"] = list(captured["args"])
            entry["final_tuple"] = {
                "summary": {k: [float(x) if k != "Success" else bool(x) for x in t[0][k].tolist()] for k in SUMMARY_KEYS},
                "trajectory_percentiles": frame_to_jsonable(t[1]), "sample_trajectories": t[2],
                "wr_percentiles": frame_to_jsonable(t[3]), "real_trajectory_percentiles": frame_to_jsonable(t[4]),
                "sample_real_trajectories": t[5], "wr_observation_counts": t[6],
            }
        r2 = client.post("/api/simulate", json=body)
        entry["simulate_status"] = r2.status_code
        entry["simulate_body"] = r2.json()
        out.append(entry)
        print(f"  server {name}: {len(events)} events, last={events[-1]['type']}, /api/simulate -> {r2.status_code}", flush=True)
    return out


def gen_10k(outdir):
    """The metric's 10k-path config: config.json, wm=233, final stream, seed 12345, engine shocks."""
    cfgd = dict(load_json("config.json"), num_processes=1)
    seed, wm, n = 12345, 233, 10_000
    sim = make_sim(cfgd, seed=seed)
    sim.use_final_seeds()
    inject_engine_shocks(sim, seed)
    t0 = time.time()
    cols = {k: np.empty(n, dtype=np.float64) for k in SUMMARY_KEYS if k != "Success"}
    succ = np.zeros(n, dtype=np.uint8)
    for i in range(n):
        r = sim._run_single_simulation_path(wm, i)
        for k in cols:
            cols[k][i] = r[k]
        succ[i] = r["Success"]
        if i % 2000 == 0:
            print(f"  10k: {i}/{n} ({time.time() - t0:.0f}s)", flush=True)
    np.savez_compressed(
        os.path.join(outdir, "metric_10k_config_json.npz"),
        success_bits=np.packbits(succ),
        **{k.replace(" ", "_"): v for k, v in cols.items()},
    )
    meta = {"cfg": cfgd, "working_months": wm, "stream": "final", "seed": seed, "n_paths": n,
            "success_count": int(succ.sum()), "success_probability_pct": float(succ.mean() * 100.0),
            "reference_seconds_1core": time.time() - t0}
    print(f"  10k: success {meta['success_count']}/{n} in {meta['reference_seconds_1core']:.0f}s", flush=True)
    return meta


def gen_native_stats():
    """Reference with NumPy's own RNG: success % (sanity bound only, not bitwise parity)."""
    out = []
    c1 = load_json("config.json")
    for name, cfgd, wm, n in [("C1_config_json_wm233", c1, 233, 4000), ("C3_jorge_wm75", load_json("jorge.json"), 75, 4000),
                              ("S60_wm120", dict(c1, initial_balance=2.0e6, inv1_returns_volatility=0.15, equity_inflation_correlation=0.3), 120, 4000)]:
        cfgd = dict(cfgd, num_processes=8)
        sim = make_sim(cfgd, seed=12345)
        sim.use_final_seeds()
        t0 = time.time()
        summary = sim.run_monte_carlo_simulations(wm, n)[0]
        out.append({"name": name, "cfg": cfgd, "working_months": wm, "n_paths": n, "main_seed": 12345,
                    "success_probability_pct": sim._success_probability(summary),
                    "median_final_balance": float(summary["Final Balance"].median()),
                    "median_start_balance": float(summary["Start Balance"].median()),
                    "seconds_8proc": time.time() - t0})
        print(f"  native {name}: {out[-1]['success_probability_pct']:.2f}% ({out[-1]['seconds_8proc']:.0f}s)", flush=True)
    return out


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as fh:
        json.dump(obj, fh, indent=None, separators=(",", ":"))
        fh.write("\n")
    print(f"wrote {name} ({os.path.getsize(os.path.join(HERE, name)) / 1024:.0f} KiB)", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-10k", action="store_true")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    O.build()
    only = set(args.only.split(",")) if args.only else None

    def want(k):
        return only is None or k in only

    if want("schema"):
        dump("config_schema.json", gen_config_schema())
    if want("helpers"):
        dump("helpers.json", gen_helpers())
    if want("deterministic"):
        dump("paths_deterministic.json", gen_deterministic())
    if want("injected"):
        dump("paths_injected.json", gen_injected_paths())
    if want("fuzz"):
        dump("paths_fuzz.json", gen_fuzz())
    if want("native_shocks"):
        dump("numpy_native_paths.json", gen_numpy_native(HERE))
    if want("aggregation"):
        dump("aggregation.json", gen_aggregation())
    if want("search"):
        dump("search.json", gen_search())
    if want("server"):
        dump("server_stream.json", gen_server_stream())
    if want("native_batch"):
        dump("numpy_native_batch.json", gen_numpy_native_batch())
    if want("native_stats"):
        dump("numpy_native_stats.json", gen_native_stats())
    if want("10k") and not args.skip_10k:
        dump("metric_10k_config_json.json", gen_10k(HERE))


if __name__ == "__main__":
    main()
