"""(f)-3 on the kernels: the run flow + response document of monte_carlo_retirement_amd.results against what the
REFERENCE's FastAPI app produced for the same scenarios and shocks (tests/golden/server_stream.json), and the
device-aggregated document for large batches against the per-path one."""

from __future__ import annotations

import json
import math

import numpy as np
import pytest

from conftest import load_golden
from monte_carlo_retirement_amd import Config
from monte_carlo_retirement_amd import aggregation as A
from monte_carlo_retirement_amd import engine as E
from monte_carlo_retirement_amd import results as R
from monte_carlo_retirement_amd.simulation import RetirementMonteCarloSimulator

pytestmark = pytest.mark.gpu

CASES = load_golden("server_stream.json")
# document values are rounded to cents (rates: 3 decimals): a 1e-9-relative kernel/CPython difference can move a
# value across a rounding boundary by one unit in the last place
CENT, REL = 0.0100001, 1e-9


def assert_doc_close(got, exp, path="$"):
    if isinstance(exp, dict):
        assert isinstance(got, dict) and set(got) == set(exp), path
        for k in exp:
            assert_doc_close(got[k], exp[k], f"{path}.{k}")
    elif isinstance(exp, list):
        assert isinstance(got, list) and len(got) == len(exp), path
        for i, (g, e) in enumerate(zip(got, exp)):
            assert_doc_close(g, e, f"{path}[{i}]")
    elif isinstance(exp, bool) or exp is None or isinstance(exp, str):
        assert got == exp and type(got) is type(exp), path
    else:
        assert got is not None and not isinstance(got, bool), path
        assert math.isclose(got, exp, rel_tol=REL, abs_tol=CENT), f"{path}: {got} != {exp}"


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"])
def test_run_scenario_on_the_gpu_reproduces_the_references_stream(case):
    events = []
    doc = R.run_scenario(Config(**case["cfg"]), case["working_months_override"], emit=events.append)
    got = json.loads(json.dumps(events, allow_nan=False))
    exp = case["events"]
    assert [e["type"] for e in got] == [e["type"] for e in exp]
    for g, e in zip(got, exp):
        if e["type"] == "result":
            assert_doc_close(g["data"], e["data"])
        else:
            assert g == e          # phases, every search probe (counts are exact), completion, errors
    if case["simulate_status"] == 200:
        assert_doc_close(json.loads(json.dumps(R.run_scenario(Config(**case["cfg"]), case["working_months_override"]))),
                         case["simulate_body"])
        assert doc is not None
    else:
        with pytest.raises(ValueError) as ei:
            R.run_scenario(Config(**case["cfg"]), case["working_months_override"])
        assert str(ei.value) == case["simulate_body"]["detail"]


def test_summary_stat_rows_kernel_matches_numpy():
    cfg = Config(**CASES[1]["cfg"])          # the failing scenario: both cohorts are non-trivial
    sim = RetirementMonteCarloSimulator(cfg)
    sim.use_final_seeds()
    n = 5003
    batch = E.DeviceBatch(sim._current_params(), 24, n, want="summary")
    batch.launch(sim._batch_rng(n), sim._stream_id, 0)
    batch.summary["start_balance"][7] = 0.0     # a path outside the withdrawal-rate cohort
    rows = A.summary_stat_rows(batch, n).cpu().numpy()[:, :n]
    s = {k: v.cpu().numpy() for k, v in batch.summary.items()}
    ok = batch.success.cpu().numpy().astype(bool)
    assert 0 < ok.sum() < n
    np.testing.assert_array_equal(rows[0], s["start_balance"])
    np.testing.assert_array_equal(rows[1], s["final_balance"])
    np.testing.assert_array_equal(rows[2], np.where(ok, s["final_balance"], np.nan))
    with np.errstate(divide="ignore", invalid="ignore"):
        rate = s["first_year_real_gross_withdrawal"] / s["start_balance"] * 100.0
    np.testing.assert_array_equal(rows[3], np.where(s["start_balance"] > 1e-6, rate, np.nan))
    assert np.isnan(rows[3][7])


@pytest.mark.parametrize("case", CASES[:2], ids=lambda c: c["name"])
@pytest.mark.parametrize("n", [1, 2, 1000, 40_000])
def test_compact_document_equals_the_per_path_document(case, n):
    """compact_result (device aggregates only) reports exactly the numbers build_result derives from the
    per-path frame with pandas, and its binned histograms equal np.histogram / the ruin-year list."""
    cfg = Config(**dict(case["cfg"], num_simulations_main=n))
    wm = case["final_args"][0]
    curve = [{"working_months": wm, "working_years": round(wm / 12, 1), "probability": 50.0}]
    a = RetirementMonteCarloSimulator(cfg)
    a.use_final_seeds()
    outputs = a.run_monte_carlo_simulations(wm, n)
    full = R.assemble_result(cfg, wm, outputs, curve)
    b = RetirementMonteCarloSimulator(cfg)
    b.use_final_seeds()
    compact = R.compact_result(cfg, b, wm, curve)
    for key in ("scenario", "summary", "trajectory", "trajectory_real", "withdrawal_rate", "search_curve", "reference_lines"):
        assert json.dumps(compact[key], allow_nan=True) == json.dumps(full[key], allow_nan=True), key
    assert compact["histogram"] == {"final_balances": [], "start_balances": [], "success_flags": []}
    summary = outputs[0]
    ok = summary["Success"].to_numpy()
    hb = compact["histogram_binned"]
    assert hb["successful_paths"] == int(ok.sum()) and hb["total_paths"] == n and len(hb["edges"]) == 61
    if ok.any():
        counts, edges = np.histogram(summary["Final Balance"].to_numpy()[ok], bins=60)
        assert hb["success_counts"] == counts.tolist() and hb["edges"] == edges.tolist()
    else:
        assert sum(hb["success_counts"]) == 0
    rh = compact["ruin_histogram"]
    assert rh["failure_count"] == full["ruin_histogram"]["failure_count"] == int((~ok).sum()) and rh["total_paths"] == n
    ytr = summary["YearsToRuin"].to_numpy()[~ok]
    # ruin bins (include/mcr.h): bin 0 = failed before retirement (YearsToRuin 0), bin k = ruin in retirement year k
    assert sum(rh["bins"]) == len(ytr) and rh["years_to_ruin"] == []


def test_compact_document_at_a_million_paths():
    """Size-independent properties at 10^6 paths (no per-path frame is ever built): the success share is the
    kernel's counter, percentiles are monotone and bracket the median, histogram counts add up."""
    with open(__import__("os").path.join(__import__("conftest").REPO, "scenarios", "config.json")) as fh:
        cfg = Config(**dict(json.load(fh), seed=12345, num_simulations_main=1_000_000))
    sim = RetirementMonteCarloSimulator(cfg)
    sim.use_final_seeds()
    doc = R.compact_result(cfg, sim, 233)
    s = doc["summary"]
    assert s["success_probability"] == pytest.approx(97.8, abs=0.2)       # 0.97801 at 10^6 paths
    p = [s["final_balance_percentiles"][k] for k in ("p1", "p5", "p10", "p25", "p50", "p75", "p90", "p95", "p99")]
    assert p == sorted(p) and p[0] >= 0.0
    hb = doc["histogram_binned"]
    assert sum(hb["success_counts"]) == hb["successful_paths"]
    assert s["success_probability"] == round(hb["successful_paths"] / 1e6 * 100.0, 2)
    assert doc["ruin_histogram"]["failure_count"] == 1_000_000 - hb["successful_paths"]
    assert len(json.dumps(doc)) < 200_000                                  # the document stays small
    assert s["median_final_balance_successful"] >= s["final_balance_percentiles"]["p50"]


def test_compact_document_on_the_bracketed_quantile_route():
    """2.2 million paths (long rows, the bracketed quantile route): the summary statistics and the bands of the compact document
    come from the sample-bracketed single-pass select, and still equal pandas on the per-path frame exactly."""
    case = CASES[1]                                   # the failing scenario: NaN-masked cohorts are non-trivial
    n = 2_200_000
    cfg = Config(**dict(case["cfg"], num_simulations_main=n))
    wm = case["final_args"][0]
    a = RetirementMonteCarloSimulator(cfg)
    a.use_final_seeds()
    full = R.assemble_result(cfg, wm, a.run_monte_carlo_simulations(wm, n), None)
    b = RetirementMonteCarloSimulator(cfg)
    b.use_final_seeds()
    compact = R.compact_result(cfg, b, wm, None)
    assert A.last_fallback_rows() >= 0                # the bracketed route was taken for the last select
    for key in ("summary", "trajectory", "trajectory_real", "withdrawal_rate", "reference_lines"):
        assert json.dumps(compact[key], allow_nan=True) == json.dumps(full[key], allow_nan=True), key
    assert compact["ruin_histogram"]["failure_count"] == full["ruin_histogram"]["failure_count"]


def test_document_preserves_the_exact_fractional_timeline():
    """The reference's test_api_preserves_exact_fractional_timeline (tests/test_simulation_correctness.py:781-817) on
    the GPU class + results.build_result: a 13-month working period puts the retirement marker at 13/12 years, not 1.1."""
    from test_simulator_cpu import _base_config

    config = _base_config(num_simulations_main=2, num_processes=1, retirement_years=1, monthly_expenses=0.0, seed=5)
    simulator = RetirementMonteCarloSimulator(config)
    result = R.build_result(config, simulator, required_w_months=13,
                            search_curve=[{"working_months": 13, "working_years": 1.1, "probability": 100.0}])
    retirement_year = 13 / 12
    assert result["trajectory"]["years"] == pytest.approx([0.0, 1.0, retirement_year, retirement_year + 1])
    assert result["withdrawal_rate"]["years"][0] == pytest.approx(retirement_year)
    assert result["reference_lines"][0]["year"] == pytest.approx(retirement_year)
    assert result["summary"]["working_period_is_estimate"] is True
