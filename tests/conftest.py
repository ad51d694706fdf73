"""Shared pytest fixtures.  `gpu`-marked tests need a real MI355X; everything else runs on CPU."""

from __future__ import annotations

import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name: str):
    with open(os.path.join(GOLDEN, name)) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure), built on demand with gcc."""
    from oracle import oracle as O

    O.build()
    return O


SUMMARY_MAP = {
    "Start Balance": "start_balance",
    "Final Balance": "final_balance",
    "YearsToRuin": "years_to_ruin",
    "First Year Gross Withdrawal": "first_year_gross_withdrawal",
    "First Year Real Gross Withdrawal": "first_year_real_gross_withdrawal",
    "Inflation At Retirement": "inflation_at_retirement",
}
STREAM_ID = {"search": 0, "final": 1}


def assert_same_float(a: float, b: float, what: str = ""):
    """Bit-level equality for doubles, with NaN == NaN."""
    if np.isnan(a) and np.isnan(b):
        return
    assert a == b, f"{what}: {a!r} != {b!r} (diff {a - b!r})"


def compare_batch_to_golden(res, results, exact: bool, rel: float = 0.0, abs_tol: float = 0.0):
    """Compare a run_batch-style dict of arrays with a list of reference result dicts."""
    n = len(results)
    for i, r in enumerate(results):
        assert bool(res["success"][i]) == bool(r["Success"]), f"path {i}: Success"
        for key, field in SUMMARY_MAP.items():
            got, exp = float(res[field][i]), float(r[key])
            if exact:
                assert_same_float(got, exp, f"path {i} {key}")
            else:
                _close(got, exp, rel, abs_tol, f"path {i} {key}")
        for key, field in (
            ("Trajectory", "trajectory"),
            ("RealTrajectory", "real_trajectory"),
            ("WithdrawalRateTrajectory", "withdrawal_rate_trajectory"),
        ):
            exp = np.asarray(r[key], dtype=np.float64)
            got = np.asarray(res[field])[:, i]
            assert got.shape == exp.shape, f"path {i} {key}: length {got.shape} vs {exp.shape}"
            if exact:
                assert np.array_equal(got, exp, equal_nan=True), f"path {i} {key}: {got} vs {exp}"
            else:
                assert np.array_equal(np.isnan(got), np.isnan(exp)), f"path {i} {key}: NaN pattern"
                np.testing.assert_allclose(got, exp, rtol=rel, atol=abs_tol, equal_nan=True,
                                           err_msg=f"path {i} {key}")
    assert n == len(results)


def _close(got, exp, rel, abs_tol, what):
    if np.isnan(exp) or np.isnan(got):
        assert np.isnan(exp) and np.isnan(got), f"{what}: {got} vs {exp}"
        return
    assert abs(got - exp) <= abs_tol + rel * abs(exp), f"{what}: {got!r} vs {exp!r}"
