"""Distribution of the engine's own random stream (Philox4x32-10 + Box-Muller + rho-mix, csrc/mcr_device.h), measured on
the device.  The reference pins no RNG-dependent number (its stochastic tests are statistical), so what the
stream owes is the distribution: standard normal marginals, the configured equity/inflation correlation, independence
across rows, paths and the two seed streams, and growth factors whose compounded mean is the configured arithmetic mean
(the reference's test_mean_realised_annual_return_matches_config, tests/test_simulation_correctness.py:137-164, run on
the device's draws).  Round 2 sized the stream's arithmetic to the 1e-9 path tolerance (one-step sqrt, one-constant
reductions: tests/test_gpu_math.py) — these checks are blind to anything that small by many orders of magnitude, which
is the point: the trims cannot have moved the distribution."""

from __future__ import annotations

import math

import numpy as np
import pytest
from scipy import stats

from monte_carlo_retirement_amd import _native as N
from monte_carlo_retirement_amd import engine as E
from monte_carlo_retirement_amd.params import arithmetic_to_log_params

pytestmark = pytest.mark.gpu

N_PATHS, N_ROWS = 4096, 600      # 2.46e6 rows = 7.4e6 normals per draw


@pytest.fixture(scope="module")
def rows():
    return E.draw_shocks_host(20260104, 1, 0, N_PATHS, N_ROWS, 0.3)


def test_marginals_are_standard_normal(rows):
    n = rows.shape[0] * rows.shape[1]
    for k, name in enumerate(("equity", "inflation", "premium")):
        z = rows[:, :, k].ravel()
        assert abs(z.mean()) < 5.0 / math.sqrt(n), (name, z.mean())
        assert abs(z.var() - 1.0) < 5.0 * math.sqrt(2.0 / n), (name, z.var())
        assert abs(stats.skew(z)) < 5.0 * math.sqrt(6.0 / n), (name, stats.skew(z))
        assert abs(stats.kurtosis(z)) < 5.0 * math.sqrt(24.0 / n), (name, stats.kurtosis(z))
        # Kolmogorov-Smirnov on a subsample (the full sample would flag differences far below anything a path can see)
        p = stats.kstest(z[::7], "norm").pvalue
        assert p > 1e-4, (name, p)
        # tails: P(|z| > 3) and P(|z| > 4) within 5 sigma of the normal's
        for t in (3.0, 4.0):
            pt = 2.0 * stats.norm.sf(t)
            got = float(np.mean(np.abs(z) > t))
            assert abs(got - pt) < 5.0 * math.sqrt(pt / n), (name, t, got, pt)
    assert np.abs(rows[:, :, (0, 2)]).max() < 6.77    # Box-Muller radius of a 32-bit uniform: sqrt(-2 ln 2^-33) = 6.764


def test_correlation_structure(rows):
    n = rows.shape[0] * rows.shape[1]
    tol = 5.0 / math.sqrt(n)
    ze, zi, zp = (rows[:, :, k].ravel() for k in range(3))
    assert abs(np.corrcoef(ze, zi)[0, 1] - 0.3) < tol          # :460-464 (rho-mix)
    assert abs(np.corrcoef(ze, zp)[0, 1]) < tol
    assert abs(np.corrcoef(zi, zp)[0, 1]) < tol
    # consecutive rows of a path (they share Philox blocks and Box-Muller pairs) and neighbouring paths are uncorrelated
    for k in range(3):
        a = rows[:, :, k]
        assert abs(np.corrcoef(a[:, :-1].ravel(), a[:, 1:].ravel())[0, 1]) < tol, k
        assert abs(np.corrcoef(a[:-1].ravel(), a[1:].ravel())[0, 1]) < tol, k
    assert abs(np.corrcoef(rows[:, :-1, 0].ravel(), rows[:, 1:, 2].ravel())[0, 1]) < tol   # cos / sin halves of one pair
    assert abs(np.corrcoef(rows[:, :-1, 2].ravel(), rows[:, 1:, 0].ravel())[0, 1]) < tol
    # squares too (a shared radius would show up here, not in the linear correlation)
    assert abs(np.corrcoef(rows[:, :-1, 2].ravel() ** 2, rows[:, 1:, 0].ravel() ** 2)[0, 1]) < tol
    assert abs(np.corrcoef(ze ** 2, zp ** 2)[0, 1]) < tol


def test_seed_streams_and_seeds_are_independent(rows):
    other_stream = E.draw_shocks_host(20260104, 0, 0, N_PATHS, N_ROWS, 0.3)    # search vs final stream (simulation.py:998-1001)
    other_seed = E.draw_shocks_host(20260105, 1, 0, N_PATHS, N_ROWS, 0.3)
    tol = 5.0 / math.sqrt(rows.shape[0] * rows.shape[1])
    for other in (other_stream, other_seed):
        assert not np.array_equal(other, rows)
        for k in range(3):
            assert abs(np.corrcoef(rows[:, :, k].ravel(), other[:, :, k].ravel())[0, 1]) < tol
    # and the same (seed, stream, path) is the same row whatever the batch it is drawn in
    again = E.draw_shocks_host(20260104, 1, 1000, 8, N_ROWS, 0.3)
    assert np.array_equal(again, rows[1000:1008])


def test_mean_realised_annual_return_matches_config(rows):
    """Reference :137-164 on the device's draws and the device's growth-factor function."""
    mean, vol = 0.12, 0.15
    mu_log, sigma_log = arithmetic_to_log_params(mean, vol)
    z = rows[:, :, 0].ravel()
    z = z[: (z.size // 12) * 12]
    gross = E.eval_helper_host(N.MCR_HELPER_MONTHLY_GROSS, None, np.column_stack((np.full(z.size, mu_log), np.full(z.size, sigma_log), z)))[:, 0]
    yearly = gross.reshape(-1, 12).prod(axis=1)                       # 2.0e5 simulated years
    se = yearly.std() / math.sqrt(yearly.size)
    assert abs(yearly.mean() - 1.0 - mean) < 5.0 * se, (yearly.mean() - 1.0, se)
    assert abs(yearly.mean() - 1.0 - mean) < 0.01                     # the reference's own bound
    assert abs(np.log(yearly).std() - sigma_log) < 5.0 * sigma_log / math.sqrt(2.0 * yearly.size)
