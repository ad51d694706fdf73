"""The final-balance histogram binned INSIDE the path kernel (mcr_outputs.hist_bins, ABI v6) against NumPy's
`np.histogram` of per-path outputs: of the kernel's own summary output (bit-identical final balances: every bin
equal), and of the CPU oracle's (reference semantics; a path within 1e-8 of an edge may sit on either side).

Reference: the CLI's chart of successful final balances, backend/plotting.py:44-59 (`plt.hist(..., bins=100)` =
`np.histogram`), SURVEY 8(e) "fixed log-spaced bins => single collective"."""

from __future__ import annotations

import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import REPO, load_golden
from monte_carlo_retirement_amd import Config, params_from_config
from monte_carlo_retirement_amd import _native as N
from monte_carlo_retirement_amd import engine as E

pytestmark = pytest.mark.gpu


def _cfg(name="config.json", **over):
    with open(os.path.join(REPO, "scenarios", name)) as fh:
        return Config(**dict(json.load(fh), seed=12345, **over))


def _s60():
    return _cfg(initial_balance=2.0e6, inv1_returns_volatility=0.15, equity_inflation_correlation=0.3)


def _np_hist(final, success, edges):
    return np.histogram(final[success.astype(bool)], bins=edges)[0].astype(np.int64)


def test_count_only_bins_equal_np_histogram_of_summary_output_1e6():
    """10^6 paths of the config.json scenario: MODE 0 + in-kernel bins vs np.histogram(bins=100, range=...) of the
    MODE 1 launch's per-path output — uniform, log-spaced and deliberately awkward edges."""
    import torch

    p = params_from_config(_cfg())
    n, wm = 1_000_000, 233
    full = E.DeviceBatch(p, wm, n, want="summary")
    full.launch(12345, 1, 0)
    final = full.summary["final_balance"].cpu().numpy()
    ok = full.success.cpu().numpy()
    assert 0.9 < ok.mean() < 1.0
    cohort = final[ok.astype(bool)]
    lo, hi = float(cohort.min()), float(cohort.max())
    cases = {
        "uniform_data_range": E.uniform_hist_edges((lo, hi), 100),                     # == np.histogram(x, bins=100)
        "uniform_fixed_range_60": E.uniform_hist_edges((0.0, 5.0e7), 60),               # the UI's 60 bins, values above hi dropped
        "log_spaced": np.geomspace(1.0e4, 1.0e9, 101),                                  # fixed log-spaced bins
        "one_bin": np.array([lo, hi]),
        "max_bins": E.uniform_hist_edges((lo, hi), N.MCR_MAX_HIST_BINS),
        "zero_width_bins": np.array([0.0, 1e6, 1e6, 1e6, 5e6, 5e6, 2e7, hi]),          # np.histogram allows equal edges
        "all_outside": np.array([-3.0, -2.0, -1.0]),
        "right_edge_closed": np.array([lo, np.median(cohort), hi]),                     # max lands in the last bin
    }
    for name, edges in cases.items():
        b = E.DeviceBatch(p, wm, n, want="count", hist_edges=edges)
        b.launch(12345, 1, 0)
        got = b.hist_bins.cpu().numpy()
        exp = _np_hist(final, ok, edges)
        assert np.array_equal(got, exp), f"{name}: {np.abs(got - exp).sum()} paths in other bins"
        assert b.counters.cpu().tolist() == full.counters.cpu().tolist(), name
        if name == "uniform_data_range":
            assert got.sum() == ok.sum()
            assert np.array_equal(got, np.histogram(cohort, bins=100)[0])
            assert np.allclose(edges, np.histogram(cohort, bins=100)[1], rtol=0, atol=0)
        if name == "all_outside":
            assert got.sum() == 0
    # bins accumulate across launches like the counters, and live in the tail of the single reduce vector
    b = E.DeviceBatch(p, wm, n // 2, want="count", hist_edges=cases["log_spaced"])
    b.launch(12345, 1, 0)
    b.launch(12345, 1, n // 2)
    assert np.array_equal(b.hist_bins.cpu().numpy(), _np_hist(final, ok, cases["log_spaced"]))
    assert b.reduce_vec.numel() == 2 + b.sizes.retirement_years + b.sizes.ruin_bins + 100
    assert b.hist_bins.data_ptr() == b.reduce_vec[-100:].data_ptr()
    del full, b
    torch.cuda.empty_cache()


@pytest.mark.parametrize("scenario", ["s60", "jorge", "annual_tax"])
def test_bins_vs_oracle_1e5(oracle, scenario):
    """10^5 paths vs np.histogram of the ORACLE's final balances (reference arithmetic).  The kernel's balances agree
    with the oracle's to 1e-9 relative, so a path may change bins only if it lies that close to an edge."""
    if scenario == "s60":
        cfg, wm = _s60(), 120
    elif scenario == "jorge":
        cfg, wm = _cfg("jorge.json", equity_inflation_correlation=0.3), 75
    else:
        cfg, wm = Config(**dict(load_golden("helpers.json")["tax_cfgs"][5], retirement_years=30, seed=7)), 61
    p = params_from_config(cfg)
    n = 100_000
    c = oracle.run_batch(p, 12345, 1, 1 << 33, n, wm, want_trajectories=False)
    ok = c["success"].astype(bool)
    cohort = c["final_balance"][ok]
    assert cohort.size > 1000
    for edges in (E.uniform_hist_edges((0.0, float(np.quantile(cohort, 0.999))), 100), np.geomspace(1e3, 1e10, 141)):
        g = E.run_batch_host(p, 12345, 1, 1 << 33, n, wm, want_summary=False, want_trajectories=False, hist_edges=edges)
        assert g["counters"].tolist() == c["counters"].tolist()
        exp = np.histogram(cohort, bins=edges)[0].astype(np.int64)
        got = g["hist_bins"].astype(np.int64)
        if not np.array_equal(got, exp):
            # only paths within 1e-8 (relative) of an edge may have moved
            near = sum(int(np.any(np.abs(edges - x) <= 1e-8 * np.maximum(1.0, np.abs(edges)))) for x in cohort)
            assert np.abs(got - exp).sum() <= 2 * near, f"{scenario}: {np.abs(got - exp).sum()} moved, {near} near an edge"
        assert got.sum() == np.count_nonzero((cohort >= edges[0]) & (cohort <= edges[-1]))


def test_every_output_mode_and_entry_point_bins_the_same(oracle):
    """The binning code is shared by the count-only / summary / full-output variants and by the device-pointer, host-buffer
    and multi-device entry points: same bins everywhere, next to unchanged per-path outputs."""
    p = params_from_config(_cfg("jorge.json", equity_inflation_correlation=0.3))
    n, wm = 5_003, 75                  # ragged: the last workgroup is partial
    edges = np.geomspace(1e4, 1e9, 33)
    ref = E.run_batch_host(p, 99, 1, 0, n, wm)
    exp = _np_hist(ref["final_balance"], ref["success"], edges)
    assert exp.sum() > 0
    for kw in (dict(want_summary=False, want_trajectories=False), dict(want_trajectories=False), dict()):
        g = E.run_batch_host(p, 99, 1, 0, n, wm, hist_edges=edges, **kw)
        assert np.array_equal(g["hist_bins"].astype(np.int64), exp), kw
        for k in g:
            if k != "hist_bins":
                assert np.array_equal(g[k], ref[k], equal_nan=True), k
    g = E.run_batch_host(p, 99, 1, 0, n, wm, want_summary=False, want_trajectories=False, hist_edges=edges, devices=[0, 0, 0])
    assert np.array_equal(g["hist_bins"].astype(np.int64), exp)
    g = E.run_batch_host(p, N.numpy_rng(2024), 1, 0, 1500, wm, want_trajectories=False, hist_edges=edges)   # NumPy-stream variant
    assert np.array_equal(g["hist_bins"].astype(np.int64), _np_hist(g["final_balance"], g["success"], edges))
    for want in ("count", "summary", "full"):
        b = E.DeviceBatch(p, wm, n, want=want, hist_edges=edges)
        b.launch(99, 1, 0)
        assert np.array_equal(b.hist_bins.cpu().numpy(), exp), want


def test_histogram_argument_errors():
    p = params_from_config(_cfg())
    with pytest.raises(ValueError):
        E.hist_edge_array([3.0, 2.0, 4.0])
    with pytest.raises(ValueError):
        E.hist_edge_array([1.0])
    with pytest.raises(ValueError):
        E.hist_edge_array(np.arange(N.MCR_MAX_HIST_BINS + 3, dtype=np.float64))
    with pytest.raises(ValueError):
        E.hist_edge_array([0.0, np.inf])
    # straight through the ABI: the host entry checks what a host pointer lets it check
    lib = N.load_library()
    o = N.McrOutputs()
    ctr = np.zeros(2, dtype=np.uint64)
    bins = np.zeros(4, dtype=np.uint64)
    o.counters = ctr.ctypes.data
    o.hist_bins = bins.ctypes.data
    rng = N.philox_rng(1)
    for edges, nb in ((np.array([0.0, 2.0, 1.0, 3.0, 4.0]), 4), (np.array([0.0, 1.0, 2.0, 3.0, np.nan]), 4), (None, 4),
                      (np.zeros(5), -1), (np.zeros(5), N.MCR_MAX_HIST_BINS + 1)):
        o.hist_edges = edges.ctypes.data if edges is not None else None
        o.hist_n_bins = nb
        rc = lib.mcr_run_batch_host_rng(C.byref(p), C.byref(rng), 1, 0, 64, 12, None, C.byref(o), 0)
        assert rc == -1 and "hist" in N.last_error(), (nb, N.last_error())
    assert ctr.tolist() == [0, 0]
    # hist_n_bins == 0 / hist_bins == NULL: simply not requested
    o.hist_n_bins = 0
    assert lib.mcr_run_batch_host_rng(C.byref(p), C.byref(rng), 1, 0, 64, 12, None, C.byref(o), 0) == 0
    assert ctr[1] == 64 and bins.sum() == 0


def test_run_sharded_histogram_fixed_edges_is_count_only_and_equals_data_ranged_bins():
    """distributed.run_sharded_histogram: with a fixed range the route is the count-only kernel (no per-path buffer);
    its bins equal the data-ranged route's when given that route's own range."""
    from monte_carlo_retirement_amd import distributed as D

    p = params_from_config(_s60())
    n = 300_000
    ranged = D.run_sharded_histogram(p, 12345, 1, n, 120, n_bins=100)
    lo, hi = float(ranged["hist_edges"][0]), float(ranged["hist_edges"][-1])
    fixed = D.run_sharded_histogram(p, 12345, 1, n, 120, n_bins=100, value_range=(lo, hi))
    assert np.array_equal(fixed["hist_bins"], ranged["hist_bins"])
    assert np.array_equal(fixed["hist_edges"], ranged["hist_edges"])
    assert fixed["counts"].success == ranged["counts"].success == int(fixed["hist_bins"].sum())
    assert fixed["counts"].wr_obs_counts.tolist() == ranged["counts"].wr_obs_counts.tolist()
    assert fixed["exchange"] == "none (1 GPU)"
    logb = D.run_sharded_histogram(p, 12345, 1, n, 120, hist_edges=np.geomspace(1e3, 1e10, 71))
    assert logb["hist_bins"].shape == (70,) and 0 < logb["hist_bins"].sum() <= fixed["counts"].success
