"""HIP path engine vs the CPU oracle and the reference's golden vectors — on a real MI355X.

Everything goes through the C ABI (libmcr_hip.so).  Tolerances (stated per test):
* helper device functions use only + - * / min max: IEEE-exact, compared BIT-FOR-BIT;
* anything through exp / log / sincospi (device libm vs glibc differ by <= a few ulp per call,
  compounding over <= ~1500 months): relative 1e-9 (plus 1e-6 absolute = the reference's own
  dollar epsilon); Success flags must be identical.
"""

from __future__ import annotations

import os

import numpy as np
import pytest

from conftest import GOLDEN, STREAM_ID, assert_same_float, compare_batch_to_golden, load_golden
from monte_carlo_retirement_amd import Config, params_from_config
from monte_carlo_retirement_amd import _native as N
from monte_carlo_retirement_amd import engine as E

pytestmark = pytest.mark.gpu

REL = 1e-9   # fp64 tolerance of the path (see module docstring)
ABS = 1e-6   # SMALL_EPSILON dollars


def _params(cfgd):
    return params_from_config(Config(**cfgd))


def test_device_is_gfx950():
    assert N.device_count() >= 1


def test_helpers_bit_exact_vs_reference():
    g = load_golden("helpers.json")
    got = E.eval_helper_host(N.MCR_HELPER_WITHDRAW, None, [r["in"] for r in g["withdraw"]])
    for row, r in zip(got, g["withdraw"]):
        for x, y in zip(row, r["out"]):
            assert_same_float(float(x), y, f"withdraw{r['in']}")
    got = E.eval_helper_host(N.MCR_HELPER_NLV, None, [r["in"] for r in g["nlv"]])
    for row, r in zip(got, g["nlv"]):
        assert_same_float(float(row[0]), r["out"], f"nlv{r['in']}")
    for ci, cfgd in enumerate(g["tax_cfgs"]):
        p = _params(cfgd)
        rows = [r for r in g["rebalance"] if r["cfg"] == ci]
        got = E.eval_helper_host(N.MCR_HELPER_REBALANCE, p, [r["in"] for r in rows])
        for row, r in zip(got, rows):
            for x, y in zip(row, r["out"]):
                assert_same_float(float(x), y, f"rebalance cfg{ci} {r['in']}")
        rows = [r for r in g["annual_tax"] if r["cfg"] == ci]
        got = E.eval_helper_host(N.MCR_HELPER_ANNUAL_TAX, p, [r["in"] for r in rows])
        for row, r in zip(got, rows):
            for x, y in zip(row[:4], r["out"][:4]):
                assert_same_float(float(x), y, f"annual_tax cfg{ci} {r['in']}")
            assert bool(row[4]) == r["out"][4]
    got = E.eval_helper_host(N.MCR_HELPER_MONTHLY_GROSS, None, [[r["mu_log"], r["sigma_log"], r["z"]] for r in g["monthly_gross"]])
    exp = np.array([r["gross"] for r in g["monthly_gross"]])
    np.testing.assert_allclose(got[:, 0], exp, rtol=6e-16)  # device exp (fused argument) vs glibc exp: <= 2 ulp + the argument's 1/2 ulp


def test_reference_helper_unit_pins():
    """The exact cases of the reference's tests :605-631 and :634-662, on the device."""
    out = E.eval_helper_host(N.MCR_HELPER_WITHDRAW, None, [[100.0, 0.0, 90.0, 1.0, 0.20], [80.0, 100.0, 40.0, 1.0, 0.20]])
    assert out[0].tolist() == pytest.approx([0.0, 0.0, 100.0, 80.0])
    assert out[1].tolist() == pytest.approx([40.0, 50.0, 40.0, 40.0])
    cfgd = load_golden("helpers.json")["tax_cfgs"][0]
    b1, c1, b2, c2 = E.eval_helper_host(N.MCR_HELPER_REBALANCE, _params(cfgd), [[70.0, 50.0, 30.0, 30.0]])[0]
    total = b1 + b2
    assert b1 / total == pytest.approx(0.60, abs=1e-10) and b2 / total == pytest.approx(0.40, abs=1e-10)
    assert total < 100.0
    gross_sale = 70.0 - b1
    basis_removed = 50.0 * (gross_sale / 70.0)
    tax_paid = (gross_sale - basis_removed) * 0.10
    assert c1 == pytest.approx(50.0 - basis_removed) and c2 == pytest.approx(30.0 + gross_sale - tax_paid)


def test_shock_rows_match_oracle(oracle):
    """Philox integers are exact; Box-Muller goes through log/sqrt/sincospi: 1.5e-14 relative (the radius: one-step
    sqrt, tests/test_gpu_math.py) + 3e-15 absolute (angle and logarithm)."""
    for seed, stream, pb, rho in [(12345, 1, 0, 0.3), (2**40 + 17, 0, 2**33 + 5, -1.0), (7, 1, 2**32 - 2, 1.0)]:
        got = E.draw_shocks_host(seed, stream, pb, 5, 700, rho)
        for i in range(5):
            exp = oracle.draw_shocks(seed, stream, pb + i, 700, rho)
            np.testing.assert_allclose(got[i], exp, rtol=1.5e-14, atol=3e-15)
    a = E.draw_shocks_host(5, 1, 4, 1, 100, 1.0)[0]
    assert np.array_equal(a[:, 1], a[:, 0])  # rho = +1 preserved exactly (reference test :185-195)
    b = E.draw_shocks_host(5, 1, 4, 1, 100, -1.0)[0]
    assert np.array_equal(b[:, 1], -b[:, 0])


def test_deterministic_paths_vs_reference():
    """sigma = 0 scenarios of the reference's own tests; only exp() separates GPU from CPython."""
    for case in load_golden("paths_deterministic.json"):
        res = E.run_batch_host(_params(case["cfg"]), 0, 1, 0, 1, case["working_months"])
        try:
            compare_batch_to_golden(res, [case["result"]], exact=False, rel=REL, abs_tol=ABS)
        except AssertionError as e:
            raise AssertionError(f"{case['name']}: {e}") from e


@pytest.mark.parametrize("fname", ["paths_injected.json", "paths_fuzz.json"])
def test_stochastic_paths_vs_reference(fname):
    """Kernel with its own in-register RNG vs the reference run on the same (injected) Philox shocks."""
    for g in load_golden(fname):
        res = E.run_batch_host(_params(g["cfg"]), g["seed"], STREAM_ID[g["stream"]], g["path_begin"], g["n_paths"], g["working_months"])
        try:
            compare_batch_to_golden(res, g["results"], exact=False, rel=REL, abs_tol=ABS)
        except AssertionError as e:
            raise AssertionError(f"{g['name']}: {e}") from e
        assert int(res["counters"][0]) == sum(r["Success"] for r in g["results"])
        assert int(res["counters"][1]) == g["n_paths"]
        wr = np.array([r["WithdrawalRateTrajectory"] for r in g["results"]])
        assert res["wr_obs_counts"].tolist() == (~np.isnan(wr)).sum(axis=0).tolist()
        assert int(res["ruin_year_bins"].sum()) == sum(not r["Success"] for r in g["results"])


def test_injected_numpy_native_shocks_vs_reference():
    """The reference's OWN NumPy-RNG shocks, replayed through the kernel's injection hook."""
    meta = load_golden("numpy_native_paths.json")
    arrays = np.load(os.path.join(GOLDEN, "numpy_native_shocks.npz"))
    for g in meta:
        sh = arrays[g["name"]]
        res = E.run_batch_host(_params(g["cfg"]), 0, 1, 0, sh.shape[0], g["working_months"], injected_shocks=sh)
        compare_batch_to_golden(res, g["results"], exact=False, rel=REL, abs_tol=ABS)


def test_metric_10k_success_probability_error():
    """BASELINE metric: |p_gpu - p_ref| on the 10k-path config (config.json, wm=233). Bar: <= 1e-4."""
    meta = load_golden("metric_10k_config_json.json")
    z = np.load(os.path.join(GOLDEN, "metric_10k_config_json.npz"))
    n = meta["n_paths"]
    res = E.run_batch_host(_params(meta["cfg"]), meta["seed"], STREAM_ID[meta["stream"]], 0, n, meta["working_months"], want_trajectories=False)
    flags = np.unpackbits(z["success_bits"])[:n]
    flips = int((res["success"] != flags).sum())
    p_gpu = float(res["counters"][0]) / n
    assert abs(p_gpu - meta["success_count"] / n) <= 1e-4, (p_gpu, meta["success_count"] / n)
    assert flips == 0
    np.testing.assert_allclose(res["start_balance"], z["Start_Balance"], rtol=REL)
    np.testing.assert_allclose(res["final_balance"], z["Final_Balance"], rtol=REL, atol=ABS)
    np.testing.assert_allclose(res["inflation_at_retirement"], z["Inflation_At_Retirement"], rtol=REL)
    np.testing.assert_allclose(res["first_year_real_gross_withdrawal"], z["First_Year_Real_Gross_Withdrawal"], rtol=REL, atol=ABS)
    assert np.array_equal(np.isnan(res["years_to_ruin"]), np.isnan(z["YearsToRuin"]))
    np.testing.assert_allclose(res["years_to_ruin"], z["YearsToRuin"], rtol=0, atol=0, equal_nan=True)


@pytest.mark.parametrize("name", ["C1_config_json_wm233", "S60_wm120", "FAILING_wm24", "ANNUAL_wm50"])
def test_kernel_vs_oracle_100k(oracle, name):
    """Identical counters, 1e5 paths: Success flags identical, summary fields within REL."""
    g = [x for x in load_golden("paths_injected.json") if x["name"] == name][0]
    p = _params(g["cfg"])
    n = int(os.environ.get("MCR_ORACLE_PATHS", "100000"))          # soak by hand: more paths, another range
    first = int(os.environ.get("MCR_ORACLE_FIRST_PATH", "0"))
    sid = STREAM_ID[g["stream"]]
    gpu = E.run_batch_host(p, g["seed"], sid, first, n, g["working_months"], want_trajectories=False)
    threads = 16 if n > 200_000 else 1
    per = (n + threads - 1) // threads
    parts = [None] * threads

    def work(t):
        b = first + t * per
        parts[t] = oracle.run_batch(p, g["seed"], sid, b, max(0, min(per, first + n - b)), g["working_months"], want_trajectories=False)

    import threading
    ths = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    cpu = {k: (sum(q[k] for q in parts) if k in ("counters", "ruin_year_bins", "wr_obs_counts") else np.concatenate([q[k] for q in parts]))
           for k in parts[0]}
    flips = int((gpu["success"] != cpu["success"]).sum())
    assert flips == 0, f"{flips} flipped Success flags out of {n}"
    assert gpu["counters"].tolist() == cpu["counters"].tolist()
    assert gpu["ruin_year_bins"].tolist() == cpu["ruin_year_bins"].tolist()
    assert gpu["wr_obs_counts"].tolist() == cpu["wr_obs_counts"].tolist()
    # 1e-9 relative to the PATH'S money scale (its balance at retirement): a final balance that is a small remainder of
    # multi-million flows carries the absolute error of those flows (1 path in 2e6 of a soak: 1e-5 dollars on 2 641)
    scale = np.maximum(np.abs(cpu["start_balance"]), 1.0)
    for k in E.SUMMARY_FIELDS:
        both_nan = np.isnan(gpu[k]) & np.isnan(cpu[k])
        err = np.where(both_nan, 0.0, np.abs(gpu[k] - cpu[k]))
        assert np.all(err <= ABS + REL * np.maximum(np.abs(np.nan_to_num(cpu[k])), scale)), (k, float(np.nanmax(err)))


def test_sharding_and_ragged_sizes_are_bit_identical():
    """Results depend only on the GLOBAL path index: any split of the range gives the same bits;
    sizes that are not multiples of the wavefront / workgroup are handled (tail lanes masked)."""
    g = [x for x in load_golden("paths_injected.json") if x["name"] == "C3_jorge_wm75_rho03"][0]
    p = _params(g["cfg"])
    n = 1000
    whole = E.run_batch_host(p, 42, 1, 0, n, 75)
    parts = [E.run_batch_host(p, 42, 1, b, m, 75) for b, m in [(0, 1), (1, 63), (64, 193), (257, 743)]]
    for k in list(E.SUMMARY_FIELDS) + ["success"]:
        assert np.array_equal(np.concatenate([q[k] for q in parts]), whole[k], equal_nan=True), k
    for k in ("trajectory", "real_trajectory", "withdrawal_rate_trajectory"):
        assert np.array_equal(np.concatenate([q[k] for q in parts], axis=1), whole[k], equal_nan=True), k
    assert sum(int(q["counters"][0]) for q in parts) == int(whole["counters"][0])
    assert sum(int(q["counters"][1]) for q in parts) == n
    assert sum(q["ruin_year_bins"] for q in parts).tolist() == whole["ruin_year_bins"].tolist()
    # empty batch: accepted, nothing written
    empty = E.run_batch_host(p, 42, 1, 0, 0, 75)
    assert empty["counters"].tolist() == [0, 0]


def test_common_random_numbers_across_working_months():
    """Shock row k drives absolute month k+1 for every candidate (simulation.py:519-520):
    the accumulation samples of a path are identical for two working-month candidates."""
    g = load_golden("paths_injected.json")[0]
    p = _params(g["cfg"])
    a = E.run_batch_host(p, 9, 0, 0, 512, 120)
    b = E.run_batch_host(p, 9, 0, 0, 512, 233)
    assert np.array_equal(a["trajectory"][:11], b["trajectory"][:11])
    # and the success probability is non-decreasing in working months (reference test :55-81)
    counts = [int(E.run_batch_host(p, 9, 0, 0, 4096, wm, want_summary=False, want_trajectories=False)["counters"][0])
              for wm in range(180, 260, 12)]
    assert counts == sorted(counts), counts


def test_full_size_invariants_1e6():
    """BASELINE config 2 size (1e6 paths, config.json, count-only): size-independent properties —
    the count-only kernel agrees with the per-path flags of the summary kernel, the histogram of
    ruin years accounts for every failed path, WR observation counts are non-increasing."""
    g = load_golden("paths_injected.json")[0]
    p = _params(g["cfg"])
    n = 1_000_000
    c = E.run_batch_host(p, 12345, 1, 0, n, 233, want_summary=False, want_trajectories=False)
    s = E.run_batch_host(p, 12345, 1, 0, n, 233, want_trajectories=False)
    assert int(c["counters"][1]) == n
    assert int(c["counters"][0]) == int(s["success"].sum()) == int(s["counters"][0])
    assert int(c["ruin_year_bins"].sum()) == n - int(c["counters"][0])
    wr = c["wr_obs_counts"].astype(np.int64)
    assert wr[0] <= n and np.all(np.diff(wr) <= 0) and wr[-1] >= int(c["counters"][0])
    assert np.all(np.isnan(s["years_to_ruin"]) == (s["success"] == 1))
    # sanity vs the reference's native-RNG estimate (independent streams: 5-sigma binomial bound)
    ref = [x for x in load_golden("numpy_native_stats.json") if x["name"] == "C1_config_json_wm233"][0]
    p_ref, n_ref = ref["success_probability_pct"] / 100.0, ref["n_paths"]
    p_gpu = int(c["counters"][0]) / n
    sigma = (p_gpu * (1 - p_gpu) * (1 / n + 1 / n_ref)) ** 0.5
    assert abs(p_gpu - p_ref) < 5 * sigma, (p_gpu, p_ref, sigma)


def test_success_probability_within_1e4_of_cpu_at_1e6_paths(oracle):
    """North-star accuracy clause: success probability within +-1e-4 of the CPU reference at 1e6
    paths (config.json scenario, identical counters).  The CPU side is the oracle on host threads."""
    import threading

    g = load_golden("paths_injected.json")[0]
    p = _params(g["cfg"])
    n, wm, n_threads = 1_000_000, 233, 16
    gpu = E.run_batch_host(p, 12345, 1, 0, n, wm, want_trajectories=False)
    per = n // n_threads
    parts = [None] * n_threads

    def work(t):
        begin = t * per
        cnt = per if t < n_threads - 1 else n - begin
        parts[t] = oracle.run_batch(p, 12345, 1, begin, cnt, wm, want_trajectories=False)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    cpu_success = np.concatenate([q["success"] for q in parts])
    cpu_count = int(sum(int(q["counters"][0]) for q in parts))
    flips = int((gpu["success"] != cpu_success).sum())
    p_gpu, p_cpu = int(gpu["counters"][0]) / n, cpu_count / n
    assert abs(p_gpu - p_cpu) <= 1e-4, (p_gpu, p_cpu, flips)
    assert flips <= 2, flips  # in practice 0: the eps-comparisons sit far from fp64 round-off
    # per-path terminal wealth: 1e-9 relative to the PATH'S scale (its balance at retirement): a final
    # balance that is a small remainder of multi-million flows carries the absolute error of those flows
    cpu_final = np.concatenate([q["final_balance"] for q in parts])
    scale = np.maximum(np.abs(cpu_final), np.concatenate([q["start_balance"] for q in parts]))
    err = np.abs(gpu["final_balance"] - cpu_final)
    assert np.all(err <= ABS + REL * scale), float((err / (ABS + REL * scale)).max())
    assert float(np.median(err / np.maximum(np.abs(cpu_final), 1.0))) < 1e-12
