"""Edge cases of the path on the GPU vs the oracle: maximum stream count (all LDS lock slots in use),
very long horizons, degenerate scenarios, ragged batch sizes, argument errors."""

from __future__ import annotations

import ctypes as C
import os

import numpy as np
import pytest

from conftest import load_golden
from monte_carlo_retirement_amd import Config, params_from_config
from monte_carlo_retirement_amd import _native as N
from monte_carlo_retirement_amd import engine as E

pytestmark = pytest.mark.gpu
REL, ABS = 1e-9, 1e-6


def _base(**over):
    d = dict(load_golden("helpers.json")["tax_cfgs"][5])
    d.update(over)
    return d


def _check(oracle, cfgd, wm, n=300, seed=11, stream=1, begin=0):
    p = params_from_config(Config(**cfgd))
    g = E.run_batch_host(p, seed, stream, begin, n, wm)
    c = oracle.run_batch(p, seed, stream, begin, n, wm)
    assert np.array_equal(g["success"], c["success"])
    assert g["counters"].tolist() == c["counters"].tolist()
    assert g["ruin_year_bins"].tolist() == c["ruin_year_bins"].tolist()
    assert g["wr_obs_counts"].tolist() == c["wr_obs_counts"].tolist()
    for k in E.SUMMARY_FIELDS + ("trajectory", "real_trajectory", "withdrawal_rate_trajectory"):
        np.testing.assert_allclose(g[k], c[k], rtol=REL, atol=ABS, equal_nan=True, err_msg=k)
    return g


def _stream(i, indexed, dur, age, amount=300.0, tax=0.1):
    return {"name": f"s{i}", "monthly_amount_today": amount, "start_at_age": age, "duration_years": dur,
            "inflation_indexed": indexed, "tax_rate": tax}


def test_sixteen_streams_all_kinds(oracle):
    """MCR_INLINE_STREAMS streams (the by-value block full): 10 non-indexed (10 LDS lock columns), finite / zero / infinite
    durations, start ages before, at and long after retirement, overlapping windows.  (Longer lists: test_gpu_many_streams.py.)"""
    streams = [_stream(i, indexed=(i % 3 == 0), dur=[None, 0, 1, 3, 7][i % 5], age=38.0 + 1.75 * i, amount=150.0 + 40 * i, tax=0.05 * (i % 4))
               for i in range(16)]
    cfgd = _base(initial_balance=900_000.0, monthly_contribution=1_000.0, monthly_expenses=5_500.0, retirement_years=25,
                 inv1_use_realized_gains_tax_system=True, inv1_realized_gains_tax_rate=0.15,
                 inv2_use_realized_gains_tax_system=True, inv2_realized_gains_tax_rate=0.15, other_income_streams=streams)
    g = _check(oracle, cfgd, wm=30)
    assert 0 < int(g["counters"][0]) < 300  # mixed outcomes: the streams matter
    # a 17th stream is no error (the reference takes any list, config.py:99): it goes behind mcr_params.extra_streams
    p17 = params_from_config(Config(**dict(cfgd, other_income_streams=streams + [_stream(16, True, None, 70.0)])))
    assert p17.n_streams == 17 and bool(p17.extra_streams)


def test_very_long_horizon(oracle):
    """100 retirement years after 70 working years: 2040 months per path, T = 171 yearly samples."""
    cfgd = _base(initial_balance=50_000.0, monthly_contribution=800.0, contribution_growth_rate_annual=0.02,
                 monthly_expenses=2_500.0, retirement_years=100, current_age=20.0,
                 inv1_annual_tax_on_gains_rate=0.1, inv2_annual_tax_on_gains_rate=0.1)
    g = _check(oracle, cfgd, wm=840, n=128)
    assert g["trajectory"].shape == (171, 128) and g["withdrawal_rate_trajectory"].shape == (100, 128)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 255, 256, 257, 1000])
def test_ragged_batch_sizes(oracle, n):
    cfgd = load_golden("paths_injected.json")[8]["cfg"]  # FAILING scenario: mixed outcomes
    _check(oracle, cfgd, wm=24, n=n, seed=5, begin=2**33 + 11)


def test_degenerate_scenarios(oracle):
    # nothing to simulate before retirement, nothing invested, nothing spent
    _check(oracle, _base(initial_balance=0.0, monthly_expenses=0.0, retirement_years=1), wm=0, n=70)
    # all in one asset, 100 % taxes everywhere
    _check(oracle, _base(allocation_inv1_pct=1.0, inv1_use_realized_gains_tax_system=True, inv1_realized_gains_tax_rate=1.0,
                         monthly_expenses=3_000.0, retirement_years=12), wm=13, n=200)
    _check(oracle, _base(allocation_inv1_pct=0.0, inv2_use_realized_gains_tax_system=False, inv2_annual_tax_on_gains_rate=1.0,
                         inflation_rate_volatility=0.08, monthly_expenses=3_000.0, retirement_years=12), wm=11, n=200)
    # extreme volatility: growth factors over many orders of magnitude
    _check(oracle, _base(inv1_returns_volatility=1.5, inv2_premium_over_inflation_volatility=0.8, inflation_rate_volatility=0.3,
                         monthly_expenses=4_000.0, retirement_years=20), wm=36, n=400)


def test_argument_errors():
    lib = N.load_library()
    p = params_from_config(Config(**_base()))
    o = N.McrOutputs()
    assert lib.mcr_run_batch_host(C.byref(p), 1, 1, 0, 8, -5, None, C.byref(o), 0) == -1 and "working_months" in N.last_error()
    assert lib.mcr_run_batch_host(C.byref(p), 1, 1, 0, 8, 12, None, C.byref(o), 99) == -1 and "device" in N.last_error()
    assert lib.mcr_run_batch_host(C.byref(p), 1, 1, 2**64 - 4, 8, 12, None, C.byref(o), 0) == -1  # path range overflows
    bad = params_from_config(Config(**_base()))
    bad.n_streams = 17
    assert lib.mcr_run_batch_host(C.byref(bad), 1, 1, 0, 8, 12, None, C.byref(o), 0) == -1
    r = N.numpy_rng(5)
    r.kind = 7
    assert lib.mcr_run_batch_host_rng(C.byref(p), C.byref(r), 1, 0, 8, 12, None, C.byref(o), 0) == -1
    with pytest.raises(ValueError):
        N.numpy_rng(2**300)
    traj = np.empty((3, 8))
    o.trajectory = traj.ctypes.data
    o.path_stride = 4  # < n_paths
    assert lib.mcr_run_batch_host(C.byref(p), 1, 1, 0, 8, 12, None, C.byref(o), 0) == -1 and "path_stride" in N.last_error()


def test_launch_on_a_side_stream_and_reuse_of_buffers():
    """mcr_run_batch is asynchronous on the hipStream_t it is given: launches on a non-default torch
    stream, back-to-back into the same buffers, give the same bits as the default stream."""
    import torch

    cfgd = load_golden("paths_injected.json")[3]["cfg"]
    p = params_from_config(Config(**cfgd))
    n, wm = 20_000, 75
    ref = E.DeviceBatch(p, wm, n, want="full")
    ref.launch(7, 1, 0)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    b = E.DeviceBatch(p, wm, n, want="full")
    with torch.cuda.stream(side):
        b.launch(99, 1, 0)          # first a different seed ...
        b.zero_counters()
        b.launch(7, 1, 0)           # ... then overwrite on the same stream, no host sync in between
    side.synchronize()
    # (columns >= n are stride padding, never written)
    assert torch.equal(b.trajectory[:, :n], ref.trajectory[:, :n])
    assert torch.allclose(b.withdrawal_rate_trajectory[:, :n], ref.withdrawal_rate_trajectory[:, :n], rtol=0, atol=0, equal_nan=True)
    assert torch.equal(b.success, ref.success) and torch.equal(b.counters, ref.counters)
    assert torch.equal(b.summary["final_balance"], ref.summary["final_balance"])


def test_concurrent_calls_from_host_threads(oracle):
    """The entry points are re-entrant (thread-local error state, per-thread stream and scratch): four host
    threads simulate different seeds at once, as the reference's server does on executor threads — and their
    kernels OVERLAP on the GPU (private non-blocking streams, no device-wide synchronisation)."""
    import threading
    import time

    cfgd = load_golden("paths_injected.json")[8]["cfg"]
    p = params_from_config(Config(**cfgd))
    n, wm = 3000, 24
    expected = [E.run_batch_host(p, 100 + t, t % 2, 0, n, wm) for t in range(4)]
    got = [None] * 4
    errs = []

    def work(t):
        try:
            for _ in range(3):
                got[t] = E.run_batch_host(p, 100 + t, t % 2, 0, n, wm)
        except Exception as e:  # pragma: no cover
            errs.append(e)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errs, errs
    for t in range(4):
        for k in ("success", "final_balance", "trajectory", "counters", "ruin_year_bins"):
            assert np.array_equal(got[t][k], expected[t][k], equal_nan=True), (t, k)

    # overlap: a 16 384-path count-only batch of 833-month paths is one wave on a quarter of the SIMDs (~1 ms,
    # latency-bound); four of them from four threads must take well under four times one
    p2 = params_from_config(Config(**load_golden("paths_injected.json")[0]["cfg"]))
    reps, n2 = 12, 16_384

    def probe(t):
        for i in range(reps):
            r = E.run_batch_host(p2, 7 + t, 1, 0, n2, 233, want_summary=False, want_trajectories=False, want_bins=False)
            assert int(r["counters"][1]) == n2

    probe(0)   # warm-up: stream + scratch of this thread, code object
    t0 = time.perf_counter()
    for t in range(4):
        probe(t)
    serial = time.perf_counter() - t0
    best = float("inf")
    for _ in range(3):
        threads = [threading.Thread(target=probe, args=(t,)) for t in range(4)]
        t0 = time.perf_counter()
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        best = min(best, time.perf_counter() - t0)
    assert best < 0.7 * serial, (best, serial)


def test_multi_device_host_entry_equals_single_device(oracle):
    """mcr_run_batch_multi_host_rng (one host thread per listed device, shards of the global path range, counters
    summed on the host) returns exactly what one device returns for the whole range — Philox counters carry the
    global path index.  The box has one GPU: listing it several times exercises the sharding, the column offsets
    into the caller's arrays and the host-side sums; MCR_DEVICE_ALL takes every visible device."""
    cfgd = load_golden("paths_injected.json")[4]["cfg"]
    p = params_from_config(Config(**cfgd))
    n, wm = 1001, 30
    one = E.run_batch_host(p, 4242, 1, 17, n, wm)
    for devs in ([0], [0, 0], [0, 0, 0], [0] * 7):
        got = E.run_batch_host(p, 4242, 1, 17, n, wm, devices=devs)
        for k in one:
            assert np.array_equal(got[k], one[k], equal_nan=True), (devs, k)
    got = E.run_batch_host(p, 4242, 1, 17, n, wm, device=N.MCR_DEVICE_ALL)
    for k in one:
        assert np.array_equal(got[k], one[k], equal_nan=True), ("all", k)
    # injected shocks and explicit NumPy path seeds are offset per shard
    sz = E.query_sizes(p, wm)
    inj = np.random.default_rng(5).standard_normal((64, sz.shock_rows, 3))
    a = E.run_batch_host(p, 1, 1, 0, 64, wm, injected_shocks=inj)
    b = E.run_batch_host(p, 1, 1, 0, 64, wm, injected_shocks=inj, devices=[0, 0, 0])
    seeds = np.arange(1000, 1064, dtype=np.uint32)
    c = E.run_batch_host(p, N.numpy_rng(12345), 1, 0, 64, wm, path_seeds=seeds)
    d = E.run_batch_host(p, N.numpy_rng(12345), 1, 0, 64, wm, path_seeds=seeds, devices=[0, 0, 0])
    for k in a:
        assert np.array_equal(a[k], b[k], equal_nan=True) and np.array_equal(c[k], d[k], equal_nan=True), k
    ref = oracle.run_batch(p, 4242, 1, 17, n, wm)
    assert one["counters"].tolist() == ref["counters"].tolist() and np.array_equal(one["success"], ref["success"])
    # a bad device in the list surfaces as an error of the call, with the device named
    with pytest.raises(RuntimeError, match="device 99"):
        E.run_batch_host(p, 4242, 1, 0, 64, wm, devices=[0, 99])


def test_entry_points_restore_the_callers_device():
    """An ABI call on device d leaves the calling thread's current device as it found it (advisor r1)."""
    import torch

    assert torch.cuda.current_device() == 0
    p = params_from_config(Config(**load_golden("paths_injected.json")[0]["cfg"]))
    E.run_batch_host(p, 1, 1, 0, 64, 12)
    assert torch.cuda.current_device() == 0
    with pytest.raises(RuntimeError, match="out of range"):
        E.run_batch_host(p, 1, 1, 0, 64, 12, device=torch.cuda.device_count())
    assert torch.cuda.current_device() == 0


def test_very_large_batches_use_64bit_addressing(oracle):
    """2e8 paths in one count-only launch, and a 3.3e7-path full-output batch whose [2T+ry, stride] slab
    has > 2^32 elements (36 GB): counters add up, and paths spot-checked across the whole range (incl. the
    last one) equal the oracle's."""
    import torch

    from monte_carlo_retirement_amd import aggregation as A

    cfgd = load_golden("paths_injected.json")[4]["cfg"]   # jorge.json, rho = 0.3
    p = params_from_config(Config(**cfgd))
    wm = 75
    big = E.DeviceBatch(p, wm, 200_000_000, want="count")
    big.launch(31337, 1, 0)
    c = big.counters.cpu().tolist()
    assert c[1] == 200_000_000 and 0.985 < c[0] / c[1] < 0.997
    assert int(big.ruin_year_bins.sum().item()) == c[1] - c[0]
    del big

    n = 33_000_000
    b = E.DeviceBatch(p, wm, n, want="full")
    assert b.slab.numel() > 2**32
    b.launch(31337, 1, 0)
    picks = [0, 1, 63, 64, 2**24 + 5, 29_999_999, n // 2, n - 2, n - 1]
    traj = torch.stack([b.trajectory[:, g] for g in picks], dim=1).cpu().numpy()
    wr = torch.stack([b.withdrawal_rate_trajectory[:, g] for g in picks], dim=1).cpu().numpy()
    fin = torch.stack([b.summary["final_balance"][g] for g in picks]).cpu().numpy()
    for j, g in enumerate(picks):
        o = oracle.run_batch(p, 31337, 1, g, 1, wm)
        np.testing.assert_allclose(traj[:, j], o["trajectory"][:, 0], rtol=REL, atol=ABS)
        np.testing.assert_allclose(wr[:, j], o["withdrawal_rate_trajectory"][:, 0], rtol=REL, equal_nan=True)
        np.testing.assert_allclose(fin[j], o["final_balance"][0], rtol=REL, atol=ABS)
    assert int(b.counters[1].item()) == n
    # bands over the > 2^32-element slab: medians must be monotone sane and the counts complete
    tq, rq, wq, wc = A.band_quantiles(b, n)
    assert tq.shape == (b.sizes.trajectory_len, 7) and np.all(np.diff(tq, axis=1) >= 0)
    assert wc[0] == n and np.all(np.diff(wc) <= 0)
    assert tq[0].tolist() == [cfgd["initial_balance"]] * 7
    # the 100-bin histogram of successful final balances accounts for every successful path
    bins, edges = A.success_histogram(b.summary["final_balance"], b.success, 100)
    assert int(bins.sum()) == int(b.counters[0].item()) and len(edges) == 101 and np.all(np.diff(edges) > 0)
    ok_big, fin_big = int(b.counters[0].item()), fin.copy()
    del b
    torch.cuda.empty_cache()
    # the drop-in class on the same > 2^32-element slab: per-path frame, bands, sampled columns — no torch
    # indexing kernel sees the slab (LABNOTES.md, rounds 1-3 section 11)
    from monte_carlo_retirement_amd.simulation import RetirementMonteCarloSimulator

    sim = RetirementMonteCarloSimulator(Config(**cfgd), main_seed_override=31337)
    sim.use_final_seeds()
    df, tdf, samples, wdf, rdf, rsamples, wcounts = sim.run_monte_carlo_simulations(wm, n)
    assert len(df) == n and int(df["Success"].sum()) == ok_big
    np.testing.assert_array_equal(df["Final Balance"].to_numpy()[picks], fin_big)
    assert np.array_equal(tdf.to_numpy(), tq, equal_nan=True) and wcounts == wc.tolist()
    sel = np.random.RandomState(31337).choice(n, size=5, replace=False)
    for j, g in enumerate(sel):
        o = oracle.run_batch(p, 31337, 1, int(g), 1, wm)
        np.testing.assert_allclose(samples[j], o["trajectory"][:, 0], rtol=REL, atol=ABS)
        np.testing.assert_allclose(rsamples[j], o["real_trajectory"][:, 0], rtol=REL, atol=ABS)


def test_plain_c_caller_equals_the_python_path(tmp_path):
    """examples/c_caller.c — C99, no Python, no torch, MCR_DEVICE_ALL — on the config.json scenario: the same success
    count as the Python binding for the same seed and path range."""
    import json
    import subprocess

    from test_abi_cpu import build_c_caller

    exe = build_c_caller(tmp_path)
    n, wm, seed = 200_000, 233, 987654321
    r = subprocess.run([exe, str(n), str(wm), str(seed)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    fields = dict(kv.split("=") for kv in r.stdout.split())
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenarios", "config.json")) as fh:
        cfg = Config(**json.load(fh))
    ref = E.run_batch_host(params_from_config(cfg), seed, 1, 0, n, wm, want_trajectories=False)
    assert int(fields["paths"]) == n == int(ref["counters"][1])
    assert int(fields["success"]) == int(ref["counters"][0])
    # its in-kernel histogram (10 log-spaced bins; the C program builds the edges with pow()) vs NumPy on the per-path output
    edges = np.array([1e5 * 10.0 ** (0.5 * k) for k in range(11)])
    exp = np.histogram(ref["final_balance"][ref["success"].astype(bool)], bins=edges)[0]
    got = np.array([int(x) for x in fields["hist"].split(",")])
    assert got.sum() > 0.9 * int(ref["counters"][0])
    assert np.abs(got - exp).sum() <= 2, (got, exp)      # pow() vs ** may differ in the last bit of an edge


def test_any_subset_of_output_pointers():
    """`mcr_outputs`: "Any pointer may be NULL = not requested; the kernel variant is chosen from what is requested".
    Random subsets of the 13 output pointers (incl. none at all, one trajectory without the summary, bins without
    counters), strides larger than n, ragged n, both streams, through the raw C ABI: every requested array equals the
    full run's, bit for bit; padding beyond n and arrays that were not requested stay untouched.
    MCR_OUT_FUZZ_SEED / MCR_OUT_FUZZ_ROUNDS lengthen it for soaks by hand."""
    lib = N.load_library()
    rng = np.random.default_rng(int(os.environ.get("MCR_OUT_FUZZ_SEED", "20260106")))
    cfgs = load_golden("paths_injected.json")
    per_path = list(E.SUMMARY_FIELDS) + ["success"]
    traj = ["trajectory", "real_trajectory", "withdrawal_rate_trajectory"]
    vecs = ["counters", "wr_obs_counts", "ruin_year_bins"]
    for _ in range(int(os.environ.get("MCR_OUT_FUZZ_ROUNDS", "40"))):
        g = cfgs[int(rng.integers(len(cfgs)))]
        p = params_from_config(Config(**g["cfg"]))
        wm = int(rng.choice([0, 1, 12, 13, g["working_months"]]))
        n = int(rng.choice([1, 63, 64, 65, 300, int(rng.integers(1, 1500))]))
        stride = n + int(rng.choice([0, 1, 64, 100]))
        seed, stream, begin = int(rng.integers(0, 2**63)), int(rng.integers(2)), int(rng.choice([0, 5, 2**32 - 7]))
        full = E.run_batch_host(p, seed, stream, begin, n, wm)
        sz = E.query_sizes(p, wm)
        want = {k: bool(rng.random() < 0.5) for k in per_path + traj + vecs}
        if rng.random() < 0.1:
            want = {k: False for k in want}
        SENT = -12345.0
        bufs = {}
        o = N.McrOutputs()
        o.path_stride = stride
        for k in per_path:
            bufs[k] = np.full(n + 3, 77, dtype=np.uint8) if k == "success" else np.full(n + 3, SENT)
        for k, rows in (("trajectory", sz.trajectory_len), ("real_trajectory", sz.trajectory_len), ("withdrawal_rate_trajectory", sz.retirement_years)):
            bufs[k] = np.full((rows, stride), SENT)
        bufs["counters"] = np.zeros(N.MCR_N_COUNTERS, dtype=np.uint64)
        bufs["wr_obs_counts"] = np.zeros(sz.retirement_years, dtype=np.uint64)
        bufs["ruin_year_bins"] = np.zeros(sz.ruin_bins, dtype=np.uint64)
        for k, w in want.items():
            if w:
                setattr(o, k, bufs[k].ctypes.data)
        rc = lib.mcr_run_batch_host(C.byref(p), seed, stream, begin, n, wm, None, C.byref(o), 0)
        assert rc == 0, (N.last_error(), want)
        ctx = (g["name"], wm, n, stride, want)
        for k in per_path:
            if want[k]:
                assert np.array_equal(bufs[k][:n], full[k], equal_nan=True), (k, ctx)
                assert np.all(bufs[k][n:] == (77 if k == "success" else SENT)), (k, ctx)
            else:
                assert np.all(bufs[k] == (77 if k == "success" else SENT)), (k, ctx)
        for k in traj:
            if want[k]:
                assert np.array_equal(bufs[k][:, :n], full[k], equal_nan=True), (k, ctx)
                assert np.all(bufs[k][:, n:] == SENT), (k, ctx)
            else:
                assert np.all(bufs[k] == SENT), (k, ctx)
        for k in vecs:
            assert bufs[k].tolist() == (full[k].tolist() if want[k] else [0] * bufs[k].size), (k, ctx)


def test_producer_consumer_split_counts_are_bit_identical(monkeypatch):
    """Small count-only launches (<= 3 072 path-wavefronts: search probes) run the SPLIT form of the path kernel — producer
    waves stage the growth factors a pair of months ahead, consumer waves run the state machine, one barrier per pair, a
    stop vote every 16th.  The arithmetic of a path is the same instructions in the same order: every counter, ruin-year
    bin, observation count and histogram bin must equal the unsplit launch — also when whole workgroups fail early, for
    odd / even / zero working months, ragged sizes, and through the shared-accumulation probes of the search."""
    rng = np.random.default_rng(int(os.environ.get("MCR_SPLIT_FUZZ_SEED", "20261004")))
    cfgs = list(load_golden("paths_injected.json"))
    # the split variants keep the first two income-stream records in SGPRs and read the others every month: a scenario with
    # five streams of every kind (indexed / frozen, open-ended / limited, starting before and after retirement) covers both
    many = dict(cfgs[0]["cfg"], other_income_streams=[
        dict(name="a", monthly_amount_today=1500.0, start_at_age=62.0, duration_years=None, inflation_indexed=True, tax_rate=0.1),
        dict(name="b", monthly_amount_today=900.0, start_at_age=45.0, duration_years=30, inflation_indexed=False, tax_rate=0.2),
        dict(name="c", monthly_amount_today=700.0, start_at_age=70.0, duration_years=10, inflation_indexed=True, tax_rate=0.0),
        dict(name="d", monthly_amount_today=1200.0, start_at_age=66.5, duration_years=None, inflation_indexed=False, tax_rate=0.3),
        dict(name="e", monthly_amount_today=0.0, start_at_age=40.0, duration_years=50, inflation_indexed=False, tax_rate=0.25)])
    five = dict(name="FIVE_STREAMS", cfg=many, working_months=cfgs[0]["working_months"])
    edges = np.geomspace(1.0, 1e12, 41)
    for it in range(int(os.environ.get("MCR_SPLIT_FUZZ_ROUNDS", "24"))):
        g = five if it < 4 else cfgs[int(rng.integers(len(cfgs)))]
        p = params_from_config(Config(**g["cfg"]))
        wm = int(rng.choice([0, 1, 2, 11, 12, 13, g["working_months"], g["working_months"] + 1, int(rng.integers(0, 400))]))
        n = int(rng.choice([1, 63, 64, 65, 255, 257, 1000, 50_000, int(rng.integers(1, 150_000))]))
        seed, stream, begin = int(rng.integers(0, 2**63)), int(rng.integers(2)), int(rng.choice([0, 7, 2**40 + 3]))
        out = {}
        for mode, waves in (("split", "3072"), ("plain", "0")):
            monkeypatch.setenv("MCR_K1_SPLIT_MAX_WAVES", waves)
            out[mode] = E.run_batch_host(p, seed, stream, begin, n, wm, want_summary=False, want_trajectories=False, hist_edges=edges)
        for k in ("counters", "wr_obs_counts", "ruin_year_bins", "hist_bins"):
            assert out["split"][k].tolist() == out["plain"][k].tolist(), (it, g["name"], wm, n, k)
        months = sorted({int(m) for m in rng.integers(0, 300, int(rng.integers(2, 7)))})
        if len(months) >= 2:
            probes = {}
            for mode, waves in (("split", "3072"), ("plain", "0")):
                monkeypatch.setenv("MCR_K1_SPLIT_MAX_WAVES", waves)
                probes[mode] = E.probe_months(p, seed, stream, begin, min(n, 60_000), months).cpu().tolist()
            assert probes["split"] == probes["plain"], (it, g["name"], months, n)
