"""Time-sliced path blocks (PHASE 3 of the path kernel; csrc/mcr_hip.hip): count-only launches of a few rounds of
workgroups cut the first blocks of the launch into segments at retirement-year boundaries, dispatched so that the small work
items come last.  A segment hands its lanes' state over through memory; a successor that does not see its predecessor's flag
recomputes the block itself.  Whatever the route — plain launch, sliced, sliced with every successor recomputing — the
counters, year bins and histogram bins must be the same integers."""

from __future__ import annotations

import os

import numpy as np
import pytest

from conftest import load_golden
from monte_carlo_retirement_amd import Config, params_from_config
from monte_carlo_retirement_amd import engine as E

pytestmark = pytest.mark.gpu

KNOBS = ("MCR_K1_SEGMENTS", "MCR_K1_SEGMENTS_ALWAYS", "MCR_K1_SEGMENT_POLLS")


def _run(p, wm, n, begin, env, edges):
    old = {k: os.environ.get(k) for k in KNOBS}
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(env)
    try:
        r = E.run_batch_host(p, 4242, 1, begin, n, wm, want_summary=False, want_trajectories=False, hist_edges=edges)
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    return np.concatenate([r["counters"], r["ruin_year_bins"], r["wr_obs_counts"], r["hist_bins"]]).astype(np.int64)


def _scenarios():
    inj = {g["name"]: g for g in load_golden("paths_injected.json")}
    c1 = inj["C1_config_json_wm233"]["cfg"]                       # one frozen stream (a lock column travels with the state)
    annual = inj["ANNUAL_wm50"]["cfg"]                            # annual-gains tax: the gain accumulators travel too
    failing = inj["FAILING_wm24"]["cfg"]                          # half of the paths fail: dead lanes and dead waves cross segments
    yield "config.json wm=233 (odd resume rows, terminal tax period)", c1, 233
    yield "config.json wm=240", c1, 240
    yield "annual tax wm=50", annual, 50
    yield "failing wm=24", failing, 24
    yield "failing wm=7", dict(failing, retirement_years=9), 7    # fewer years than 2 x the segment count: the count is lowered to 4


@pytest.mark.parametrize("n", [400_000, 393_216 + 999])
def test_sliced_launch_counts_equal_the_plain_launch(n):
    edges = np.geomspace(1.0, 1e13, 65)
    for name, cfgd, wm in _scenarios():
        p = params_from_config(Config(**cfgd))
        plain = _run(p, wm, n, 2**33 + 5, {"MCR_K1_SEGMENTS": "0"}, edges)
        assert int(plain[1]) == n and 0 < int(plain[0]) <= n, name
        for env in ({"MCR_K1_SEGMENTS_ALWAYS": "1"},                                    # the launcher's segment count, hand-over through memory
                    {"MCR_K1_SEGMENTS_ALWAYS": "1", "MCR_K1_SEGMENTS": "4"},
                    {"MCR_K1_SEGMENTS_ALWAYS": "1", "MCR_K1_SEGMENTS": "2"},
                    {"MCR_K1_SEGMENTS_ALWAYS": "1", "MCR_K1_SEGMENTS": "7"},
                    {"MCR_K1_SEGMENTS_ALWAYS": "1", "MCR_K1_SEGMENT_POLLS": "0"},       # every successor recomputes its block from month 0
                    {}):                                                               # the launcher's own rule
            got = _run(p, wm, n, 2**33 + 5, env, edges)
            assert np.array_equal(got, plain), (name, n, env, np.nonzero(got != plain)[0][:8].tolist())


def _run_outputs(p, wm, n, begin, env, want_trajectories):
    old = {k: os.environ.get(k) for k in KNOBS}
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(env)
    try:
        return E.run_batch_host(p, 4242, 1, begin, n, wm, want_trajectories=want_trajectories)
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


@pytest.mark.parametrize("want_trajectories", [False, True])
def test_sliced_launch_outputs_equal_the_plain_launch(want_trajectories):
    """The variants with per-path outputs slice too (their occupancy is five workgroups per CU: 10^6 paths are 3.05 rounds): the
    hand-over state also carries the balance and price level at retirement and the three write-once columns; every per-path
    field, trajectory sample and withdrawal rate must come out bit-identical, whichever segment wrote it."""
    n = 330_000                      # 1 290 path blocks on 1 280 resident slots
    for name, cfgd, wm in _scenarios():
        p = params_from_config(Config(**cfgd))
        plain = _run_outputs(p, wm, n, 7, {"MCR_K1_SEGMENTS": "0"}, want_trajectories)
        for env in ({"MCR_K1_SEGMENTS_ALWAYS": "1"}, {"MCR_K1_SEGMENTS_ALWAYS": "1", "MCR_K1_SEGMENTS": "3"},
                    {"MCR_K1_SEGMENTS_ALWAYS": "1", "MCR_K1_SEGMENT_POLLS": "0"}, {}):
            got = _run_outputs(p, wm, n, 7, env, want_trajectories)
            assert set(got) == set(plain)
            for k in plain:
                assert np.array_equal(got[k], plain[k], equal_nan=True), (name, env, k)


def test_sliced_probe_window_equals_plain_and_single_launches():
    """The search's verification window (17 consecutive candidate months over 50 000 paths: 17 x 196 workgroups = 2.17 rounds of
    the resident slots) resumes every candidate from its accumulation snapshot AND time-slices the decumulations (PHASE 4):
    per-candidate counters equal the plain shared-prefix route, the all-recompute route and one launch per candidate."""
    cfgd = [g for g in load_golden("paths_injected.json") if g["name"] == "C1_config_json_wm233"][0]["cfg"]
    p = params_from_config(Config(**cfgd))
    months = list(range(217, 234))
    n = 50_000

    def probes(env):
        old = {k: os.environ.get(k) for k in KNOBS}
        for k in KNOBS:
            os.environ.pop(k, None)
        os.environ.update(env)
        try:
            return E.probe_months(p, 4242, 0, 11, n, months).cpu().numpy()
        finally:
            for k, v in old.items():
                os.environ.pop(k, None)
                if v is not None:
                    os.environ[k] = v

    plain = probes({"MCR_K1_SEGMENTS": "0"})
    assert plain[:, 1].tolist() == [n] * len(months) and 0 < int(plain[0, 0]) <= int(plain[-1, 0]) <= n
    for env in ({}, {"MCR_K1_SEGMENTS_ALWAYS": "1", "MCR_K1_SEGMENTS": "3"}, {"MCR_K1_SEGMENT_POLLS": "0"}):
        assert np.array_equal(probes(env), plain), env
    for m, row in zip(months[::4], plain[::4]):
        one = E.run_batch_host(p, 4242, 0, 11, n, m, want_summary=False, want_trajectories=False)
        assert row.tolist() == one["counters"].tolist(), m


def test_sliced_launch_fuzz():
    """Random scenarios (the reference-fixture fuzz set: zero allocations, annual taxes, frozen and indexed streams, odd working
    months, terminal tax periods, 1-8 retirement years ...), random batch sizes just above the resident capacity, random segment
    counts and output modes: sliced == plain in every output.  MCR_SLICE_FUZZ_ROUNDS / MCR_SLICE_FUZZ_SEED for soaks."""
    rng = np.random.default_rng(int(os.environ.get("MCR_SLICE_FUZZ_SEED", "2026")))
    groups = [g for g in load_golden("paths_fuzz.json") if len(g["cfg"]["other_income_streams"]) <= 16] + load_golden("paths_injected.json")
    checked = 0
    for _ in range(int(os.environ.get("MCR_SLICE_FUZZ_ROUNDS", "10"))):
        g = groups[int(rng.integers(len(groups)))]
        cfgd = dict(g["cfg"], retirement_years=int(rng.choice([g["cfg"]["retirement_years"], 4, 7, 12, 30])))
        p = params_from_config(Config(**cfgd))
        wm = int(rng.choice([g["working_months"], 0, 1, 11, 12, 13, int(rng.integers(0, 60))]))
        mode = int(rng.integers(3))
        slots = 256 * (6 if mode == 0 else 5)
        n = int(rng.integers(slots * 256 + 1, slots * 256 * 2))
        if mode == 2:
            n = min(n, 360_000)                       # (host buffers of the trajectory outputs)
        env = {"MCR_K1_SEGMENTS_ALWAYS": "1", "MCR_K1_SEGMENTS": str(int(rng.choice([2, 3, 4, 6, 8])))}
        if rng.random() < 0.25:
            env["MCR_K1_SEGMENT_POLLS"] = "0"
        begin = int(rng.choice([0, 2**32 - 3, 2**40 + 17]))
        kw = dict(want_summary=mode >= 1, want_trajectories=mode == 2)

        def run(e):
            old = {k: os.environ.get(k) for k in KNOBS}
            for k in KNOBS:
                os.environ.pop(k, None)
            os.environ.update(e)
            try:
                return E.run_batch_host(p, 99, int(rng.integers(2)) * 0 + 1, begin, n, wm, **kw)
            finally:
                for k, v in old.items():
                    os.environ.pop(k, None)
                    if v is not None:
                        os.environ[k] = v

        plain, sliced = run({"MCR_K1_SEGMENTS": "0"}), run(env)
        for k in plain:
            assert np.array_equal(plain[k], sliced[k], equal_nan=True), (g["name"], wm, cfgd["retirement_years"], mode, n, env, k)
        checked += 1
    assert checked > 0
