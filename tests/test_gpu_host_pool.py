"""Host-buffer entry points lease their stream + scratch block from a process-wide pool (csrc/mcr_host.h): short-lived
caller threads — the reference's server runs every request on an executor thread, backend/server.py:309,405 — and the
per-call shard workers of the multi-device entry leave nothing behind.  Round 2 kept one context per (thread, device) in
thread-local storage and never freed it: device memory grew with every call of mcr_run_batch_multi_host_rng."""

from __future__ import annotations

import json
import os
import threading

import numpy as np
import pytest

from conftest import REPO
from monte_carlo_retirement_amd import Config, params_from_config
from monte_carlo_retirement_amd import _native as N
from monte_carlo_retirement_amd import engine as E

pytestmark = pytest.mark.gpu
MiB = 1 << 20


def _params():
    with open(os.path.join(REPO, "scenarios", "config.json")) as fh:
        return params_from_config(Config(**dict(json.load(fh), seed=12345)))


def _free_bytes():
    import torch

    torch.cuda.synchronize()
    return int(torch.cuda.mem_get_info(0)[0])


def test_sixty_four_short_lived_threads_leave_no_device_memory_behind():
    p = _params()
    lib = N.load_library()
    n, wm = 400_000, 12                       # 49 B/path of summary output: a 19 MiB scratch block per context
    ref = E.run_batch_host(p, 5, 1, 0, n, wm, want_trajectories=False)
    assert lib.mcr_release_cached(-1) == 0
    base = _free_bytes()
    errors = []

    def call(i):
        try:
            r = E.run_batch_host(p, 5, 1, 0, n, wm, want_trajectories=False)
            if not np.array_equal(r["final_balance"], ref["final_balance"]) or r["counters"].tolist() != ref["counters"].tolist():
                errors.append(f"thread {i}: results differ")
        except Exception as exc:  # noqa: BLE001
            errors.append(f"thread {i}: {exc!r}")

    for i in range(64):                       # one after the other: each thread is gone before the next starts
        t = threading.Thread(target=call, args=(i,))
        t.start()
        t.join()
    assert not errors, errors[:3]
    held = base - _free_bytes()
    assert held <= 4 * 20 * MiB + 32 * MiB, f"{held / MiB:.0f} MiB still held after 64 sequential threads (pool keeps <= 4 idle contexts)"
    # eight at a time: eight contexts while they run, at most four kept afterwards
    for _ in range(4):
        ts = [threading.Thread(target=call, args=(100 + k,)) for k in range(8)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    assert not errors, errors[:3]
    held = base - _free_bytes()
    assert held <= 4 * 20 * MiB + 32 * MiB, f"{held / MiB:.0f} MiB held after concurrent callers"
    assert lib.mcr_release_cached(-1) == 0
    assert abs(base - _free_bytes()) <= 32 * MiB, "mcr_release_cached did not return the pool's memory"


def test_multi_device_entry_called_fifty_times_keeps_memory_flat():
    """every call of mcr_run_batch_multi_host_rng starts one worker thread per shard; their contexts were never
    reused or freed.  50 calls with devices=[0, 0] (2 x 24 MiB blocks each): free memory must stay flat."""
    p = _params()
    lib = N.load_library()
    n, wm = 1_000_000, 12
    assert lib.mcr_release_cached(-1) == 0
    base = _free_bytes()
    first = None
    lows = []
    for i in range(50):
        r = E.run_batch_host(p, 7, 1, 0, n, wm, want_trajectories=False, devices=[0, 0])
        if first is None:
            first = r
        else:
            assert np.array_equal(r["final_balance"], first["final_balance"]) and r["counters"].tolist() == first["counters"].tolist()
        if i in (4, 49):
            lows.append(base - _free_bytes())
    assert lows[1] <= lows[0] + 16 * MiB, f"device memory grew from {lows[0] / MiB:.0f} to {lows[1] / MiB:.0f} MiB held over 45 calls"
    assert lows[1] <= 4 * 25 * MiB + 32 * MiB
    one = E.run_batch_host(p, 7, 1, 0, n, wm, want_trajectories=False)
    assert np.array_equal(one["final_balance"], first["final_balance"])       # sharding never changes the numbers
    assert lib.mcr_release_cached(0) == 0
    assert abs(base - _free_bytes()) <= 32 * MiB


def test_probe_side_streams_are_leased_too():
    """mcr_probe_months_rng's fork/join streams come from the same kind of pool: callers on many short-lived threads get
    the same counts as one caller, and mcr_release_cached() may run at any time in between."""
    import torch

    p = _params()
    lib = N.load_library()
    months = [30, 30, 18, 6]                  # a duplicate candidate forces the forked one-launch-per-candidate route
    want = E.probe_months(p, 3, 0, 0, 20_000, months).cpu().tolist()
    got, errors = {}, []

    def call(i):
        try:
            got[i] = E.probe_months(p, 3, 0, 0, 20_000, months).cpu().tolist()
        except Exception as exc:  # noqa: BLE001
            errors.append(repr(exc))

    for i in range(24):
        t = threading.Thread(target=call, args=(i,))
        t.start()
        t.join()
        if i % 8 == 7:
            assert lib.mcr_release_cached(-1) == 0
    ts = [threading.Thread(target=call, args=(100 + k,)) for k in range(6)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors[:3]
    assert all(v == want for v in got.values()) and len(got) == 30
    assert want[0] == want[1]
