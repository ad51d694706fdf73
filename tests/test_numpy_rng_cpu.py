"""The NumPy random stream the engine reproduces (rng="numpy"), checked on the CPU: the executable
model of SeedSequence / PCG64 / ziggurat (tools/numpy_rng_model.py — the blueprint of
csrc/mcr_numpy_rng.h) against the installed NumPy, and the committed device tables against the
tables that model validates."""

from __future__ import annotations

import os
import re
import sys

import numpy as np
import pytest

from conftest import GOLDEN, REPO

sys.path.insert(0, os.path.join(REPO, "tools"))
import numpy_rng_model as M  # noqa: E402


@pytest.fixture(scope="module")
def tables():
    t = np.load(os.path.join(GOLDEN, "numpy_ziggurat_tables.npz"))
    return [int(v) for v in t["ki"]], [float(v) for v in t["wi"]], [float(v) for v in t["fi"]]


def test_seedsequence_model_matches_numpy():
    for ent, key in [(0, ()), (12345, ()), (12345, (1,)), (12345, (1, 7)), (2**40 + 3, (0, 99)), (2**130 + 5, (1, 2)),
                     (7, (1, 2**32 + 5))]:
        ss = np.random.SeedSequence(ent, spawn_key=key)
        assert [int(v) for v in ss.pool] == M.seedseq_pool(ent, _key_words(key))
        assert [int(v) for v in ss.generate_state(9)] == M.generate_state_u32(M.seedseq_pool(ent, _key_words(key)), 9)
    final = np.random.SeedSequence(12345).spawn(2)[1]
    assert [int(k.generate_state(1)[0]) for k in final.spawn(6)] == [M.path_seed_u32(12345, 1, j) for j in range(6)]


def _key_words(key):
    out = []
    for k in key:  # numpy coerces each spawn-key element to uint32 words
        out += M._u32_words(k)
    return tuple(out)


def test_pcg64_model_matches_numpy():
    for seed in (0, 1, 42, 3735928559, 2**32 - 1):
        g = M.PCG64(seed)
        assert [g.next64() for _ in range(32)] == [int(v) for v in np.random.PCG64(seed).random_raw(32)]


def test_ziggurat_model_matches_numpy_bit_for_bit(tables):
    z = M.Ziggurat(*tables)
    for seed in (5, 2024, 987654321):
        g = M.PCG64(seed)
        ref = np.random.default_rng(seed).standard_normal(150_000)
        mine = np.array([z.normal(g) for _ in range(150_000)])
        assert np.array_equal(mine.view(np.uint64), ref.view(np.uint64))


def test_device_table_header_is_the_validated_table(tables):
    hdr = open(os.path.join(REPO, "monte_carlo_retirement_amd", "csrc", "mcr_numpy_tables.h")).read()
    ki = [int(x, 16) for x in re.findall(r"0x([0-9a-f]{16})ull", hdr)]
    flo = [float.fromhex(x) for x in re.findall(r"^\s+(-?0x[0-9a-f.]+p[-+]\d+),", hdr, flags=re.M)]
    assert ki == tables[0]
    assert flo[:256] == tables[1] and flo[256:512] == tables[2]
