"""CPU-only checks of the C-ABI library: it loads, exports every symbol include/mcr.h
declares, its host-side derivations match the reference's, and without a GPU every compute
entry point fails loudly (there is no CPU fallback)."""

from __future__ import annotations

import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import REPO, load_golden
from monte_carlo_retirement_amd import Config, params_from_config
from monte_carlo_retirement_amd import _native as N
from monte_carlo_retirement_amd import engine as E


@pytest.fixture(scope="module")
def lib():
    from monte_carlo_retirement_amd.csrc import build

    build.build()
    return N.load_library()


def test_header_and_binding_agree_on_symbols(lib):
    hdr = open(os.path.join(REPO, "include", "mcr.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)  # prose in comments mentions call syntax too
    declared = set(re.findall(r"\b(mcr_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(N.ABI_SYMBOLS), declared ^ set(N.ABI_SYMBOLS)
    for sym in declared:
        assert hasattr(lib, sym), f"libmcr_hip.so does not export {sym}"
    assert lib.mcr_abi_version() == N.MCR_ABI_VERSION


def test_struct_layout_matches_header():
    assert C.sizeof(N.McrStream) == 32
    assert C.sizeof(N.McrParams) == 17 * 8 + 4 * 4 + N.MCR_INLINE_STREAMS * 32 + 8
    assert C.sizeof(N.McrSizes) == 24
    assert C.sizeof(N.McrOutputs) == 17 * 8


def test_struct_layout_matches_what_a_c_compiler_sees(tmp_path):
    """sizeof / offsetof of every ABI struct as gcc lays them out from include/mcr.h (C99) vs the ctypes mirrors."""
    import subprocess

    structs = {"mcr_stream": N.McrStream, "mcr_params": N.McrParams, "mcr_sizes": N.McrSizes, "mcr_outputs": N.McrOutputs,
               "mcr_rng": N.McrRng}
    lines = []
    for cname, T in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in T._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "mcr.h"\nint main(void) {\n' + "\n".join(lines) + "\nreturn 0; }\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(REPO, "include"), "-o", str(exe), str(src)])
    seen = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, T in structs.items():
        assert int(seen[cname]) == C.sizeof(T), cname
        for fname, _ in T._fields_:
            assert int(seen[f"{cname}.{fname}"]) == getattr(T, fname).offset, f"{cname}.{fname}"


def test_host_derivations_match_reference(lib):
    g = load_golden("helpers.json")
    for r in g["stream_start_month_index"]:
        assert E.stream_start_month_index(r["current_age"], r["working_months"], r["start_at_age"]) == r["start_month"]
    cfg = Config(**load_golden("paths_injected.json")[0]["cfg"])
    sz = E.query_sizes(params_from_config(cfg), 233)
    assert (sz.total_months, sz.shock_rows, sz.num_working_years, sz.trajectory_len, sz.retirement_years, sz.ruin_bins) == (833, 833, 20, 71, 50, 52)
    sz = E.query_sizes(params_from_config(cfg), 0)
    assert (sz.total_months, sz.num_working_years, sz.trajectory_len) == (600, 0, 51)
    with pytest.raises(ValueError):
        E.query_sizes(params_from_config(cfg), -1)


def test_validate_params_rejects_what_the_config_would(lib):
    """The kernel drops clamps that are no-ops only inside the Config's ranges (config.py:56-99): a caller that
    bypasses pydantic must get MCR_ERR_INVALID_ARG, not different arithmetic.  No device is needed."""
    cfg = Config(**load_golden("paths_injected.json")[0]["cfg"])
    good = params_from_config(cfg)
    assert lib.mcr_validate_params(C.byref(good)) == 0
    for field, value in [("inv1_realized_gains_tax_rate", 1.5), ("inv2_realized_gains_tax_rate", -0.1),
                         ("allocation_inv1_pct", 1.0000001), ("inv1_annual_tax_on_gains_rate", float("nan")),
                         ("initial_balance", -1.0), ("monthly_expenses", float("inf")), ("monthly_contribution", -5.0),
                         ("contribution_growth_rate_annual", -0.01), ("equity_inflation_rho", 1.2),
                         ("inv1_sigma_log", -0.1), ("inf_mu_log", float("nan")), ("prem_sigma_log", 100.0),
                         ("n_streams", N.MCR_INLINE_STREAMS + 1),   # (beyond the block with extra_streams == NULL)
                         ("n_streams", -1)]:
        bad = params_from_config(cfg)
        setattr(bad, field, value)
        assert lib.mcr_validate_params(C.byref(bad)) == -1, field
        assert field.split("_mu_log")[0].split("_sigma_log")[0] in N.last_error(), (field, N.last_error())
    bad = params_from_config(cfg)
    assert bad.n_streams >= 1
    bad.streams[0].tax_rate = 2.0
    assert lib.mcr_validate_params(C.byref(bad)) == -1 and "streams[0]" in N.last_error()
    # the compute entry points apply the same check before anything else that needs a device... or after: either
    # way an out-of-range block never computes
    bad = params_from_config(cfg)
    bad.inv1_realized_gains_tax_rate = 1.5
    o = N.McrOutputs()
    assert lib.mcr_run_batch_host(C.byref(bad), 1, 1, 0, 8, 12, None, C.byref(o), 0) != 0
    assert lib.mcr_validate_params(None) == -1


def test_stream_lists_of_any_length_are_valid(lib):
    """other_income_streams has no length limit in the reference (backend/config.py:99): 16 records sit in the block, the
    rest behind `extra_streams`; the host-side checks walk the whole list."""
    cfgd = dict(load_golden("paths_fuzz.json")[-1]["cfg"])          # streams40_mixed_annual_tax
    assert len(cfgd["other_income_streams"]) == 40
    p = params_from_config(Config(**cfgd))
    assert p.n_streams == 40 and bool(p.extra_streams)
    for i, st in enumerate(cfgd["other_income_streams"]):
        rec = p.stream(i)
        assert (rec.monthly_amount_today, rec.start_at_age, rec.tax_rate) == (st["monthly_amount_today"], st["start_at_age"], st["tax_rate"])
        assert rec.duration_years == (-1 if st["duration_years"] is None else st["duration_years"])
        assert rec.inflation_indexed == int(st["inflation_indexed"])
    assert lib.mcr_validate_params(C.byref(p)) == 0
    assert E.query_sizes(p, 49).trajectory_len == 1 + 5 + cfgd["retirement_years"]
    p.stream(39).monthly_amount_today = -1.0
    assert lib.mcr_validate_params(C.byref(p)) == -1 and "streams[39]" in N.last_error()
    few = params_from_config(Config(**dict(cfgd, other_income_streams=cfgd["other_income_streams"][:16])))
    assert few.n_streams == 16 and not bool(few.extra_streams)


def test_no_cpu_fallback(lib):
    """Without a HIP device the compute entry points must fail loudly, never compute on the CPU."""
    if lib.mcr_device_count() > 0:
        pytest.skip("a GPU is present")
    cfg = Config(**load_golden("paths_injected.json")[0]["cfg"])
    with pytest.raises(RuntimeError, match="no usable HIP device"):
        E.run_batch_host(params_from_config(cfg), 1, 1, 0, 8, 12)
    o = N.McrOutputs()
    rc = lib.mcr_run_batch_host(C.byref(params_from_config(cfg)), 1, 1, 0, 8, 12, None, C.byref(o), 0)
    assert rc == -2 and "no usable HIP device" in N.last_error()
    rc = lib.mcr_run_batch(C.byref(params_from_config(cfg)), 1, 1, 0, 8, 12, None, C.byref(o), 0, None)
    assert rc == -2
    with pytest.raises(RuntimeError):
        E.eval_helper_host(N.MCR_HELPER_NLV, None, [[1.0, 1.0, 0.0, 0.0]])
    with pytest.raises(RuntimeError):
        E.draw_shocks_host(1, 1, 0, 1, 4, 0.0)


def test_product_package_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under monte_carlo_retirement_amd/ may reference it."""
    root = os.path.join(REPO, "monte_carlo_retirement_amd")
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "liboracle" not in text and "mcr_oracle" not in text, f


def build_c_caller(tmp_path) -> str:
    """Compile examples/c_caller.c (plain C99 against include/mcr.h, linked with the built library)."""
    import subprocess

    exe = os.path.join(str(tmp_path), "c_caller")
    libdir = os.path.join(REPO, "monte_carlo_retirement_amd", "csrc")
    cmd = ["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(REPO, "include"),
           os.path.join(REPO, "examples", "c_caller.c"), "-o", exe, "-L", libdir, "-lmcr_hip", f"-Wl,-rpath,{libdir}", "-lm"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_plain_c_caller_compiles_and_fails_loudly_without_a_gpu(lib, tmp_path):
    """The boundary is a C ABI: the header is valid C99 (-pedantic -Werror), a C program links against the library
    with nothing but libm, and without a HIP device it exits with an error instead of computing anything."""
    import subprocess

    exe = build_c_caller(tmp_path)
    if lib.mcr_device_count() > 0:
        pytest.skip("a GPU is present (the run itself is tests/test_gpu_edge_cases.py)")
    r = subprocess.run([exe, "1000"], capture_output=True, text=True)
    assert r.returncode == 3 and "no HIP device" in r.stderr and r.stdout == ""


def test_sample_columns_equals_numpy_randomstate_choice(lib):
    """`mcr_sample_columns` restates RandomState(seed).choice(n, k, replace=False) — what pandas' sample(n=5, axis=1,
    random_state=main_seed) draws in the reference (simulation.py:1063-1078) — without shuffling n indices: the same
    MT19937 draws, the k positions traced back through the swaps.  Host-only; must equal NumPy on every input."""
    from monte_carlo_retirement_amd.simulation import sample_columns

    rng = np.random.default_rng(20261004)
    cases = [(0, 1, 1), (0, 2, 2), (2**32 - 1, 5, 5), (12345, 64, 64), (7, 65, 64), (99, 2**16, 5), (99, 2**16 + 1, 5), (1, 2**20 - 1, 3)]
    for _ in range(400):
        n = int(rng.choice([1, 2, 3, 5, 6, 7, 63, 64, 65, 1000, 4096, 65537, int(rng.integers(5, 400_000))]))
        cases.append((int(rng.integers(0, 2**32)), n, min(n, int(rng.integers(1, 9)))))
    for seed, n, k in cases:
        got = sample_columns(seed, n, k)
        exp = np.random.RandomState(seed).choice(n, size=k, replace=False)
        assert got.dtype == np.int64 and np.array_equal(got, exp), (seed, n, k, got, exp)
    # the scenario files' shape: 5 of 10^6 for the seeds the tests and the bench use
    for seed in (12345, 2024, 99):
        assert np.array_equal(sample_columns(seed, 1_000_000, 5), np.random.RandomState(seed).choice(1_000_000, size=5, replace=False))
    # outside the restatement's domain: NumPy's own path (more than 64 picks) or NumPy's own error (seed >= 2**32 -> None, logged)
    assert np.array_equal(sample_columns(5, 1000, 100), np.random.RandomState(5).choice(1000, size=100, replace=False))
    assert sample_columns(2**32, 1000, 5) is None
    assert sample_columns(5, 3, 5) is None                      # more picks than paths: NumPy raises, the reference logs
    out = (C.c_int64 * 4)()
    for seed, n, k in ((1, 10, 0), (1, 10, 65), (1, 3, 4), (1, 2**32 + 1, 2)):
        assert lib.mcr_sample_columns(seed, n, k, out) == -1 and "mcr_sample_columns" in N.last_error()
    assert lib.mcr_sample_columns(1, 10, 2, None) == -1
