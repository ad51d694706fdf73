"""ctypes binding of the C ABI declared in include/mcr.h (``csrc/libmcr_hip.so``).

This is the only door into the HIP kernels.  There is no CPU fallback: if the shared
library is missing, or no HIP device is usable, the compute entry points raise
``RuntimeError`` — loudly, by design.
"""

from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional

MCR_ABI_VERSION = 7
MCR_INLINE_STREAMS = 16  # other_income_streams records inside mcr_params; the rest follow through extra_streams
MCR_N_COUNTERS = 2
MCR_N_STAT_ROWS = 4
MCR_CTR_SUCCESS = 0
MCR_CTR_PATHS = 1
MCR_STREAM_SEARCH = 0
MCR_STREAM_FINAL = 1
MCR_RNG_PHILOX = 0
MCR_RNG_NUMPY = 1
MCR_MAX_ENTROPY_WORDS = 8
MCR_DEVICE_ALL = -2
MCR_MAX_HIST_BINS = 4096

MCR_HELPER_WITHDRAW = 0
MCR_HELPER_NLV = 1
MCR_HELPER_REBALANCE = 2
MCR_HELPER_ANNUAL_TAX = 3
MCR_HELPER_MONTHLY_GROSS = 4
MCR_HELPER_MATH_EXP = 5
MCR_HELPER_MATH_DIV = 6
MCR_HELPER_MATH_SQRT = 7
MCR_HELPER_MATH_NEG2LOG = 8
MCR_HELPER_MATH_SINCOS = 9
MCR_HELPER_MATH_DIV_PATH = 10
MCR_HELPER_WITHDRAW2_PATH = 11
MCR_HELPER_NLV2_PATH = 12
MCR_HELPER_REBALANCE_PATH = 13
MCR_HELPER_ANNUAL_TAX_PATH = 14
MCR_HELPER_MATH_EXP_PATH = 15
MCR_HELPER_MATH_NEG2LOG_PATH = 16
MCR_HELPER_MATH_SINCOS_PATH = 17
MCR_HELPER_WITHDRAW_MONTH = 18
MCR_HELPER_REBALANCE_MONTH = 19
_HELPER_ARITY = {  # which -> (n_in, n_out)
    MCR_HELPER_WITHDRAW: (5, 4),
    MCR_HELPER_NLV: (4, 1),
    MCR_HELPER_REBALANCE: (4, 4),
    MCR_HELPER_ANNUAL_TAX: (6, 5),
    MCR_HELPER_MONTHLY_GROSS: (3, 1),
    MCR_HELPER_MATH_EXP: (1, 1),
    MCR_HELPER_MATH_DIV: (2, 1),
    MCR_HELPER_MATH_SQRT: (1, 1),
    MCR_HELPER_MATH_NEG2LOG: (1, 1),
    MCR_HELPER_MATH_SINCOS: (1, 2),
    MCR_HELPER_MATH_DIV_PATH: (2, 1),
    MCR_HELPER_WITHDRAW2_PATH: (6, 8),
    MCR_HELPER_NLV2_PATH: (4, 2),
    MCR_HELPER_REBALANCE_PATH: (4, 4),
    MCR_HELPER_ANNUAL_TAX_PATH: (6, 5),
    MCR_HELPER_MATH_EXP_PATH: (1, 1),
    MCR_HELPER_MATH_NEG2LOG_PATH: (1, 1),
    MCR_HELPER_MATH_SINCOS_PATH: (1, 2),
    MCR_HELPER_WITHDRAW_MONTH: (5, 6),
    MCR_HELPER_REBALANCE_MONTH: (4, 4),
}


#: include/mcr.h: mcr_reduce_fn — int (*)(void* ctx, void* device_buf, int64_t count, int32_t dtype)
REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32)
MCR_DT_I32, MCR_DT_I64 = 0, 1


class McrStream(C.Structure):
    _fields_ = [
        ("monthly_amount_today", C.c_double),
        ("start_at_age", C.c_double),
        ("tax_rate", C.c_double),
        ("duration_years", C.c_int32),
        ("inflation_indexed", C.c_int32),
    ]


class McrParams(C.Structure):
    _fields_ = [
        ("initial_balance", C.c_double),
        ("monthly_contribution", C.c_double),
        ("contribution_growth_rate_annual", C.c_double),
        ("monthly_expenses", C.c_double),
        ("current_age", C.c_double),
        ("allocation_inv1_pct", C.c_double),
        ("inv1_annual_tax_on_gains_rate", C.c_double),
        ("inv1_realized_gains_tax_rate", C.c_double),
        ("inv2_annual_tax_on_gains_rate", C.c_double),
        ("inv2_realized_gains_tax_rate", C.c_double),
        ("inv1_mu_log", C.c_double),
        ("inv1_sigma_log", C.c_double),
        ("inf_mu_log", C.c_double),
        ("inf_sigma_log", C.c_double),
        ("prem_mu_log", C.c_double),
        ("prem_sigma_log", C.c_double),
        ("equity_inflation_rho", C.c_double),
        ("retirement_years", C.c_int32),
        ("inv1_use_realized_gains_tax_system", C.c_int32),
        ("inv2_use_realized_gains_tax_system", C.c_int32),
        ("n_streams", C.c_int32),
        ("streams", McrStream * MCR_INLINE_STREAMS),
        ("extra_streams", C.POINTER(McrStream)),
    ]

    #: the ctypes array `extra_streams` points into (a Structure keeps no reference to what its pointer fields point at)
    _extra_keepalive = None

    def set_streams(self, records) -> None:
        """Fill the stream list from `(monthly_amount_today, start_at_age, tax_rate, duration_years, inflation_indexed)`
        tuples — ANY number of them (backend/config.py:99 has no limit): the first MCR_INLINE_STREAMS go into the block, the
        rest into an array this object keeps alive and `extra_streams` points at."""
        records = list(records)
        self.n_streams = len(records)
        extra = (McrStream * max(0, len(records) - MCR_INLINE_STREAMS))()
        for i, (amount, start_age, tax_rate, duration_years, indexed) in enumerate(records):
            s = self.streams[i] if i < MCR_INLINE_STREAMS else extra[i - MCR_INLINE_STREAMS]
            s.monthly_amount_today = amount
            s.start_at_age = start_age
            s.tax_rate = tax_rate
            s.duration_years = -1 if duration_years is None else int(duration_years)
            s.inflation_indexed = int(bool(indexed))
        self._extra_keepalive = extra if len(extra) else None
        self.extra_streams = C.cast(extra, C.POINTER(McrStream)) if len(extra) else C.POINTER(McrStream)()

    def stream(self, i: int) -> McrStream:
        """Entry i of the list, wherever it lives."""
        return self.streams[i] if i < MCR_INLINE_STREAMS else self.extra_streams[i - MCR_INLINE_STREAMS]


class McrRng(C.Structure):
    """Random-stream descriptor (include/mcr.h: mcr_rng)."""

    _fields_ = [
        ("kind", C.c_uint32),
        ("n_entropy_words", C.c_uint32),
        ("entropy", C.c_uint32 * MCR_MAX_ENTROPY_WORDS),
        ("philox_seed", C.c_uint64),
        ("child_offset", C.c_uint64),
        ("path_seeds", C.c_void_p),
    ]


def philox_rng(seed: int) -> McrRng:
    r = McrRng()
    r.kind = MCR_RNG_PHILOX
    r.philox_seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return r


def numpy_rng(main_seed: int, child_offset: int = 0, path_seeds_ptr: int = 0) -> McrRng:
    """The reference's NumPy stream for `main_seed` (SeedSequence entropy words, least significant first)."""
    words, s = [], int(main_seed)
    if s < 0:
        raise ValueError("seed must be nonnegative")
    while True:
        words.append(s & 0xFFFFFFFF)
        s >>= 32
        if s == 0:
            break
    if len(words) > MCR_MAX_ENTROPY_WORDS:
        raise ValueError(f"numpy rng: seeds above {32 * MCR_MAX_ENTROPY_WORDS} bits are not supported")
    r = McrRng()
    r.kind = MCR_RNG_NUMPY
    r.n_entropy_words = len(words)
    for i, w in enumerate(words):
        r.entropy[i] = w
    r.child_offset = int(child_offset)
    r.path_seeds = path_seeds_ptr or None
    return r


class McrSizes(C.Structure):
    _fields_ = [
        ("total_months", C.c_int32),
        ("shock_rows", C.c_int32),
        ("num_working_years", C.c_int32),
        ("trajectory_len", C.c_int32),
        ("retirement_years", C.c_int32),
        ("ruin_bins", C.c_int32),
    ]


class McrOutputs(C.Structure):
    """Pointers are plain addresses (device or host, depending on the entry point)."""

    _fields_ = [
        ("start_balance", C.c_void_p),
        ("final_balance", C.c_void_p),
        ("years_to_ruin", C.c_void_p),
        ("first_year_gross_withdrawal", C.c_void_p),
        ("first_year_real_gross_withdrawal", C.c_void_p),
        ("inflation_at_retirement", C.c_void_p),
        ("success", C.c_void_p),
        ("trajectory", C.c_void_p),
        ("real_trajectory", C.c_void_p),
        ("withdrawal_rate_trajectory", C.c_void_p),
        ("path_stride", C.c_int64),
        ("counters", C.c_void_p),
        ("wr_obs_counts", C.c_void_p),
        ("ruin_year_bins", C.c_void_p),
        ("hist_edges", C.c_void_p),
        ("hist_bins", C.c_void_p),
        ("hist_n_bins", C.c_int32),
        ("hist_reserved", C.c_int32),
    ]


#: The symbols include/mcr.h declares; tests check that the built library exports them all.
ABI_SYMBOLS = (
    "mcr_abi_version",
    "mcr_device_count",
    "mcr_last_error",
    "mcr_query_sizes",
    "mcr_stream_start_month_index",
    "mcr_run_batch",
    "mcr_run_batch_host",
    "mcr_draw_shocks_host",
    "mcr_run_batch_rng",
    "mcr_run_batch_host_rng",
    "mcr_draw_shocks_host_rng",
    "mcr_probe_months_rng",
    "mcr_run_batch_multi_host_rng",
    "mcr_validate_params",
    "mcr_release_cached",
    "mcr_sample_columns",
    "mcr_eval_helper_host",
    "mcr_row_quantiles_scratch_bytes",
    "mcr_row_quantiles",
    "mcr_row_quantiles_last_fallback_rows",
    "mcr_row_quantiles_sharded",
    "mcr_row_quantiles_reduce_block",
    "mcr_row_quantiles_begin",
    "mcr_row_quantiles_hist",
    "mcr_row_quantiles_scan",
    "mcr_minmax_success",
    "mcr_histogram_success",
    "mcr_summary_stat_rows",
)

_LIB_NAME = "libmcr_hip.so"
_lib: Optional[C.CDLL] = None
_lib_lock = threading.Lock()


def library_path() -> str:
    """csrc/libmcr_hip.so next to this file; MCR_HIP_LIBRARY names another build of the same ABI (A/B measurements of
    compile-time variants, tools/)."""
    return os.environ.get("MCR_HIP_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", _LIB_NAME)


def _declare(lib: C.CDLL) -> None:
    P = C.POINTER
    lib.mcr_abi_version.restype = C.c_int
    lib.mcr_abi_version.argtypes = []
    lib.mcr_device_count.restype = C.c_int
    lib.mcr_device_count.argtypes = []
    lib.mcr_last_error.restype = C.c_char_p
    lib.mcr_last_error.argtypes = []
    lib.mcr_query_sizes.restype = C.c_int
    lib.mcr_query_sizes.argtypes = [P(McrParams), C.c_int32, P(McrSizes)]
    lib.mcr_stream_start_month_index.restype = C.c_int32
    lib.mcr_stream_start_month_index.argtypes = [C.c_double, C.c_int32, C.c_double]
    lib.mcr_run_batch.restype = C.c_int
    lib.mcr_run_batch.argtypes = [
        P(McrParams), C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_int32,
        C.c_void_p, P(McrOutputs), C.c_int, C.c_void_p,
    ]
    lib.mcr_run_batch_host.restype = C.c_int
    lib.mcr_run_batch_host.argtypes = [
        P(McrParams), C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_int32,
        C.c_void_p, P(McrOutputs), C.c_int,
    ]
    lib.mcr_draw_shocks_host.restype = C.c_int
    lib.mcr_draw_shocks_host.argtypes = [
        C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_int32, C.c_double,
        C.c_void_p, C.c_int,
    ]
    lib.mcr_run_batch_rng.restype = C.c_int
    lib.mcr_run_batch_rng.argtypes = [
        P(McrParams), P(McrRng), C.c_uint32, C.c_uint64, C.c_uint64, C.c_int32,
        C.c_void_p, P(McrOutputs), C.c_int, C.c_void_p,
    ]
    lib.mcr_probe_months_rng.restype = C.c_int
    lib.mcr_probe_months_rng.argtypes = [
        P(McrParams), P(McrRng), C.c_uint32, C.c_uint64, C.c_uint64, P(C.c_int32), C.c_int32,
        C.c_void_p, C.c_int, C.c_void_p,
    ]
    lib.mcr_run_batch_host_rng.restype = C.c_int
    lib.mcr_run_batch_host_rng.argtypes = [
        P(McrParams), P(McrRng), C.c_uint32, C.c_uint64, C.c_uint64, C.c_int32,
        C.c_void_p, P(McrOutputs), C.c_int,
    ]
    lib.mcr_run_batch_multi_host_rng.restype = C.c_int
    lib.mcr_run_batch_multi_host_rng.argtypes = [
        P(McrParams), P(McrRng), C.c_uint32, C.c_uint64, C.c_uint64, C.c_int32,
        C.c_void_p, P(McrOutputs), P(C.c_int32), C.c_int32,
    ]
    lib.mcr_validate_params.restype = C.c_int
    lib.mcr_validate_params.argtypes = [P(McrParams)]
    lib.mcr_release_cached.restype = C.c_int
    lib.mcr_release_cached.argtypes = [C.c_int]
    lib.mcr_sample_columns.restype = C.c_int
    lib.mcr_sample_columns.argtypes = [C.c_uint32, C.c_uint64, C.c_int32, C.c_void_p]
    lib.mcr_draw_shocks_host_rng.restype = C.c_int
    lib.mcr_draw_shocks_host_rng.argtypes = [
        P(McrRng), C.c_uint32, C.c_uint64, C.c_uint64, C.c_int32, C.c_double, C.c_void_p, C.c_int,
    ]
    lib.mcr_eval_helper_host.restype = C.c_int
    lib.mcr_eval_helper_host.argtypes = [
        C.c_int, P(McrParams), C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
    ]
    lib.mcr_row_quantiles_scratch_bytes.restype = C.c_int64
    lib.mcr_row_quantiles_scratch_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int64]
    lib.mcr_row_quantiles.restype = C.c_int
    lib.mcr_row_quantiles.argtypes = [
        C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_void_p, C.c_int32,
        C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
    ]
    lib.mcr_row_quantiles_sharded.restype = C.c_int
    lib.mcr_row_quantiles_sharded.argtypes = [
        C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
        C.c_int32, C.c_int32, REDUCE_FN, C.c_void_p, C.c_int, C.c_void_p,
    ]
    lib.mcr_row_quantiles_reduce_block.restype = C.c_int64
    lib.mcr_row_quantiles_reduce_block.argtypes = [C.c_int32, P(C.c_int64)]
    lib.mcr_row_quantiles_begin.restype = C.c_int
    lib.mcr_row_quantiles_begin.argtypes = [C.c_void_p, C.c_int32, C.c_int, C.c_void_p]
    lib.mcr_row_quantiles_hist.restype = C.c_int
    lib.mcr_row_quantiles_hist.argtypes = [
        C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int, C.c_void_p,
    ]
    lib.mcr_row_quantiles_scan.restype = C.c_int
    lib.mcr_row_quantiles_scan.argtypes = [
        C.c_int32, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
    ]
    lib.mcr_minmax_success.restype = C.c_int
    lib.mcr_minmax_success.argtypes = [
        C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p,
    ]
    lib.mcr_histogram_success.restype = C.c_int
    lib.mcr_histogram_success.argtypes = [
        C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p,
        C.c_int, C.c_void_p,
    ]
    lib.mcr_row_quantiles_last_fallback_rows.restype = C.c_int
    lib.mcr_row_quantiles_last_fallback_rows.argtypes = []
    lib.mcr_summary_stat_rows.restype = C.c_int
    lib.mcr_summary_stat_rows.argtypes = [
        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p,
    ]


def load_library() -> C.CDLL:
    """Load (once) and return the HIP engine.  Raises RuntimeError if it is not built."""
    global _lib
    with _lib_lock:
        if _lib is None:
            path = library_path()
            if not os.path.exists(path):
                raise RuntimeError(
                    f"HIP engine not built: {path} is missing. Run "
                    "`python -c 'import __graft_entry__ as g; g.build()'` "
                    "(there is no CPU fallback)."
                )
            # One HIP runtime per process: PyTorch (our device-memory / stream / RCCL plumbing) ships
            # its own libamdhip64 and looks it up by a different file name than its SONAME, so if
            # this library pulled in the system runtime first, torch would load a second copy and
            # find "No HIP GPUs".  Importing torch first makes both bind to the same runtime.
            try:
                import torch  # noqa: F401
            except Exception:  # torch-less callers (plain ctypes integration) use the system runtime
                pass
            try:
                lib = C.CDLL(path)
            except OSError as exc:
                raise RuntimeError(f"cannot load HIP engine {path}: {exc}") from exc
            _declare(lib)
            if lib.mcr_abi_version() != MCR_ABI_VERSION:
                raise RuntimeError(
                    f"{path}: ABI version {lib.mcr_abi_version()} != {MCR_ABI_VERSION}; rebuild"
                )
            _lib = lib
        return _lib


def last_error() -> str:
    msg = load_library().mcr_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed (code {rc}): {last_error()}")


def device_count() -> int:
    return int(load_library().mcr_device_count())


def require_device() -> None:
    if device_count() <= 0:
        raise RuntimeError(
            "no usable HIP device: the Monte Carlo path engine runs only on a gfx950 GPU "
            "(there is no CPU fallback)"
        )


def helper_arity(which: int):
    return _HELPER_ARITY[which]
