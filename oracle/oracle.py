"""ctypes loader for the CPU oracle (oracle/mcr_oracle.c) — TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The product package (monte_carlo_retirement_amd/) never does.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional

import numpy as np

from monte_carlo_retirement_amd._native import (
    MCR_N_COUNTERS,
    McrOutputs,
    McrParams,
    McrSizes,
)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib: Optional[C.CDLL] = None


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc (no FMA contraction)."""
    src = os.path.join(_HERE, "mcr_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "mcr.h")
    stale = (
        force
        or not os.path.exists(_LIB_PATH)
        or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr))
    )
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        P = C.POINTER
        L.orc_arithmetic_to_log_params.restype = C.c_int
        L.orc_arithmetic_to_log_params.argtypes = [C.c_double, C.c_double, P(C.c_double), P(C.c_double)]
        L.orc_stream_start_month_index.restype = C.c_int32
        L.orc_stream_start_month_index.argtypes = [C.c_double, C.c_int32, C.c_double]
        L.orc_trajectory_time_points.restype = C.c_int32
        L.orc_trajectory_time_points.argtypes = [C.c_int32, C.c_int32, P(C.c_double)]
        L.orc_withdraw.restype = None
        L.orc_withdraw.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int, C.c_double] + [P(C.c_double)] * 4
        L.orc_nlv.restype = C.c_double
        L.orc_nlv.argtypes = [C.c_double, C.c_double, C.c_int, C.c_double]
        L.orc_rebalance.restype = None
        L.orc_rebalance.argtypes = [P(McrParams)] + [P(C.c_double)] * 4
        L.orc_annual_tax.restype = C.c_int
        L.orc_annual_tax.argtypes = [P(McrParams)] + [P(C.c_double)] * 4 + [C.c_double, C.c_double]
        L.orc_monthly_gross.restype = C.c_double
        L.orc_monthly_gross.argtypes = [C.c_double, C.c_double, C.c_double]
        L.orc_philox4x32_10.restype = None
        L.orc_philox4x32_10.argtypes = [P(C.c_uint32), P(C.c_uint32), P(C.c_uint32)]
        L.orc_draw_shocks.restype = None
        L.orc_draw_shocks.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_int32, C.c_double, C.c_void_p]
        L.orc_query_sizes.restype = C.c_int
        L.orc_query_sizes.argtypes = [P(McrParams), C.c_int32, P(McrSizes)]
        L.orc_run_batch.restype = C.c_int
        L.orc_run_batch.argtypes = [
            P(McrParams), C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_int32,
            C.c_void_p, P(McrOutputs),
        ]
        _lib = L
    return _lib


# ---- thin pythonic wrappers -----------------------------------------------------------

def log_params(mean: float, vol: float):
    mu, sg = C.c_double(), C.c_double()
    if lib().orc_arithmetic_to_log_params(mean, vol, C.byref(mu), C.byref(sg)) != 0:
        raise ValueError("invalid mean/vol")
    return mu.value, sg.value


def stream_start_month_index(current_age: float, wm: int, start_at_age: float) -> int:
    return int(lib().orc_stream_start_month_index(current_age, wm, start_at_age))


def trajectory_time_points(wm: int, ry: int):
    buf = (C.c_double * (2 + wm // 12 + 1 + ry))()
    n = lib().orc_trajectory_time_points(wm, ry, buf)
    return [buf[i] for i in range(n)]


def withdraw(bal, cb, target, use_real, rate):
    o = [C.c_double() for _ in range(4)]
    lib().orc_withdraw(bal, cb, target, int(bool(use_real)), rate, *[C.byref(x) for x in o])
    return tuple(x.value for x in o)


def nlv(bal, cb, use_real, rate) -> float:
    return float(lib().orc_nlv(bal, cb, int(bool(use_real)), rate))


def rebalance(params: McrParams, b1, cb1, b2, cb2):
    v = [C.c_double(x) for x in (b1, cb1, b2, cb2)]
    lib().orc_rebalance(C.byref(params), *[C.byref(x) for x in v])
    return tuple(x.value for x in v)


def annual_tax(params: McrParams, b1, cb1, b2, cb2, g1, g2):
    v = [C.c_double(x) for x in (b1, cb1, b2, cb2)]
    failed = lib().orc_annual_tax(C.byref(params), *[C.byref(x) for x in v], g1, g2)
    return tuple(x.value for x in v) + (bool(failed),)


def monthly_gross(mu_log, sigma_log, z) -> float:
    return float(lib().orc_monthly_gross(mu_log, sigma_log, z))


def philox4x32_10(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return [int(x) for x in o]


def draw_shocks(seed: int, stream_id: int, path: int, n_months: int, rho: float) -> np.ndarray:
    out = np.empty((n_months, 3), dtype=np.float64)
    lib().orc_draw_shocks(seed, stream_id, path, n_months, rho, out.ctypes.data)
    return out


def query_sizes(params: McrParams, wm: int) -> McrSizes:
    s = McrSizes()
    rc = lib().orc_query_sizes(C.byref(params), wm, C.byref(s))
    if rc != 0:
        raise ValueError("invalid params / working_months")
    return s


def run_batch(
    params: McrParams,
    seed: int,
    stream_id: int,
    path_begin: int,
    n_paths: int,
    wm: int,
    injected_shocks: Optional[np.ndarray] = None,
    want_summary: bool = True,
    want_trajectories: bool = True,
) -> Dict[str, np.ndarray]:
    """Run the oracle over a path range; returns numpy arrays keyed like mcr_outputs."""
    sz = query_sizes(params, wm)
    n = int(n_paths)
    res: Dict[str, np.ndarray] = {}
    o = McrOutputs()
    o.path_stride = n
    if want_summary:
        for k in (
            "start_balance", "final_balance", "years_to_ruin", "first_year_gross_withdrawal",
            "first_year_real_gross_withdrawal", "inflation_at_retirement",
        ):
            res[k] = np.empty(n, dtype=np.float64)
            setattr(o, k, res[k].ctypes.data)
        res["success"] = np.empty(n, dtype=np.uint8)
        o.success = res["success"].ctypes.data
    if want_trajectories:
        res["trajectory"] = np.empty((sz.trajectory_len, n), dtype=np.float64)
        res["real_trajectory"] = np.empty((sz.trajectory_len, n), dtype=np.float64)
        res["withdrawal_rate_trajectory"] = np.empty((sz.retirement_years, n), dtype=np.float64)
        o.trajectory = res["trajectory"].ctypes.data
        o.real_trajectory = res["real_trajectory"].ctypes.data
        o.withdrawal_rate_trajectory = res["withdrawal_rate_trajectory"].ctypes.data
    res["counters"] = np.zeros(MCR_N_COUNTERS, dtype=np.uint64)
    res["wr_obs_counts"] = np.zeros(sz.retirement_years, dtype=np.uint64)
    res["ruin_year_bins"] = np.zeros(sz.ruin_bins, dtype=np.uint64)
    o.counters = res["counters"].ctypes.data
    o.wr_obs_counts = res["wr_obs_counts"].ctypes.data
    o.ruin_year_bins = res["ruin_year_bins"].ctypes.data
    inj = None
    if injected_shocks is not None:
        inj_arr = np.ascontiguousarray(injected_shocks, dtype=np.float64)
        assert inj_arr.shape == (n, sz.shock_rows, 3), (inj_arr.shape, (n, sz.shock_rows, 3))
        inj = inj_arr.ctypes.data
    rc = lib().orc_run_batch(C.byref(params), seed, stream_id, path_begin, n, wm, inj, C.byref(o))
    if rc != 0:
        raise RuntimeError(f"orc_run_batch failed: {rc}")
    return res
