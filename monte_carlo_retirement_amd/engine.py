"""Python face of the C ABI (include/mcr.h): numpy in / numpy out, HIP underneath.

Everything here runs the hand-written gfx950 kernels in csrc/ — there is no CPU path.
Two families:

* ``*_host`` helpers move small batches through host (numpy) buffers; the library
  allocates device scratch, launches, copies back (``mcr_run_batch_host``).
* :class:`DeviceBatch` keeps the outputs of a large batch resident in HBM (torch tensors
  are used purely as device-memory handles / stream plumbing) for the aggregation kernels.
"""

from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence

import numpy as np

from . import _native as N
from ._native import McrOutputs, McrParams, McrRng, McrSizes

SUMMARY_FIELDS = (
    "start_balance",
    "final_balance",
    "years_to_ruin",
    "first_year_gross_withdrawal",
    "first_year_real_gross_withdrawal",
    "inflation_at_retirement",
)


def query_sizes(params: McrParams, working_months: int) -> McrSizes:
    """Shapes for (params, working_months); ``ValueError`` on invalid input."""
    sz = McrSizes()
    rc = N.load_library().mcr_query_sizes(C.byref(params), int(working_months), C.byref(sz))
    if rc != 0:
        raise ValueError(N.last_error() or "invalid params / working_months")
    return sz


def stream_start_month_index(current_age: float, working_months: int, start_at_age: float) -> int:
    return int(N.load_library().mcr_stream_start_month_index(current_age, working_months, start_at_age))


def _as_rng(seed) -> McrRng:
    """`seed` may be an int (Philox key) or a ready McrRng descriptor (copied: callers' objects are never mutated)."""
    if isinstance(seed, McrRng):
        dup = McrRng()
        C.memmove(C.byref(dup), C.byref(seed), C.sizeof(McrRng))
        return dup
    return N.philox_rng(int(seed))


def hist_edge_array(edges) -> np.ndarray:
    """Bin edges of the in-kernel final-balance histogram as the contiguous float64 array the ABI wants;
    ``ValueError`` unless they are finite, ascending and 2..MCR_MAX_HIST_BINS+1 long (np.histogram's own rule)."""
    e = np.ascontiguousarray(np.asarray(edges, dtype=np.float64).reshape(-1))
    if not (2 <= e.shape[0] <= N.MCR_MAX_HIST_BINS + 1):
        raise ValueError(f"hist_edges: need 2..{N.MCR_MAX_HIST_BINS + 1} edges, got {e.shape[0]}")
    if not np.all(np.isfinite(e)) or np.any(e[1:] < e[:-1]):
        raise ValueError("hist_edges must be finite and increase monotonically")
    return e


def uniform_hist_edges(value_range, n_bins: int) -> np.ndarray:
    """The edges ``np.histogram(x, bins=n_bins, range=value_range)`` bins on (numpy/lib/_histograms_impl.py:
    ``_get_outer_edges`` widens a degenerate range by 0.5 either side, then ``np.linspace``)."""
    lo, hi = (float(v) for v in value_range)
    if not (np.isfinite(lo) and np.isfinite(hi)) or lo > hi:
        raise ValueError("value_range must be finite with lo <= hi")
    if lo == hi:
        lo, hi = lo - 0.5, hi + 0.5
    return np.linspace(lo, hi, int(n_bins) + 1, endpoint=True, dtype=np.float64)


def run_batch_host(
    params: McrParams,
    seed,
    stream_id: int,
    path_begin: int,
    n_paths: int,
    working_months: int,
    injected_shocks: Optional[np.ndarray] = None,
    want_summary: bool = True,
    want_trajectories: bool = True,
    want_bins: bool = True,
    device: int = 0,
    path_seeds: Optional[np.ndarray] = None,
    devices: Optional[Sequence[int]] = None,
    hist_edges: Optional[np.ndarray] = None,
) -> Dict[str, np.ndarray]:
    """Simulate paths [path_begin, path_begin+n_paths) on the GPU; numpy arrays out.

    ``device``: HIP ordinal, or ``_native.MCR_DEVICE_ALL`` = shard over every visible device.  ``devices``: an
    explicit device list instead (``mcr_run_batch_multi_host_rng``: one host thread per entry, contiguous shards
    of the path range, counters summed on the host) — same numbers whatever the list.

    ``seed``: int (Philox key) or an ``McrRng`` (e.g. ``_native.numpy_rng(main_seed, child_offset)``);
    ``path_seeds``: explicit uint32 seed per path for the NumPy stream.

    Keys follow ``mcr_outputs``: the six float summary fields + ``success`` (uint8),
    ``trajectory``/``real_trajectory`` ``[T, n]``, ``withdrawal_rate_trajectory`` ``[ry, n]``,
    ``counters`` ``[2]``, ``wr_obs_counts`` ``[ry]``, ``ruin_year_bins`` ``[ry+2]``.

    ``hist_edges``: ascending bin edges ``[n_bins + 1]`` -> ``hist_bins`` ``[n_bins]`` =
    ``np.histogram(final_balance[success], bins=hist_edges)[0]``, binned inside the path kernel
    (``mcr_outputs.hist_bins``; no per-path output is needed for it).
    """
    lib = N.load_library()
    N.require_device()
    sz = query_sizes(params, working_months)
    n = int(n_paths)
    res: Dict[str, np.ndarray] = {}
    o = McrOutputs()
    o.path_stride = n
    if want_summary:
        for k in SUMMARY_FIELDS:
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
    res["counters"] = np.zeros(N.MCR_N_COUNTERS, dtype=np.uint64)
    o.counters = res["counters"].ctypes.data
    if want_bins:
        res["wr_obs_counts"] = np.zeros(sz.retirement_years, dtype=np.uint64)
        res["ruin_year_bins"] = np.zeros(sz.ruin_bins, dtype=np.uint64)
        o.wr_obs_counts = res["wr_obs_counts"].ctypes.data
        o.ruin_year_bins = res["ruin_year_bins"].ctypes.data
    edges_arr = None
    if hist_edges is not None:
        edges_arr = hist_edge_array(hist_edges)
        res["hist_bins"] = np.zeros(edges_arr.shape[0] - 1, dtype=np.uint64)
        o.hist_edges = edges_arr.ctypes.data
        o.hist_bins = res["hist_bins"].ctypes.data
        o.hist_n_bins = edges_arr.shape[0] - 1
    inj = None
    inj_arr = None
    if injected_shocks is not None:
        inj_arr = np.ascontiguousarray(injected_shocks, dtype=np.float64)
        if inj_arr.shape != (n, sz.shock_rows, 3):
            raise ValueError(f"injected_shocks shape {inj_arr.shape} != {(n, sz.shock_rows, 3)}")
        inj = inj_arr.ctypes.data
    rng = _as_rng(seed)
    seeds_arr = None
    if path_seeds is not None:
        seeds_arr = np.ascontiguousarray(path_seeds, dtype=np.uint32)
        if seeds_arr.shape != (n,):
            raise ValueError("path_seeds must have one uint32 per path")
        rng.path_seeds = seeds_arr.ctypes.data
    if devices is not None:
        devs = (C.c_int32 * len(devices))(*[int(d) for d in devices])
        rc = lib.mcr_run_batch_multi_host_rng(
            C.byref(params), C.byref(rng), int(stream_id), int(path_begin), n, int(working_months),
            inj, C.byref(o), devs, len(devices),
        )
        N.check(rc, "mcr_run_batch_multi_host_rng")
        return res
    rc = lib.mcr_run_batch_host_rng(
        C.byref(params), C.byref(rng), int(stream_id), int(path_begin), n, int(working_months),
        inj, C.byref(o), int(device),
    )
    N.check(rc, "mcr_run_batch_host_rng")
    return res


def draw_shocks_host(
    seed, stream_id: int, path_begin: int, n_paths: int, n_months: int, rho: float, device: int = 0,
    path_seeds: Optional[np.ndarray] = None,
) -> np.ndarray:
    """Engine shock rows ``[n_paths, n_months, 3]`` (equity, inflation, premium) from the GPU."""
    N.require_device()
    out = np.empty((int(n_paths), int(n_months), 3), dtype=np.float64)
    rng = _as_rng(seed)
    seeds_arr = None
    if path_seeds is not None:
        seeds_arr = np.ascontiguousarray(path_seeds, dtype=np.uint32)
        rng.path_seeds = seeds_arr.ctypes.data
    rc = N.load_library().mcr_draw_shocks_host_rng(
        C.byref(rng), int(stream_id), int(path_begin), int(n_paths), int(n_months), float(rho),
        out.ctypes.data, int(device),
    )
    N.check(rc, "mcr_draw_shocks_host_rng")
    return out


def eval_helper_host(which: int, params: Optional[McrParams], rows, device: int = 0) -> np.ndarray:
    """Evaluate one of the scalar device functions (MCR_HELPER_*) on the GPU for each input row."""
    N.require_device()
    n_in, n_out = N.helper_arity(which)
    a = np.ascontiguousarray(np.asarray(rows, dtype=np.float64).reshape(-1, n_in))
    out = np.empty((a.shape[0], n_out), dtype=np.float64)
    rc = N.load_library().mcr_eval_helper_host(
        int(which), C.byref(params) if params is not None else None, a.ctypes.data,
        out.ctypes.data, a.shape[0], int(device),
    )
    N.check(rc, "mcr_eval_helper_host")
    return out


# ---------------------------------------------------------------------------------------------
# Device-resident batches (torch = device memory + stream plumbing only)
# ---------------------------------------------------------------------------------------------
class DeviceBatch:
    """Outputs of one (possibly multi-launch) batch kept resident in HBM.

    ``want``: ``"count"`` (counters/bins only), ``"summary"`` (+ per-path fields) or
    ``"full"`` (+ time-major trajectories).  Tensors are allocated once and reused; launches
    go to torch's current stream through ``mcr_run_batch`` (device pointers).
    """

    def __init__(self, params: McrParams, working_months: int, n_paths: int, want: str = "count",
                 device: int = 0, hist_edges=None):
        import torch

        if want not in ("count", "summary", "full"):
            raise ValueError(want)
        N.require_device()
        self.torch = torch
        self.params = params
        self.working_months = int(working_months)
        self.n_paths = int(n_paths)
        self.want = want
        self.device = int(device)
        self.sizes = query_sizes(params, working_months)
        dev = torch.device("cuda", self.device)
        self.stride = (self.n_paths + 63) // 64 * 64
        f64, i64 = torch.float64, torch.int64
        # counters | wr_obs_counts | ruin_year_bins live in ONE int64 vector: the path's single exchange step
        # across GPUs is one all-reduce(sum) of it
        ry_ = self.sizes.retirement_years
        self.hist_edges = None if hist_edges is None else hist_edge_array(hist_edges)
        n_hist = 0 if self.hist_edges is None else self.hist_edges.shape[0] - 1
        n_fixed = N.MCR_N_COUNTERS + ry_ + self.sizes.ruin_bins
        self.reduce_vec = torch.zeros(n_fixed + n_hist, dtype=i64, device=dev)
        self.counters = self.reduce_vec[:N.MCR_N_COUNTERS]
        self.wr_obs_counts = self.reduce_vec[N.MCR_N_COUNTERS:N.MCR_N_COUNTERS + ry_]
        self.ruin_year_bins = self.reduce_vec[N.MCR_N_COUNTERS + ry_:n_fixed]
        self.hist_bins = self.reduce_vec[n_fixed:] if n_hist else None
        self._hist_edges_dev = torch.as_tensor(self.hist_edges, device=dev) if n_hist else None
        self.summary = {}
        self.success = None
        self.trajectory = self.real_trajectory = self.withdrawal_rate_trajectory = None
        if want in ("summary", "full"):
            for k in SUMMARY_FIELDS:
                self.summary[k] = torch.empty(self.n_paths, dtype=f64, device=dev)
            self.success = torch.empty(self.n_paths, dtype=torch.uint8, device=dev)
        self.slab = None
        if want == "full":
            # one [2T+ry, stride] slab (nominal | real | withdrawal-rate rows) so the quantile bands of all
            # rows come from a single radix-select call
            T, ry = self.sizes.trajectory_len, self.sizes.retirement_years
            self.slab = torch.empty((2 * T + ry, self.stride), dtype=f64, device=dev)
            self.trajectory = self.slab[:T]
            self.real_trajectory = self.slab[T:2 * T]
            self.withdrawal_rate_trajectory = self.slab[2 * T:]
        o = McrOutputs()
        o.path_stride = self.stride
        o.counters = self.counters.data_ptr()
        o.wr_obs_counts = self.wr_obs_counts.data_ptr()
        o.ruin_year_bins = self.ruin_year_bins.data_ptr()
        if n_hist:
            o.hist_edges = self._hist_edges_dev.data_ptr()
            o.hist_bins = self.hist_bins.data_ptr()
            o.hist_n_bins = n_hist
        for k, t in self.summary.items():
            setattr(o, k, t.data_ptr())
        if self.success is not None:
            o.success = self.success.data_ptr()
        if self.trajectory is not None:
            o.trajectory = self.trajectory.data_ptr()
            o.real_trajectory = self.real_trajectory.data_ptr()
            o.withdrawal_rate_trajectory = self.withdrawal_rate_trajectory.data_ptr()
        self._out = o
        self._lib = N.load_library()

    def zero_counters(self) -> None:
        self.reduce_vec.zero_()

    def release_scratch(self) -> None:
        """Drop the selection scratch the aggregation calls cached on this batch (`aggregation.band_quantiles` /
        `row_quantiles(..., scratch_owner=batch)`: ~1.5 GB at 136 rows x 1e7 paths, kept next to the slab for the batch's
        lifetime so that repeated selections do not re-allocate).  The next selection allocates it again."""
        self._rq_scratch = None

    def launch(self, seed, stream_id: int, path_begin: int, n_paths: Optional[int] = None) -> None:
        """Enqueue one kernel launch over ``n_paths`` (default: the whole batch) on the current stream.
        ``seed``: int (Philox key) or an ``McrRng`` descriptor."""
        n = self.n_paths if n_paths is None else int(n_paths)
        if n > self.n_paths:
            raise ValueError("launch larger than the batch buffers")
        stream = self.torch.cuda.current_stream(self.device).cuda_stream
        rng = _as_rng(seed)
        rc = self._lib.mcr_run_batch_rng(
            C.byref(self.params), C.byref(rng), int(stream_id), int(path_begin), n,
            self.working_months, None, C.byref(self._out), self.device, C.c_void_p(stream),
        )
        N.check(rc, "mcr_run_batch_rng")


def probe_months(params: McrParams, seed, stream_id: int, path_begin: int, n_paths: int, working_months,
                 device: int = 0):
    """Success counters of several candidate working-month counts over the same path range: one
    count-only launch per candidate, run concurrently on the library's side streams (fork/join on
    torch's current stream).  Returns a device int64 tensor ``[len(working_months), 2]`` =
    ``{successes, paths}``; asynchronous (reading it synchronises)."""
    import torch

    N.require_device()
    months = (C.c_int32 * len(working_months))(*[int(m) for m in working_months])
    counts = torch.empty((len(working_months), N.MCR_N_COUNTERS), dtype=torch.int64,
                         device=torch.device("cuda", int(device)))
    if len(working_months) == 0:
        return counts
    rng = _as_rng(seed)
    stream = torch.cuda.current_stream(int(device)).cuda_stream
    rc = N.load_library().mcr_probe_months_rng(
        C.byref(params), C.byref(rng), int(stream_id), int(path_begin), int(n_paths), months,
        len(working_months), counts.data_ptr(), int(device), C.c_void_p(stream),
    )
    N.check(rc, "mcr_probe_months_rng")
    return counts
