"""Device-side aggregation (csrc/mcr_aggregate.hip) with pandas/NumPy semantics.

Replaces the pandas block of the reference's run_monte_carlo_simulations
(backend/simulation.py:1045-1118): quantile bands per time point, WR observation counts,
and the histogram of successful final balances (backend/plotting.py:53-59).
torch tensors are device-memory handles only; all arithmetic is in the HIP kernels.
"""

from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _native as N

#: percentile sets of the reference (simulation.py:1045 and :1108-1110)
TRAJECTORY_QUANTILES = (0.05, 0.10, 0.25, 0.50, 0.75, 0.90, 0.95)
WR_QUANTILES = (0.05, 0.25, 0.50, 0.75, 0.95)


#: columns of TRAJECTORY_QUANTILES that make up WR_QUANTILES
_WR_COLS = [TRAJECTORY_QUANTILES.index(q) for q in WR_QUANTILES]


def band_quantiles(batch, n: int, reduce_counts=None, n_total: Optional[int] = None):
    """All quantile bands of a full-output ``DeviceBatch`` in ONE radix-select call over its
    [2T+ry, stride] slab: ``(trajectory_q[T,7], real_trajectory_q[T,7], wr_q[ry,5], wr_counts[ry])``.
    The WR percentile set is a subset of the trajectory set, so the WR rows simply keep 5 of the 7 columns."""
    T = batch.sizes.trajectory_len
    q, counts = row_quantiles(batch.slab, n, TRAJECTORY_QUANTILES, reduce_counts=reduce_counts, n_total=n_total, scratch_owner=batch)
    return q[:T], q[T:2 * T], np.ascontiguousarray(q[2 * T:][:, _WR_COLS]), counts[2 * T:]


def _pandas_q(qs: Sequence[float]) -> np.ndarray:
    """pandas hands ``qs * 100`` to np.percentile, which divides by 100 again
    (pandas/core/array_algos/quantile.py, numpy percentile): reproduce that round trip."""
    return np.true_divide(np.asarray(qs, dtype=np.float64) * 100.0, 100)


def row_quantiles(rows, n: int, qs: Sequence[float], device: int = 0, reduce_counts=None,
                  n_total: Optional[int] = None, scratch_owner=None) -> Tuple[np.ndarray, np.ndarray]:
    """Quantiles of each row of a device tensor ``rows[n_rows, stride]`` over its first ``n``
    entries, NaNs skipped — ``DataFrame(rows.T).quantile(qs, axis=0)`` semantics.

    Multi-GPU: when every rank holds ``n`` of the ``n_total`` entries of each row, pass
    ``reduce_counts`` (a callable that sums an int32 tensor across ranks in place, e.g.
    ``distributed.all_reduce_sum_``): the digit histograms are summed after every pass and every
    rank returns the exact GLOBAL quantiles; the rows themselves never leave their GPU.

    Returns ``(q[n_rows, len(qs)], non_nan_counts[n_rows])`` as numpy arrays.
    """
    import torch

    if reduce_counts is not None:
        return _row_quantiles_sharded(rows, int(n), qs, reduce_counts, int(n_total if n_total is not None else n), scratch_owner)

    assert rows.is_cuda and rows.dtype == torch.float64 and rows.dim() == 2 and rows.stride(1) == 1
    lib = N.load_library()
    n_rows = int(rows.shape[0])
    q = _pandas_q(qs)
    dev = rows.device
    # quantiles [n_rows, n_q] (float64) and counts [n_rows] (uint64) share ONE buffer in PINNED HOST memory, which the
    # device writes directly (a few KB over the host link): the bracketed route ends with a stream synchronisation of its
    # own (it reads one word back), after which the results are already on the host — no separate download and second
    # synchronisation.  The short-row radix route is asynchronous, hence the synchronise below (a no-op after the other).
    res = _pinned_result(n_rows * (len(q) + 1))
    out, counts = res[:n_rows * len(q)], res[n_rows * len(q):]
    nbytes = int(lib.mcr_row_quantiles_scratch_bytes(n_rows, len(q), int(n)))
    if nbytes <= 0:
        raise ValueError("unsupported number of rows / quantiles")
    scratch = _scratch(scratch_owner, nbytes, dev)
    cur = torch.cuda.current_stream(dev)
    rc = lib.mcr_row_quantiles(
        rows.data_ptr(), int(rows.stride(0)), n_rows, int(n), q.ctypes.data, len(q),
        out.data_ptr(), counts.data_ptr(), scratch.data_ptr(), dev.index or 0, C.c_void_p(cur.cuda_stream),
    )
    N.check(rc, "mcr_row_quantiles")
    cur.synchronize()
    host = res.numpy().copy()
    return host[:n_rows * len(q)].reshape(n_rows, len(q)), host[n_rows * len(q):].view(np.uint64).astype(np.int64)


_pinned = threading.local()


def _pinned_result(n_doubles: int):
    """A pinned (page-locked, device-visible) float64 host buffer of at least `n_doubles`, cached per calling thread.
    (torch pins with hipHostMalloc's default flags: coherent host memory, so what the kernels wrote is visible to the host
    once the stream has been synchronised.)"""
    import torch

    buf = getattr(_pinned, "buf", None)
    if buf is None or buf.numel() < n_doubles:
        buf = torch.empty(max(n_doubles, 4096), dtype=torch.float64, pin_memory=True)
        _pinned.buf = buf
    return buf[:n_doubles]


def _scratch(owner, nbytes: int, dev):
    """Scratch buffer of a selection call.  A caller that selects repeatedly over the same batch (the bands of a
    DeviceBatch, then the summary statistics) passes the batch as `owner`: the buffer then lives and dies with it."""
    import torch

    if owner is not None:
        buf = getattr(owner, "_rq_scratch", None)
        if buf is not None and buf.numel() >= nbytes and buf.device == dev:
            return buf
    buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    if owner is not None:
        owner._rq_scratch = buf
    return buf


def last_fallback_rows() -> int:
    """Diagnostics for the calling thread's last :func:`row_quantiles`: -1 = the radix route was taken, else how
    many rows of the bracketed route needed the full radix passes."""
    return int(N.load_library().mcr_row_quantiles_last_fallback_rows())


#: rows sharded over ranks take the bracketed single-pass route from this many entries per row IN TOTAL; every shard must
#: hold at least _SHARD_MIN_LOCAL entries (the first sample is rank 0's).  On one GPU the bracketed route wins from 2^14
#: entries (csrc/mcr_aggregate.hip: rq_bracket_min_n); sharded, both routes add their collectives — ~10 small all-reduces
#: here against 8 of the digit histograms there — so the same argument holds per rank once every shard is a few times the
#: first sample: 2^18 in total with shards of 2^14 and more (round 2 waited for 2^21 / 2^16).
_SHARDED_BRACKET_MIN_TOTAL = 1 << 18
_SHARD_MIN_LOCAL = 1 << 14


def _row_quantiles_sharded(rows, n_local: int, qs, reduce_counts, n_total: int, scratch_owner=None):
    import torch
    import torch.distributed as dist

    assert rows.is_cuda and rows.dtype == torch.float64 and rows.dim() == 2 and rows.stride(1) == 1
    lib = N.load_library()
    n_rows = int(rows.shape[0])
    q = _pandas_q(qs)
    dev = rows.device
    di = dev.index or 0
    res = torch.empty(n_rows * (len(q) + 1), dtype=torch.float64, device=dev)
    out, counts = res[:n_rows * len(q)], res[n_rows * len(q):]
    nbytes = int(lib.mcr_row_quantiles_scratch_bytes(n_rows, len(q), n_local))
    if nbytes <= 0:
        raise ValueError("unsupported number of rows / quantiles")
    scratch = _scratch(scratch_owner, nbytes, dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def unpack():
        host = res.cpu().numpy()
        return host[:n_rows * len(q)].reshape(n_rows, len(q)), host[n_rows * len(q):].view(np.uint64).astype(np.int64)

    # Route: the same on every rank (it depends on n_total and on the SMALLEST shard, agreed through one tiny reduction)
    grouped = dist.is_available() and dist.is_initialized()
    rank, world = (dist.get_rank(), dist.get_world_size()) if grouped else (0, 1)
    floor = int(os.environ.get("MCR_RQ_SHARDED_BRACKET_MIN_N", _SHARDED_BRACKET_MIN_TOTAL))
    use_bracket = grouped and n_total >= floor and 2 * len(q) < 32 and world <= 64
    if use_bracket:
        # (never below the library's own minimum — 4096 entries on rank 0, whose first sample seeds the brackets — whatever
        #  the override says: the library checks that on rank 0 only, and a route must be taken by every rank or by none)
        short = torch.tensor([1 if n_local < max(4096, min(_SHARD_MIN_LOCAL, floor)) else 0], dtype=torch.int32, device=dev)
        reduce_counts(short)
        use_bracket = int(short.item()) == 0
    if use_bracket:
        base = scratch.data_ptr()
        errors = []

        def reduce(_ctx, ptr, count, dtype):
            try:
                width, tdt = (4, torch.int32) if dtype == N.MCR_DT_I32 else (8, torch.int64)
                off = int(ptr) - base
                reduce_counts(scratch[off:off + int(count) * width].view(tdt))
                return 0
            except Exception as exc:  # never let an exception cross the C boundary
                errors.append(exc)
                return 1

        cb = N.REDUCE_FN(reduce)
        rc = lib.mcr_row_quantiles_sharded(
            rows.data_ptr(), int(rows.stride(0)), n_rows, n_local, n_total, q.ctypes.data, len(q), out.data_ptr(),
            counts.data_ptr(), scratch.data_ptr(), rank, world, cb, None, di, stream)
        if errors:
            raise errors[0]
        N.check(rc, "mcr_row_quantiles_sharded")
        return unpack()

    n_words = C.c_int64()
    off = int(lib.mcr_row_quantiles_reduce_block(n_rows, C.byref(n_words)))
    block = scratch[off:off + 4 * n_words.value].view(torch.int32)  # the dense counter block to sum across ranks
    N.check(lib.mcr_row_quantiles_begin(scratch.data_ptr(), n_rows, di, stream), "mcr_row_quantiles_begin")
    for p in range(8):
        N.check(lib.mcr_row_quantiles_hist(rows.data_ptr(), int(rows.stride(0)), n_rows, n_local, len(q), p,
                                           scratch.data_ptr(), di, stream), "mcr_row_quantiles_hist")
        reduce_counts(block)
        N.check(lib.mcr_row_quantiles_scan(n_rows, n_total, q.ctypes.data, len(q), p, out.data_ptr(), counts.data_ptr(),
                                           scratch.data_ptr(), di, stream), "mcr_row_quantiles_scan")
    return unpack()


def success_histogram(values, success, n_bins: int = 100, value_range: Optional[Tuple[float, float]] = None,
                      reduce_range=None, reduce_bins=None):
    """``np.histogram(values[success], bins=n_bins)`` on the device.

    ``reduce_range(minmax_tensor)`` / ``reduce_bins(bins_tensor)`` are optional hooks where a
    multi-GPU caller all-reduces (min/max, then sum) in place.  Returns ``(counts, edges)``.
    """
    import torch

    assert values.is_cuda and success.is_cuda and values.dtype == torch.float64 and success.dtype == torch.uint8
    lib = N.load_library()
    dev = values.device
    n = int(values.shape[0])
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    # range and bins share ONE device buffer (2 doubles | n_bins int64): one download and one synchronisation at the end
    both = torch.zeros(2 + n_bins, dtype=torch.int64, device=dev)
    minmax, bins = both[:2].view(torch.float64), both[2:]
    if value_range is None:
        N.check(lib.mcr_minmax_success(values.data_ptr(), success.data_ptr(), n, minmax.data_ptr(), dev.index or 0, stream),
                "mcr_minmax_success")
        if reduce_range is not None:
            reduce_range(minmax)
    else:
        minmax.copy_(torch.tensor(value_range, dtype=torch.float64))
    N.check(lib.mcr_histogram_success(values.data_ptr(), success.data_ptr(), n, minmax.data_ptr(), n_bins,
                                      bins.data_ptr(), dev.index or 0, stream), "mcr_histogram_success")
    if reduce_bins is not None:
        reduce_bins(bins)
    host = both.cpu()
    lo, hi = (float(x) for x in host[:2].view(torch.float64).tolist())
    if not np.isfinite(lo) or not np.isfinite(hi):  # empty cohort: numpy's default range
        lo, hi = 0.0, 1.0
    if lo == hi:
        lo, hi = lo - 0.5, hi + 0.5
    return host[2:].numpy().copy(), np.linspace(lo, hi, n_bins + 1)


def summary_stat_rows(batch, n: int):
    """Device rows ``[4, stride]`` for the response document's summary statistics (start balance | final
    balance | final balance of successful paths | first-year real withdrawal rate in %, NaN = outside the
    cohort), ready for :func:`row_quantiles` — see ``mcr_summary_stat_rows`` (include/mcr.h)."""
    import torch

    s = batch.summary
    stride = (int(n) + 63) // 64 * 64
    rows = torch.empty((N.MCR_N_STAT_ROWS, stride), dtype=torch.float64, device=batch.success.device)
    stream = C.c_void_p(torch.cuda.current_stream(rows.device).cuda_stream)
    N.check(N.load_library().mcr_summary_stat_rows(
        s["start_balance"].data_ptr(), s["final_balance"].data_ptr(), s["first_year_real_gross_withdrawal"].data_ptr(),
        batch.success.data_ptr(), int(n), rows.data_ptr(), stride, rows.device.index or 0, stream), "mcr_summary_stat_rows")
    return rows
