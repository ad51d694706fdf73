"""Multi-GPU execution: one process per GPU, independent path ranges, ONE all-reduce.

The reference has no distributed code (its only parallelism is a ``multiprocessing.Pool`` over
independent paths, backend/simulation.py:996-1001).  Here the global path range ``[0, N)`` is
cut into contiguous shards, one per rank; the Philox counter carries the GLOBAL path index, so
the union of the shards is bit-identical to a single-GPU run.  The only exchange step is a
sum all-reduce of one small int64 vector ``[success, paths, wr_obs_counts[ry], ruin_bins[ry+2]]``
(RCCL over xGMI when the backend is "nccl"; latency-bound, < 2 KB) — plus, for the histogram of
final balances, a min/max all-reduce of two doubles before the bins are summed, and for exact
cross-GPU quantile bands the radix select's digit histograms summed once per pass
(``sharded_row_quantiles``).  Trajectories never leave the GPU that produced them; the only O(n)
exchange is the all-gather of the 49 B/path summary when a caller wants the per-path frame
(``RetirementMonteCarloSimulator.run_monte_carlo_simulations`` under a process group).

``torch.distributed`` is plumbing only; all path arithmetic is in the HIP kernels.
"""

from __future__ import annotations

import contextlib
import threading
from dataclasses import dataclass
from typing import Callable, Dict, Optional, Tuple

import numpy as np


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard of the global path range for ``rank``: ``(path_begin, n_paths)``.
    Shards are ``ceil(N / world)`` long (the last ones shorter or empty); they tile ``[0, N)``."""
    if world <= 0 or not (0 <= rank < world) or n_total < 0:
        raise ValueError("bad shard arguments")
    per = -(-n_total // world)
    begin = min(rank * per, n_total)
    return begin, min(per, n_total - begin)


_local = threading.local()


@contextlib.contextmanager
def local_only():
    """Inside this block the calling thread computes as if no process group existed: every layer that shards
    transparently (`run_monte_carlo_simulations`, the search's probes, `compact_result`) runs its single-GPU route.
    For cross-checks — e.g. `bench.py` verifies that the candidate-split search replays the single-GPU search
    probe for probe.  Every rank must enter/leave such a block at the same point of its program."""
    prev = getattr(_local, "off", False)
    _local.off = True
    try:
        yield
    finally:
        _local.off = prev


def is_active() -> bool:
    """True when a torch.distributed process group with more than one rank is initialised (and the calling thread
    is not inside :func:`local_only`)."""
    if getattr(_local, "off", False):
        return False
    try:
        import torch.distributed as dist
    except Exception:  # pragma: no cover
        return False
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def _comm_device():
    import torch
    import torch.distributed as dist

    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


@dataclass
class ReducedCounts:
    success: int
    paths: int
    wr_obs_counts: np.ndarray   # [ry]
    ruin_year_bins: np.ndarray  # [ry + 2]

    @property
    def success_probability_pct(self) -> float:
        return float(np.float64(self.success) / np.float64(self.paths) * 100.0) if self.paths else 0.0


def pack_counts(counters, wr_obs_counts, ruin_year_bins) -> np.ndarray:
    return np.concatenate([np.asarray(counters, dtype=np.int64), np.asarray(wr_obs_counts, dtype=np.int64),
                           np.asarray(ruin_year_bins, dtype=np.int64)])


def unpack_counts(vec: np.ndarray, retirement_years: int) -> ReducedCounts:
    ry = int(retirement_years)
    assert vec.shape[0] == 2 + ry + ry + 2
    return ReducedCounts(int(vec[0]), int(vec[1]), vec[2:2 + ry].copy(), vec[2 + ry:].copy())


def broadcast_int(value: int, src: int = 0) -> int:
    """``value`` of rank ``src`` on every rank (no-op without a process group).  Values up to 2**63 - 1."""
    if not is_active():
        return int(value)
    import torch
    import torch.distributed as dist

    t = torch.tensor([int(value)], dtype=torch.int64, device=_comm_device())
    dist.broadcast(t, src=src)
    return int(t.item())


def all_reduce_sum_(tensor) -> None:
    """In-place sum over ranks (the path's single exchange step)."""
    import torch.distributed as dist

    dist.all_reduce(tensor, op=dist.ReduceOp.SUM)


def probe_candidates(months, n_paths: int, shard_min_paths: int, probe):
    """Success counters of several candidate working-month counts over the path range ``[0, n_paths)`` under a process
    group: ``probe(path_begin, count, months) -> int64 [len(months), 2]`` (``{successes, paths}`` per month; a numpy array or a
    torch tensor on any device) runs the local kernels.  Batches of at least ``shard_min_paths`` are sharded by PATH RANGE
    (every rank counts all candidates on its shard); smaller ones are split by CANDIDATE (rank r takes ``months[r::world]`` over
    the whole range — with fewer candidates than ranks the others contribute zeros).  Either way the per-candidate counter
    block is summed with ONE all-reduce and every rank returns the same numpy array ``[len(months), 2]``.  Without a process
    group: ``probe(0, n_paths, months)``."""
    import torch

    months = [int(m) for m in months]
    n = int(n_paths)

    def as_tensor(x, device):
        t = x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x, dtype=np.int64))
        return t.to(device=device, dtype=torch.int64).reshape(-1, 2)

    if not is_active():
        return as_tensor(probe(0, n, months), torch.device("cpu")).numpy()
    import torch.distributed as dist

    rank, world = dist.get_rank(), dist.get_world_size()
    counts = torch.zeros((len(months), 2), dtype=torch.int64, device=_comm_device())
    if n >= int(shard_min_paths):
        begin, count = shard_range(n, rank, world)
        if count > 0:
            counts.copy_(as_tensor(probe(begin, count, months), counts.device))
    else:
        mine = list(range(rank, len(months), world))
        if mine:
            counts[mine] = as_tensor(probe(0, n, [months[i] for i in mine]), counts.device)
    all_reduce_sum_(counts)
    return counts.cpu().numpy()


def all_reduce_minmax_(minmax) -> None:
    """In-place: ``minmax[0]`` = min over ranks, ``minmax[1]`` = max over ranks (histogram range)."""
    import torch.distributed as dist

    lo, hi = minmax[0:1].clone(), minmax[1:2].clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    minmax[0:1].copy_(lo)
    minmax[1:2].copy_(hi)


LocalRunner = Callable[[int, int], Dict[str, np.ndarray]]
"""``runner(path_begin, n_paths) -> {"counters", "wr_obs_counts", "ruin_year_bins"}`` for one shard."""


def run_sharded_counts(n_total: int, retirement_years: int, local_runner: LocalRunner) -> ReducedCounts:
    """Every rank simulates its shard with ``local_runner`` and the counter vectors are summed with a
    single all-reduce; every rank returns the same global ``ReducedCounts``."""
    import torch
    import torch.distributed as dist

    rank, world = (dist.get_rank(), dist.get_world_size()) if is_active() else (0, 1)
    begin, count = shard_range(int(n_total), rank, world)
    ry = int(retirement_years)
    if count > 0:
        r = local_runner(begin, count)
        vec = pack_counts(r["counters"], r["wr_obs_counts"], r["ruin_year_bins"])
    else:
        vec = np.zeros(2 + ry + ry + 2, dtype=np.int64)
    if world > 1:
        t = torch.as_tensor(vec, device=_comm_device())
        all_reduce_sum_(t)
        vec = t.cpu().numpy()
    return unpack_counts(vec, ry)


def gpu_count_runner(params, seed: int, stream_id: int, working_months: int, device: Optional[int] = None) -> LocalRunner:
    """The product's local runner: count-only HIP kernel over the shard (no per-path HBM traffic)."""
    from . import engine as E

    def run(path_begin: int, n_paths: int) -> Dict[str, np.ndarray]:
        import torch

        dev = torch.cuda.current_device() if device is None else device
        batch = E.DeviceBatch(params, working_months, n_paths, want="count", device=dev)
        batch.launch(seed, stream_id, path_begin)
        return {"counters": batch.counters.cpu().numpy(), "wr_obs_counts": batch.wr_obs_counts.cpu().numpy(),
                "ruin_year_bins": batch.ruin_year_bins.cpu().numpy()}

    return run


def sharded_row_quantiles(rows, n_local: int, qs):
    """Exact GLOBAL row quantiles when every rank holds ``n_local`` entries of each row: the radix
    select's digit histograms are summed across ranks after every pass (8 small all-reduces); the
    rows never move.  Every rank returns the same ``(quantiles, non_nan_counts)``."""
    import torch

    from . import aggregation as A

    if not is_active():
        return A.row_quantiles(rows, n_local, qs)
    tot = torch.tensor([int(n_local)], dtype=torch.int64, device=_comm_device())
    all_reduce_sum_(tot)
    return A.row_quantiles(rows, n_local, qs, reduce_counts=all_reduce_sum_, n_total=int(tot.item()))


def sharded_band_quantiles(batch, n_local: int):
    """``aggregation.band_quantiles`` over rows sharded across the ranks (exact global bands)."""
    import torch

    from . import aggregation as A

    if not is_active():
        return A.band_quantiles(batch, n_local)
    tot = torch.tensor([int(n_local)], dtype=torch.int64, device=_comm_device())
    all_reduce_sum_(tot)
    return A.band_quantiles(batch, n_local, reduce_counts=all_reduce_sum_, n_total=int(tot.item()))


def run_sharded_bands(params, rng_or_seed, stream_id: int, n_total: int, working_months: int, n_bins: int = 100):
    """BASELINE configs[2]/[3] on N GPUs: every rank simulates its shard of the global path range with
    full trajectory output kept in its own HBM; counters and histogram bins are summed, the histogram
    range is min/max-reduced, and the quantile bands come from the distributed radix select.  Returns
    the same dict on every rank."""
    import torch
    import torch.distributed as dist

    from . import aggregation as A
    from . import engine as E

    rank, world = (dist.get_rank(), dist.get_world_size()) if is_active() else (0, 1)
    begin, count = shard_range(int(n_total), rank, world)
    dev = torch.cuda.current_device()
    batch = E.DeviceBatch(params, working_months, max(count, 1), want="full", device=dev)
    if count > 0:
        batch.launch(rng_or_seed, stream_id, begin, count)
    ry = batch.sizes.retirement_years
    vec = batch.reduce_vec
    if world > 1:
        all_reduce_sum_(vec)
    red = unpack_counts(vec.cpu().numpy(), ry)
    out = {"counts": red, "shard": (begin, count)}
    out["trajectory_q"], out["real_trajectory_q"], out["wr_q"], out["wr_counts"] = sharded_band_quantiles(batch, count)
    fb, ok = batch.summary["final_balance"][:max(count, 0)], batch.success[:max(count, 0)]
    out["hist_bins"], out["hist_edges"] = A.success_histogram(
        fb, ok, n_bins,
        reduce_range=all_reduce_minmax_ if world > 1 else None,
        reduce_bins=all_reduce_sum_ if world > 1 else None,
    )
    return out


def run_sharded_histogram(params, rng_or_seed, stream_id: int, n_total: int, working_months: int, n_bins: int = 100,
                          value_range: Optional[Tuple[float, float]] = None, hist_edges=None):
    """BASELINE configs[3] on N GPUs (success counts + histogram of successful final balances, no trajectories).

    * FIXED edges — ``value_range=(lo, hi)`` (``np.histogram(x, bins=n_bins, range=...)``) or explicit ascending
      ``hist_edges`` (``np.histogram(x, bins=hist_edges)``; e.g. ``np.geomspace`` for log-spaced bins): every rank runs the
      COUNT-ONLY kernel over its shard of the global path range and each lane bins its own final balance inside it
      (``mcr_outputs.hist_bins``).  No per-path buffer, no second kernel, and the exchange is ONE sum all-reduce of the
      integer block ``[success, paths, wr_obs[ry], ruin_bins[ry+2], hist_bins[n_bins]]`` — the north-star's shape.
    * data-ranged edges (neither given; ``np.histogram(x, bins=n_bins)`` on the union of the shards): the summary-output
      kernel keeps 49 B/path in the rank's HBM, the cohort's min/max are reduced, then the bins (three collectives).

    Returns the same dict on every rank; ``exchange`` names the collectives that ran."""
    import torch
    import torch.distributed as dist

    from . import aggregation as A
    from . import engine as E

    rank, world = (dist.get_rank(), dist.get_world_size()) if is_active() else (0, 1)
    begin, count = shard_range(int(n_total), rank, world)
    dev = torch.cuda.current_device()
    if hist_edges is not None or value_range is not None:
        edges = E.hist_edge_array(hist_edges) if hist_edges is not None else E.uniform_hist_edges(value_range, n_bins)
        batch = E.DeviceBatch(params, working_months, max(count, 1), want="count", device=dev, hist_edges=edges)
        if count > 0:
            batch.launch(rng_or_seed, stream_id, begin, count)
        vec = batch.reduce_vec
        if world > 1:
            if dist.get_backend() != "nccl":
                vec = vec.cpu()
            all_reduce_sum_(vec)
        host = vec.cpu().numpy()
        n_fixed = host.shape[0] - (edges.shape[0] - 1)
        return {"counts": unpack_counts(host[:n_fixed], batch.sizes.retirement_years), "shard": (begin, count),
                "hist_bins": host[n_fixed:].copy(), "hist_edges": edges,
                "exchange": "none (1 GPU)" if world == 1 else f"1 all-reduce(sum) of {host.shape[0]} int64 words"}
    batch = E.DeviceBatch(params, working_months, max(count, 1), want="summary", device=dev)
    if count > 0:
        batch.launch(rng_or_seed, stream_id, begin, count)
    vec = batch.reduce_vec
    if world > 1:
        all_reduce_sum_(vec)
    red = unpack_counts(vec.cpu().numpy(), batch.sizes.retirement_years)
    bins, edges = A.success_histogram(
        batch.summary["final_balance"][:count], batch.success[:count], n_bins,
        reduce_range=all_reduce_minmax_ if world > 1 else None,
        reduce_bins=all_reduce_sum_ if world > 1 else None,
    )
    return {"counts": red, "shard": (begin, count), "hist_bins": bins, "hist_edges": edges,
            "exchange": "none (1 GPU)" if world == 1 else
            "all-reduce(sum) of the counter vector + all-reduce(min,max) of the range + all-reduce(sum) of the bins"}
