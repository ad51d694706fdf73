"""Drop-in host surface of the engine: ``RetirementMonteCarloSimulator``.

Same class name, constructor, attributes, method names, argument meaning, return shapes and
error behaviour as the reference's simulator (rflamino/monte_carlo_retirement,
backend/simulation.py:126-1342) so FastAPI / CLI / search callers and the reference's own
tests work unchanged — but every path is simulated by the hand-written HIP kernels in csrc/
(through the C ABI of include/mcr.h).  There is no CPU fallback: without a gfx950 device the
compute methods raise ``RuntimeError``.

What differs, by design (documented in DESIGN.md):
* random numbers come from a counter-based Philox4x32-10 + Box-Muller stream keyed by
  ``main_seed`` with counter (path index, month, stream id) instead of NumPy
  SeedSequence -> PCG64 -> ziggurat; ``path_seed`` therefore means *global path index*;
  common random numbers across working-month candidates hold exactly as in the reference;
* the pandas aggregation (quantile bands, counts) is computed on the device with the same
  arithmetic (NumPy ``linear`` interpolation, NaN skipping).
"""

from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import pandas as pd

from . import _native as N
from . import aggregation as A
from . import distributed as D
from . import engine as E
from ._logging import logger
from .config import Config
from .constants import MONTHS_PER_YEAR, SMALL_EPSILON
from .params import arithmetic_to_log_params, params_from_config

__all__ = [
    "RetirementMonteCarloSimulator",
    "arithmetic_to_log_params",
    "retirement_age",
    "stream_payment_start_age",
    "stream_payment_start_month_index",
    "age_at_retirement_year",
    "years_from_t0_to_age",
    "median_first_year_withdrawal_rate",
    "trajectory_time_points",
]

SUMMARY_COLUMNS = [  # simulation.py:1013-1024 (column order of summary_df)
    "Start Balance",
    "Final Balance",
    "Success",
    "YearsToRuin",
    "First Year Gross Withdrawal",
    "First Year Real Gross Withdrawal",
    "Inflation At Retirement",
]
_FIELD_OF = {
    "Start Balance": "start_balance",
    "Final Balance": "final_balance",
    "YearsToRuin": "years_to_ruin",
    "First Year Gross Withdrawal": "first_year_gross_withdrawal",
    "First Year Real Gross Withdrawal": "first_year_real_gross_withdrawal",
    "Inflation At Retirement": "inflation_at_retirement",
}


# ---- module-level helpers (simulation.py:32-123) ----------------------------------------------
def retirement_age(current_age: float, working_months: int) -> float:
    """Age when retirement starts (simulation.py:32-34)."""
    return current_age + working_months / MONTHS_PER_YEAR


def stream_payment_start_age(current_age: float, working_months: int, start_at_age: float) -> float:
    """Payments start once eligible AND retired (simulation.py:37-44)."""
    return max(retirement_age(current_age, working_months), float(start_at_age))


def stream_payment_start_month_index(current_age: float, working_months: int, start_at_age: float) -> int:
    """First retirement-month index paid (simulation.py:47-63): ceil of the gap in months, with the
    reference's 1e-6 guard so an exact month boundary does not round up."""
    gap_years = stream_payment_start_age(current_age, working_months, start_at_age) - retirement_age(
        current_age, working_months
    )
    return max(0, int(math.ceil(gap_years * MONTHS_PER_YEAR - SMALL_EPSILON)))


def age_at_retirement_year(current_age: float, working_months: int, year_num: int) -> float:
    """Age at the start of retirement year ``year_num`` (simulation.py:66-70)."""
    return retirement_age(current_age, working_months) + year_num


def years_from_t0_to_age(current_age: float, target_age: float) -> float:
    """Years from T=0 until ``target_age`` (simulation.py:73-75)."""
    return max(0.0, float(target_age) - float(current_age))


def median_first_year_withdrawal_rate(summary_df: pd.DataFrame) -> float:
    """Median over paths (start balance > eps) of first-year real gross withdrawal / start
    balance, in percent (simulation.py:78-96)."""
    if summary_df.empty:
        return float("nan")
    col = (
        "First Year Real Gross Withdrawal"
        if "First Year Real Gross Withdrawal" in summary_df.columns
        else "First Year Gross Withdrawal"
    )
    start = summary_df["Start Balance"]
    keep = start > SMALL_EPSILON
    if not keep.any():
        return float("nan")
    return float(((summary_df[col][keep] / start[keep]) * 100.0).median())


def trajectory_time_points(working_months: int, retirement_years: int) -> List[float]:
    """Year values of the yearly trajectory samples (simulation.py:99-123)."""
    full_years, leftover = divmod(working_months, MONTHS_PER_YEAR)
    t_ret = working_months / MONTHS_PER_YEAR
    pts = [0.0] + [float(y) for y in range(1, full_years + 1)]
    if leftover:
        pts.append(t_ret)
    pts.extend(t_ret + y for y in range(1, retirement_years + 1))
    return pts


def _gather_columns(rows, picked) -> np.ndarray:
    """``rows[:, picked].T`` as a host array.  Each pick is a strided column VIEW (``select`` is metadata only: the
    64-bit offset is folded into the view's data pointer on the host) and the stack kernel then indexes k tensors of
    T elements each.  No torch INDEXING kernel (``index_select`` / advanced indexing) ever sees the [2T+ry, stride]
    slab, which exceeds 2**32 elements from ~3.2e7 paths on: the one GPU-side abort of round 1 was such a gather
    on a 3.3e7-path slab (LABNOTES.md, rounds 1-3 section 11)."""
    import torch

    return torch.stack([rows[:, int(g)] for g in picked]).cpu().numpy()


class _SummaryDownload:
    """summary_df (simulation.py:1012-1027) from a device batch, in two halves so that the transfer overlaps the device
    work that follows the path kernel.  `start`: the six float columns and the flags go device->host as plain contiguous 1-D
    copies into ONE pinned [6, n] buffer (no stack / indexing kernel on the device, so the element count any torch kernel sees
    stays n whatever the batch size) on a COPY STREAM of its own that waits for the path kernel only — 56 bytes a path: 560 MB
    at 10^7 paths, tens of milliseconds of PCIe time that used to sit, serially, between the path kernel and the band
    selection (round 3: 131 ms end to end for 77 ms of kernels).  `frame`: waits for the copy stream and wraps the buffer in a
    no-copy DataFrame with the reference's column order and dtypes."""

    def __init__(self, batch, n: int):
        torch = batch.torch
        dev = batch.success.device
        self.fields = list(_FIELD_OF.items())
        main = torch.cuda.current_stream(dev)
        self.stream = _copy_stream(torch, dev)
        # (the pinned buffers come from torch's caching host allocator: allocated while the path kernel runs)
        self.host = torch.empty((len(self.fields), n), dtype=torch.float64, pin_memory=True)
        self.flags = torch.empty(n, dtype=torch.uint8, pin_memory=True)
        self.stream.wait_stream(main)                    # = after the path kernel that was just enqueued
        with torch.cuda.stream(self.stream):
            for i, (_, f) in enumerate(self.fields):
                self.host[i].copy_(batch.summary[f][:n], non_blocking=True)
            self.flags.copy_(batch.success[:n], non_blocking=True)
        self._batch = batch                               # the source tensors stay alive until the copies have run

    def frame(self) -> pd.DataFrame:
        self.stream.synchronize()
        self._batch = None
        h = self.host.numpy()
        cols = {name: h[i] for i, (name, _) in enumerate(self.fields)}
        cols["Success"] = self.flags.numpy().view(np.bool_)
        return pd.DataFrame({c: cols[c] for c in SUMMARY_COLUMNS}, copy=False)


_copy_streams: Dict[Tuple[int, int], object] = {}


def _copy_stream(torch, dev):
    """One side stream per (thread, device) for device->host transfers that overlap kernels on the caller's stream."""
    import threading

    key = (threading.get_ident(), dev.index or 0)
    st = _copy_streams.get(key)
    if st is None:
        st = _copy_streams[key] = torch.cuda.Stream(device=dev)
    return st


def _summary_frame(batch, n: int) -> pd.DataFrame:
    """The whole download in one call (callers with nothing to overlap it with)."""
    return _SummaryDownload(batch, n).frame()


class _BackgroundCall:
    """`fn(*args)` on a host thread of its own; `result()` joins and returns its value or re-raises its exception."""

    def __init__(self, fn, *args):
        import threading

        self._value, self._error = None, None

        def run():
            try:
                self._value = fn(*args)
            except BaseException as exc:  # noqa: BLE001  (handed to the caller's thread)
                self._error = exc

        self._thread = threading.Thread(target=run, daemon=True)
        self._thread.start()

    def result(self):
        self._thread.join()
        if self._error is not None:
            raise self._error
        return self._value


def sample_columns(seed: int, n: int, k: int):
    """``numpy.random.RandomState(seed).choice(n, k, replace=False)`` — the column indices pandas' ``DataFrame.sample(n=k,
    axis=1, random_state=seed)`` picks (simulation.py:1063-1078) — or None (error logged) where NumPy raises, as the
    reference does (:1079-1083; e.g. seed >= 2**32).  NumPy shuffles all n indices for it; the library's restatement
    (``mcr_sample_columns``: same MT19937 draws, the k positions traced back through the swaps) returns the same indices in
    about half the time — it has to hide under a kernel launch of n paths."""
    if 0 <= int(seed) < 2**32 and 1 <= k <= 64 and k <= n <= 2**32:
        out = np.empty(k, dtype=np.int64)
        rc = N.load_library().mcr_sample_columns(int(seed), int(n), int(k), out.ctypes.data)
        if rc == 0:
            return out
    try:
        return np.random.RandomState(seed).choice(int(n), size=int(k), replace=False)
    except ValueError as ve:
        logger.error(f"Error sampling trajectories: {ve}")
        return None


def _fold_seed_u64(seed: int) -> int:
    """Philox key = the seed folded to 64 bits (seeds are unbounded Python ints in the Config)."""
    s, out = int(seed), 0
    while True:
        out ^= s & 0xFFFFFFFFFFFFFFFF
        s >>= 64
        if s == 0:
            return out


class RetirementMonteCarloSimulator:
    """Monte Carlo retirement simulator whose paths run on an MI355X (see module docstring)."""

    def __init__(self, params_model: Config, main_seed_override: Optional[int] = None, device: int = 0,
                 rng: str = "philox"):
        """``rng="philox"`` (default): the engine's counter-based stream; ``path_seed`` = global path
        index.  ``rng="numpy"``: the reference's OWN stream reproduced on the device (SeedSequence ->
        PCG64 -> ziggurat): the same ``seed`` then yields the reference's numbers, ``path_seed`` is the
        reference's uint32 path seed and `_path_seeds` follows its spawn-and-cache rule (:187-199)."""
        if rng not in ("philox", "numpy"):
            raise ValueError("rng must be 'philox' or 'numpy'")
        self.rng = rng
        self.params_model = params_model.model_copy(deep=True)  # simulation.py:136

        if main_seed_override is not None:  # :138-145
            if main_seed_override < 0:
                raise ValueError("main_seed_override must be nonnegative.")
            self.main_seed = main_seed_override
        elif self.params_model.seed is not None:
            self.main_seed = self.params_model.seed
        else:
            # No seed configured: the reference hashes the current time (utils.py:16-18).  With one process per
            # GPU every rank would draw its OWN seed — different Philox keys per shard, different sampled
            # columns per rank — so rank 0's seed is broadcast and every rank uses it.
            self.main_seed = D.broadcast_int(_seed_from_timestamp()) if D.is_active() else _seed_from_timestamp()

        # Two independent streams (search vs final run, :147-151): the stream id is one word of the
        # Philox counter, so the streams never overlap.
        self._stream_name = "final"
        self._engine_seed = _fold_seed_u64(self.main_seed)
        self.device = int(device)
        #: under torch.distributed, batches smaller than this are computed in full on EVERY rank (identical
        #: results, no communication): a 50 000-path probe is latency-bound (~1 ms) and sharding it would add
        #: an all-reduce + host sync per probe.  Larger batches are sharded across the ranks.
        self.shard_min_paths = 1_000_000
        #: under torch.distributed, sharded batches of MORE paths than this deliver the per-path summary frame (n x 7: 56 bytes
        #: a path, 5.6 GB at 1e8 paths) to rank 0 ONLY — the other ranks get a frame with the reference's columns and no rows —
        #: instead of all-gathering it into the host memory of every rank.  Everything else of the 7-tuple (bands, sampled
        #: paths, observation counts) is identical on all ranks either way; `results.compact_result` needs no per-path frame.
        self.gather_all_max_paths = 20_000_000
        #: which ranks hold the rows of the last sharded run's `summary_df`: "all", or "rank0" (a deliberate empty frame elsewhere)
        self.last_summary_scope = "all"
        # NumPy stream bookkeeping: children spawned so far per stream, and the offset at which each
        # (stream, n) batch was spawned — the reference's _path_seed_cache rule (:154, :192-199)
        self._np_children_spawned = {"search": 0, "final": 0}
        self._np_batch_offset: Dict[Tuple[str, int], int] = {}

        p = self.params_model
        self._inv1_mu_log, self._inv1_sigma_log = arithmetic_to_log_params(  # :157-166
            p.inv1_returns_mean, p.inv1_returns_volatility
        )
        self._inf_mu_log, self._inf_sigma_log = arithmetic_to_log_params(
            p.inflation_rate_mean, p.inflation_rate_volatility
        )
        self._inv2_prem_mu_log, self._inv2_prem_sigma_log = arithmetic_to_log_params(
            p.inv2_premium_over_inflation_mean, p.inv2_premium_over_inflation_volatility
        )
        self._equity_inflation_rho = p.equity_inflation_correlation  # :170
        self._params = params_from_config(p)
        logger.info(
            f"Simulator initialized for scenario '{p.Nickname}' with main seed: {self.main_seed}"
        )

    # ---- seed streams (:177-199) ---------------------------------------------------------------
    @property
    def _stream_id(self) -> int:
        return N.MCR_STREAM_SEARCH if self._stream_name == "search" else N.MCR_STREAM_FINAL

    def use_search_seeds(self) -> None:
        self._stream_name = "search"

    def use_final_seeds(self) -> None:
        self._stream_name = "final"

    def _path_seeds(self, num_simulations: int) -> List[int]:
        """Path identifiers of a batch.  Philox stream: a path's "seed" is its global index in the
        active stream.  NumPy stream: the reference's uint32 seeds, spawned once per (stream, n) and
        cached (:187-199).  Either way the list is the same for every working-month candidate."""
        n = int(num_simulations)
        if self.rng == "philox":
            return list(range(n))
        off = self._np_offset(n)  # children (stream, off .. off+n-1) of SeedSequence(main_seed)
        return [
            int(np.random.SeedSequence(self.main_seed, spawn_key=(self._stream_id, off + j)).generate_state(1)[0])
            for j in range(n)
        ]

    def _local_device(self) -> int:
        """The GPU this process computes on: the configured device, or — with one process per GPU under
        torch.distributed — the rank's current device."""
        if D.is_active():
            import torch

            return int(torch.cuda.current_device())
        return self.device

    def _np_offset(self, n: int) -> int:
        """Spawn offset of the (active stream, n) batch: assigned on first use, then cached."""
        key = (self._stream_name, int(n))
        if key not in self._np_batch_offset:
            self._np_batch_offset[key] = self._np_children_spawned[self._stream_name]
            self._np_children_spawned[self._stream_name] += int(n)
        return self._np_batch_offset[key]

    def _batch_rng(self, n: int):
        """RNG descriptor of an n-path batch on the active stream (int = Philox key)."""
        if self.rng == "philox":
            return self._engine_seed
        return N.numpy_rng(self.main_seed, child_offset=self._np_offset(n))

    # ---- scalar helpers: evaluated by the SAME device functions the path kernel inlines -----------
    def _calculate_withdrawal_and_update(
        self, bal_inv: float, cb_inv: float, net_withdrawal_target_for_inv: float,
        use_real_tax: bool, real_tax_rate: float,
    ) -> Tuple[float, float, float, float]:
        """(new_balance, new_cost_basis, gross_withdrawal, net_cash) — simulation.py:201-254."""
        r = E.eval_helper_host(
            N.MCR_HELPER_WITHDRAW, None,
            [[bal_inv, cb_inv, net_withdrawal_target_for_inv, 1.0 if use_real_tax else 0.0, real_tax_rate]],
            self._local_device(),
        )[0]
        return float(r[0]), float(r[1]), float(r[2]), float(r[3])

    def _net_liquidation_value(
        self, balance: float, cost_basis: float, use_realized_gains_tax: bool, realized_gains_tax_rate: float
    ) -> float:
        """Cash after liquidating an asset and paying gains tax — simulation.py:256-272."""
        return float(E.eval_helper_host(
            N.MCR_HELPER_NLV, None,
            [[balance, cost_basis, 1.0 if use_realized_gains_tax else 0.0, realized_gains_tax_rate]],
            self._local_device(),
        )[0][0])

    def _rebalance_portfolio(
        self, bal_inv1: float, cb_inv1: float, bal_inv2: float, cb_inv2: float
    ) -> Tuple[float, float, float, float]:
        """Tax-aware rebalance to the target allocation — simulation.py:274-359."""
        r = E.eval_helper_host(N.MCR_HELPER_REBALANCE, self._current_params(),
                               [[bal_inv1, cb_inv1, bal_inv2, cb_inv2]], self._local_device())[0]
        return float(r[0]), float(r[1]), float(r[2]), float(r[3])

    def _apply_annual_gain_taxes(
        self, balance_inv1: float, cost_basis_inv1: float, balance_inv2: float, cost_basis_inv2: float,
        gain_inv1: float, gain_inv2: float,
    ) -> Tuple[float, float, float, float, bool]:
        """Annual mark-to-market tax for one period — simulation.py:361-450."""
        r = E.eval_helper_host(
            N.MCR_HELPER_ANNUAL_TAX, self._current_params(),
            [[balance_inv1, cost_basis_inv1, balance_inv2, cost_basis_inv2, gain_inv1, gain_inv2]],
            self._local_device(),
        )[0]
        return float(r[0]), float(r[1]), float(r[2]), float(r[3]), bool(r[4])

    def _monthly_gross_from_shock(self, mu_log: float, sigma_log: float, z: float) -> float:
        """exp(mu/12 + sigma/sqrt(12) z) — simulation.py:468-474."""
        return float(E.eval_helper_host(N.MCR_HELPER_MONTHLY_GROSS, None, [[mu_log, sigma_log, z]], self._local_device())[0][0])

    def _draw_shock_path(self, n_months: int, path_seed: int) -> np.ndarray:
        """Shock rows (equity, inflation, premium) of one path, shape (n_months, 3) — the engine's
        replacement of simulation.py:452-466."""
        if self.rng == "numpy":
            return E.draw_shocks_host(
                N.numpy_rng(self.main_seed), self._stream_id, 0, 1, int(n_months), self._equity_inflation_rho,
                self._local_device(), path_seeds=np.array([int(path_seed)], dtype=np.uint32),
            )[0]
        return E.draw_shocks_host(
            self._engine_seed, self._stream_id, int(path_seed), 1, int(n_months), self._equity_inflation_rho,
            self._local_device(),
        )[0]

    def _current_params(self):
        """The parameter block of the *current* params_model (validate_assignment lets callers
        mutate the config between runs, as they can in the reference)."""
        self._params = params_from_config(self.params_model)
        return self._params

    # ---- one path (:476-950) -------------------------------------------------------------------
    def _run_single_simulation_path(
        self, working_months: int, path_seed: int
    ) -> Dict[str, Union[float, List[float]]]:
        """Simulate ONE path on the device and return the reference's 10-key dict."""
        if self.rng == "numpy":
            r = E.run_batch_host(
                self._current_params(), N.numpy_rng(self.main_seed), self._stream_id, 0, 1,
                int(working_months), want_bins=False, device=self._local_device(),
                path_seeds=np.array([int(path_seed)], dtype=np.uint32),
            )
        else:
            r = E.run_batch_host(
                self._current_params(), self._engine_seed, self._stream_id, int(path_seed), 1,
                int(working_months), want_bins=False, device=self._local_device(),
            )
        return {
            "Start Balance": float(r["start_balance"][0]),
            "Final Balance": float(r["final_balance"][0]),
            "Success": bool(r["success"][0]),
            "YearsToRuin": float(r["years_to_ruin"][0]),
            "First Year Gross Withdrawal": float(r["first_year_gross_withdrawal"][0]),
            "First Year Real Gross Withdrawal": float(r["first_year_real_gross_withdrawal"][0]),
            "Trajectory": r["trajectory"][:, 0].tolist(),
            "RealTrajectory": r["real_trajectory"][:, 0].tolist(),
            "WithdrawalRateTrajectory": r["withdrawal_rate_trajectory"][:, 0].tolist(),
            "Inflation At Retirement": float(r["inflation_at_retirement"][0]),
        }

    # ---- batch driver (:952-1128) --------------------------------------------------------------
    def run_monte_carlo_simulations(
        self, working_months: int, num_simulations: int
    ) -> Tuple[
        pd.DataFrame,
        Optional[pd.DataFrame],
        Optional[List[List[float]]],
        Optional[pd.DataFrame],
        Optional[pd.DataFrame],
        Optional[List[List[float]]],
        Optional[List[int]],
    ]:
        """Simulate ``num_simulations`` paths in ONE kernel launch and aggregate on the device.

        Returns the reference's 7-tuple: ``(summary_df, trajectory_percentiles_df,
        sample_trajectories, wr_percentiles_df, real_trajectory_percentiles_df,
        sample_real_trajectories, wr_observation_counts)``.
        """
        n = int(num_simulations)
        wm = int(working_months)
        if D.is_active() and n >= self.shard_min_paths:
            return self._run_sharded(wm, n)
        dev = self._local_device()
        logger.debug(f"Running {n} simulations on HIP device {dev} for {wm} working months.")
        batch = E.DeviceBatch(self._current_params(), wm, n, want="full", device=dev)
        batch.launch(self._batch_rng(n), self._stream_id, 0)
        # The 5 sampled columns are the ones trajectory_df.sample(n=5, axis=1, random_state=main_seed) picks (:1063-1078):
        # pandas draws them with RandomState(seed).choice(n, 5, replace=False), which permutes all n indices — 7 ms at 1e6
        # paths, as long as the kernel itself.  It is host-only work: done HERE, while the (asynchronous) launch runs.
        # (at 10^7 paths the draw is 30-50 ms of sequential host work — as long as the path kernel: it runs on a host thread of
        #  its own (the library call releases the GIL) and is joined where the sampled columns are gathered)
        sampler = _BackgroundCall(self._sample_columns, n) if n >= 200_000 else None
        picked = None if sampler else self._sample_columns(n)
        # Device -> host of the per-path summary (56 B / path) on the copy stream, behind the path kernel only; the band
        # selection (K3) is enqueued on the main stream right after and runs while the summary crosses the host link; the
        # frame is built last (the reference builds it first, simulation.py:1012-1027: same values, another order of work)
        download = _SummaryDownload(batch, n)

        traj_q, real_q, wr_q, wr_counts = A.band_quantiles(batch, n)
        qcols = pd.Index(list(A.TRAJECTORY_QUANTILES), dtype="float64")
        trajectory_percentiles_df = pd.DataFrame(traj_q, columns=qcols)
        real_trajectory_percentiles_df = pd.DataFrame(real_q, columns=qcols)
        wr_percentiles_df = pd.DataFrame(wr_q, columns=pd.Index(list(A.WR_QUANTILES), dtype="float64"))
        wr_observation_counts = [int(v) for v in wr_counts.tolist()]

        sample_trajectories_list: Optional[List[List[float]]] = None
        sample_real_trajectories_list: Optional[List[List[float]]] = None
        if sampler:
            picked = sampler.result()
        if picked is not None:
            sample_trajectories_list = _gather_columns(batch.trajectory, picked).tolist()
            sample_real_trajectories_list = _gather_columns(batch.real_trajectory, picked).tolist()
        summary_df = download.frame()
        return (
            summary_df,
            trajectory_percentiles_df,
            sample_trajectories_list,
            wr_percentiles_df,
            real_trajectory_percentiles_df,
            sample_real_trajectories_list,
            wr_observation_counts,
        )

    def _sample_columns(self, n: int):
        """Indices of the sampled paths — ``RandomState(main_seed).choice(n, min(n, 5), replace=False)``, what pandas'
        ``sample(n=5, axis=1, random_state=main_seed)`` draws (:1063-1078) — or None where the reference logs an error and
        returns no samples (e.g. main_seed >= 2**32, :1079-1083).  The same on every rank of a process group."""
        k = min(int(n), 5)
        if k <= 0:
            return None
        return sample_columns(self.main_seed, int(n), k)

    def _run_sharded(self, wm: int, n: int):
        """run_monte_carlo_simulations with one process per GPU (torch.distributed initialised): every rank
        simulates its shard of the global path range [0, n) and keeps its trajectories in its own HBM; the
        per-path summary is all-gathered (49 B/path; above `gather_all_max_paths` paths: gathered to rank 0 only,
        the reference's `summary_df` contract being a single-process one, simulation.py:1012-1027), the quantile bands come from the distributed radix
        select (digit histograms summed across ranks), the sampled paths are contributed by the rank that
        owns them.  Every rank returns the same bands, sampled paths and observation counts, bit-identical to the single-GPU
        result; `summary_df` is the single-GPU frame on every rank up to `gather_all_max_paths` paths and, above that, on rank 0
        ONLY — the other ranks then return a frame with the reference's columns and NO rows, and
        `self.last_summary_scope == "rank0"` tells such a caller that the empty frame is deliberate ("all" otherwise)."""
        import torch
        import torch.distributed as dist

        rank, world = dist.get_rank(), dist.get_world_size()
        dev = torch.cuda.current_device()
        begin, count = D.shard_range(n, rank, world)
        per = -(-n // world)
        batch = E.DeviceBatch(self._current_params(), wm, max(count, 1), want="full", device=dev)
        if count > 0:
            batch.launch(self._batch_rng(n), self._stream_id, begin, count)
        sampler = _BackgroundCall(self._sample_columns, n)   # (host work on a thread of its own; identical on every rank)
        comm = D._comm_device()
        # ---- per-path summary: pack [7, per] (six doubles + the flag), all-gather, trim ----
        fields = list(_FIELD_OF.values())
        local = torch.zeros((len(fields) + 1, per), dtype=torch.float64, device=batch.success.device)
        if count > 0:
            for i, f in enumerate(fields):
                local[i, :count] = batch.summary[f][:count]
            local[len(fields), :count] = batch.success[:count].to(torch.float64)
        local = local.to(comm)
        to_rank0_only = n > self.gather_all_max_paths
        self.last_summary_scope = "rank0" if to_rank0_only else "all"
        if to_rank0_only:
            logger.warning(f"{n} paths over {world} ranks: the per-path summary frame goes to rank 0 only "
                           f"(gather_all_max_paths = {self.gather_all_max_paths}); the other ranks return an empty frame")
            gathered = [torch.empty_like(local) for _ in range(world)] if rank == 0 else None
            pending = dist.gather(local, gathered, dst=0, async_op=True)
        else:
            gathered = [torch.empty_like(local) for _ in range(world)]
            pending = dist.all_gather(gathered, local, async_op=True)
        # ---- bands: enqueued while the summary gather is in flight (the selection's own small collectives queue behind it) ----
        traj_q, real_q, wr_q, wr_counts = D.sharded_band_quantiles(batch, count)
        pending.wait()
        del local
        if gathered is None:
            summary_df = pd.DataFrame({c: pd.Series(dtype=bool if c == "Success" else np.float64) for c in SUMMARY_COLUMNS})
        else:
            # trimmed and joined on the HOST (64-bit NumPy indexing whatever n is)
            allf = np.concatenate([g[:, :D.shard_range(n, r, world)[1]].cpu().numpy() for r, g in enumerate(gathered)], axis=1)
            del gathered
            cols = {name: allf[i] for i, name in enumerate(_FIELD_OF.keys())}
            cols["Success"] = allf[len(fields)] != 0.0
            summary_df = pd.DataFrame({c: cols[c] for c in SUMMARY_COLUMNS})
        qcols = pd.Index(list(A.TRAJECTORY_QUANTILES), dtype="float64")
        trajectory_percentiles_df = pd.DataFrame(traj_q, columns=qcols)
        real_trajectory_percentiles_df = pd.DataFrame(real_q, columns=qcols)
        wr_percentiles_df = pd.DataFrame(wr_q, columns=pd.Index(list(A.WR_QUANTILES), dtype="float64"))
        wr_observation_counts = [int(v) for v in wr_counts.tolist()]
        # ---- the 5 sampled paths: owner ranks fill their columns, the rest stays 0, sum-reduce ----
        samples = real_samples = None
        picked = sampler.result()
        if picked is not None:    # (None on every rank or on none: same seed, same n — the collective below cannot be skipped by one rank)
            T = batch.sizes.trajectory_len
            buf = torch.zeros((2, len(picked), T), dtype=torch.float64, device=batch.trajectory.device)
            for j, g in enumerate(picked):
                if begin <= g < begin + count:
                    buf[0, j] = batch.trajectory[:, g - begin]
                    buf[1, j] = batch.real_trajectory[:, g - begin]
            buf = buf.to(comm)
            D.all_reduce_sum_(buf)
            samples = buf[0].cpu().numpy().tolist()
            real_samples = buf[1].cpu().numpy().tolist()
        return (summary_df, trajectory_percentiles_df, samples, wr_percentiles_df,
                real_trajectory_percentiles_df, real_samples, wr_observation_counts)

    def _success_probability(self, summary_df: pd.DataFrame) -> float:
        """Share of paths that funded all spending, in percent (simulation.py:1130-1136)."""
        if summary_df.empty:
            return 0.0
        if "Success" in summary_df.columns:
            return float(summary_df["Success"].astype(bool).mean() * 100.0)
        return float((summary_df["Final Balance"] > SMALL_EPSILON).mean() * 100.0)

    # ---- count-only probes used by the search ------------------------------------------------------
    def _probe_many(self, months: Sequence[int], num_simulations: int) -> Dict[int, float]:
        """Success % of several candidate working-month counts over the same paths, from count-only
        kernels run concurrently (``mcr_probe_months_rng``).  Each value equals
        ``_success_probability(run_monte_carlo_simulations(m, n)[0])`` bit-for-bit: count/n*100.

        Under a process group: batches of at least ``shard_min_paths`` are sharded by path range (every
        rank counts all candidates on its shard); smaller ones are split by CANDIDATE (rank r takes
        ``months[r::world]`` over the whole range).  Either way the counter block is summed with one
        all-reduce and every rank sees the same probabilities (and replays the same search)."""
        months = [int(m) for m in months]
        n = int(num_simulations)
        params, rng, dev = self._current_params(), self._batch_rng(n), self._local_device()

        def probe(path_begin, count, ms):
            return E.probe_months(params, rng, self._stream_id, path_begin, count, ms, device=dev)

        counts = D.probe_candidates(months, n, self.shard_min_paths, probe)
        return {m: float(np.float64(int(counts[i, N.MCR_CTR_SUCCESS])) / np.float64(n) * 100.0) for i, m in enumerate(months)}

    def _probe_success_probability(self, working_months: int, num_simulations: int) -> float:
        """Success % of one batch from the count-only kernel (no per-path HBM traffic)."""
        return self._probe_many([working_months], num_simulations)[int(working_months)]

    def _speculation_slots(self, num_simulations: int) -> int:
        """How many candidate months one round of the search may evaluate together at the cost of (about) one.
        Under a process group small batches are split by candidate across the ranks: every rank takes one candidate for
        free.  On ONE GPU a small probe is latency-bound — 50 000 paths are 782 wavefronts on 1 024 SIMDs — and candidates of
        one call share their accumulation sweep (`mcr_probe_months_rng`): measured at 50 000 paths, months 217+, one candidate
        1.19 ms, two 1.40, four 1.82 (tools/probe_latency.py).  Three candidates a round = the bisection's midpoint and both
        midpoints of the next level (two levels per round), or three bracket points: 1.6 ms instead of 2 x 1.19.  A month the
        reference would not have probed never reaches the curve or the callback (`ahead` in find_minimum_working_months).
        Large probes (>= 200 000 paths) fill the chip on their own: nothing is evaluated on speculation there."""
        if D.is_active() and int(num_simulations) < self.shard_min_paths:
            import torch.distributed as dist

            return dist.get_world_size()
        return 3 if int(num_simulations) < 200_000 else 1

    # ---- search driver (:1138-1342) ------------------------------------------------------------
    def find_minimum_working_months(
        self,
        verbose: bool = True,
        progress_callback: Optional[Callable[[dict], None]] = None,
    ) -> Tuple[int, float, List[Dict[str, float]]]:
        """Smallest working-month count reaching the target success probability.

        Same procedure as the reference: coarse bracket (step 12, grown to 24 while > 20 points
        short), bisection, then every month of the statistically plausible window is verified
        (3-sigma binomial margin) because Monte Carlo estimates are locally non-monotone.  Uses the
        search stream with common random numbers.  Returns ``(months, probability, search_curve)``;
        ``months == -1`` when the target is not reachable within 70 years.
        """
        self.use_search_seeds()
        p = self.params_model
        first = p.starting_working_months_search
        target = p.target_probability
        n_sims = p.num_simulations_search
        horizon = first + 70 * MONTHS_PER_YEAR
        curve: List[Dict[str, float]] = []
        memo: Dict[int, float] = {}
        state = {"iter": 0, "best_seen": -1.0, "lo": first, "hi": None}
        # the reference's tests (and callers) may replace run_monte_carlo_simulations per instance;
        # then the search must go through it.  Otherwise use the count-only kernels, and evaluate
        # candidates the driver is about to ask for TOGETHER with the one it asks for now (they share
        # the GPU concurrently / are split across ranks).  `ahead` only holds results early: months
        # are still reported one by one, in the reference's order, and a month the reference would
        # not have probed never reaches memo, the curve or the callback.
        # (replaced on the instance, by a subclass, or by patching the class attribute)
        patched = ("run_monte_carlo_simulations" in self.__dict__
                   or getattr(type(self), "run_monte_carlo_simulations", None) is not _ENGINE_RUN)
        ahead: Dict[int, float] = {}
        slots = 1 if patched else self._speculation_slots(n_sims)

        if verbose:
            logger.info(f"Estimating working months to achieve {target:.2f}% success for '{p.Nickname}'.")
            logger.info(f"Starting search from {first} months. Simulations per test: {n_sims}.")

        def probe(months: int, likely_next: Sequence[int] = ()) -> float:
            if months in memo:
                return memo[months]
            state["iter"] += 1
            if verbose:
                logger.info(f"Search iter {state['iter']}: Testing {months} m "
                            f"({months / MONTHS_PER_YEAR:.1f} yrs) with {n_sims} sims.")
            if patched:
                summary_df = self.run_monte_carlo_simulations(months, n_sims)[0]
                prob = self._success_probability(summary_df)
            else:
                if months not in ahead:
                    batch = [months] + [m for m in dict.fromkeys(likely_next)
                                        if m != months and m not in ahead and m not in memo]
                    ahead.update(self._probe_many(batch, n_sims))
                prob = ahead[months]
            memo[months] = prob
            if verbose:
                logger.info(f"  Search iter {state['iter']}: Prob for {months} m: {prob:.2f}% (Target: {target:.2f}%)")
            years = round(months / MONTHS_PER_YEAR, 1)
            curve.append({"working_months": months, "working_years": years, "probability": round(prob, 2)})
            if progress_callback:
                progress_callback({
                    "type": "search_iter", "iteration": state["iter"], "working_months": months,
                    "working_years": years, "probability": round(prob, 2), "target": target,
                    "sim_count": n_sims, "lo": state["lo"], "hi": state["hi"],
                })
            state["best_seen"] = max(state["best_seen"], prob)
            return prob

        def bisection_mids(lo: int, hi: int, budget: int) -> List[int]:
            """Midpoints the bisection can reach from (lo, hi), breadth first, at most ``budget``."""
            out: List[int] = []
            frontier = [(lo, hi)]
            while frontier and len(out) < budget:
                nxt_frontier = []
                for a, b in frontier:
                    if b - a > 1 and len(out) < budget:
                        mid = (a + b) // 2
                        out.append(mid)
                        nxt_frontier += [(a, mid), (mid, b)]
                frontier = nxt_frontier
            return out

        # phase 1: bracket
        step = 12
        at = first
        prob_lo = probe(at)
        if prob_lo >= target:
            if verbose:
                logger.info(f"  Target met at starting point {at} months.")
            return at, prob_lo, curve
        best_prob = prob_lo
        while at < horizon:
            shortfall = target - prob_lo
            step = max(step, 24 if shortfall > 20 else (12 if shortfall > 10 else 6))
            nxt = min(at + step, horizon)
            if nxt <= at:
                break
            # the step never shrinks, so the following bracket points are (almost always) nxt + k*step
            prob = probe(nxt, [min(nxt + k * step, horizon) for k in range(1, slots)])
            if prob >= target:
                state["lo"], state["hi"], best_prob = at, nxt, prob
                if verbose:
                    logger.info(f"  Bracketed: lo={at} m (miss), hi={nxt} m (hit). Bisecting…")
                if progress_callback:
                    progress_callback({"type": "search_refining", "working_months": nxt, "lo": at, "hi": nxt})
                break
            state["lo"] = nxt
            prob_lo = prob
            at = nxt
        if state["hi"] is None:
            if verbose:
                logger.warning(f"Search for '{p.Nickname}' reached max limit "
                               f"({horizon / MONTHS_PER_YEAR:.1f} yrs). Target NOT met.")
                logger.warning(f"Highest probability achieved: {state['best_seen']:.2f}%.")
            return -1, state["best_seen"], curve

        # phase 2: bisect, remembering the smallest month that met the target
        best = state["hi"]
        while state["hi"] - state["lo"] > 1:
            mid = (state["lo"] + state["hi"]) // 2
            prob = probe(mid, bisection_mids(state["lo"], state["hi"], slots))
            if prob >= target:
                best, best_prob = mid, prob
                state["hi"] = mid
            else:
                state["lo"] = mid

        # phase 3: verify every month from one tested point before the first plausible month
        margin = min(100.0, 150.0 / math.sqrt(n_sims))
        tested = sorted(m for m in memo if m <= best)
        near = next((i for i, m in enumerate(tested) if memo[m] >= target - margin), len(tested) - 1)
        verify_from = max(first, tested[max(0, near - 1)])
        if verbose:
            logger.info(f"  Verifying each month from {verify_from} to {best} "
                        "to handle locally non-monotone Monte Carlo estimates.")
        window = list(range(verify_from, best + 1))
        for month in window:
            probe(month, window)   # every month of the window is needed: evaluate them in one batch
        hits = [m for m, pr in memo.items() if first <= m <= best and pr >= target]
        if hits:
            best = min(hits)
            best_prob = memo[best]
        if verbose:
            logger.info(f"  Search complete: estimated minimum {best} months "
                        f"({best / MONTHS_PER_YEAR:.1f} yrs) with prob {best_prob:.2f}%.")
        return best, best_prob, curve


#: the engine's own batch driver: the search takes the count-only fast path only while this is what a call reaches
_ENGINE_RUN = RetirementMonteCarloSimulator.run_monte_carlo_simulations


def _seed_from_timestamp() -> int:
    """Seed when none is configured: hash of the UTC timestamp (backend/utils.py:16-18)."""
    import datetime as _dt
    import hashlib

    ts = _dt.datetime.now(_dt.timezone.utc).isoformat()
    return int.from_bytes(hashlib.sha256(ts.encode()).digest()[:8], "big") % (2**32 - 1)
