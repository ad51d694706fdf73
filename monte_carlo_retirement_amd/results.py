"""Caller-side data formats of the path: the response payload and the run flow around it.

The reference's server turns the 7-tuple of ``run_monte_carlo_simulations`` into the JSON document its
UI plots (``_build_result``, backend/server.py:416-562; schema ``SimulationResponse``, :103-112) and
wraps search + final run into one flow that reports progress events (``_run_simulation`` :231-266 and
the SSE generator :322-413).  This module restates those two pieces on top of the drop-in simulator so a
caller (the reference's FastAPI app, its CLI, a batch job) gets the same document and the same event
sequence — plus a payload for batches far larger than the UI's, assembled from device-side aggregates
only (``compact_result``): no per-path list ever crosses PCIe.

Parity: tests/golden/server_stream.json holds the events and payloads the REFERENCE's own app produced
(FastAPI TestClient, engine shocks injected); ``assemble_result`` reproduces them exactly from the same
7-tuple, and the GPU route reproduces them end to end.
"""

from __future__ import annotations

import math
from typing import Any, Callable, Dict, List, Optional, Sequence

import numpy as np
import pandas as pd

from . import aggregation as A
from . import distributed as D
from . import engine as E
from ._native import MCR_N_COUNTERS as N_COUNTERS
from ._logging import logger
from .config import Config
from .constants import MONTHS_PER_YEAR, SMALL_EPSILON
from .simulation import (
    RetirementMonteCarloSimulator,
    median_first_year_withdrawal_rate,
    retirement_age,
    stream_payment_start_month_index,
    trajectory_time_points,
)

FINAL_BALANCE_PERCENTILES = (0.01, 0.05, 0.10, 0.25, 0.50, 0.75, 0.90, 0.95, 0.99)  # server.py:456-458


def _pct_key(q: float) -> str:
    return f"p{int(q * 100)}"          # the reference's key rule (0.05 -> "p5", 0.1 -> "p10"), server.py:216,509


def _r2(v) -> float:
    return round(float(v), 2)          # Python's correctly-rounded decimal rounding, as the reference uses


def _finite_or_none(v: float) -> Optional[float]:
    """NaN / inf -> None (JSON has neither), else rounded to cents (server.py:188-192)."""
    return _r2(v) if math.isfinite(v) else None


def dedupe_search_curve(points: Sequence[dict]) -> List[dict]:
    """One point per working-month count (the last one reported wins), ascending (server.py:195-200)."""
    latest = {int(pt["working_months"]): pt for pt in points}
    return [latest[m] for m in sorted(latest)]


def _band_payload(bands: Optional[pd.DataFrame], samples, years: List[float]) -> Optional[dict]:
    """Percentile bands + sample paths of one trajectory family (server.py:203-228)."""
    if bands is None or bands.empty:
        return None
    if len(bands) != len(years):
        raise ValueError("Trajectory time-point count does not match trajectory data "
                         f"({len(years)} != {len(bands)}).")
    return {
        "years": years,
        "percentiles": {_pct_key(q): [_r2(v) for v in bands[q]] for q in bands.columns},
        "sample_paths": [[_r2(v) for v in path] for path in samples] if samples else [],
    }


def _reference_lines(config: Config, wm: int) -> List[dict]:
    """Vertical markers of the plots: retirement start and the first payment of every income stream that
    pays anything (server.py:480-501)."""
    t_ret = wm / MONTHS_PER_YEAR
    lines = [{"name": "Retirement Starts", "year": t_ret}]
    for s in config.other_income_streams or []:
        if s.monthly_amount_today <= SMALL_EPSILON or s.duration_years == 0:
            continue
        first = stream_payment_start_month_index(config.current_age, wm, s.start_at_age)
        lines.append({"name": s.name, "year": round(t_ret + first / MONTHS_PER_YEAR, 3)})
    return lines


def _withdrawal_rate_payload(wr_bands: Optional[pd.DataFrame], counts, wm: int, total_paths: int) -> Optional[dict]:
    if wr_bands is None or wr_bands.empty:
        return None
    t_ret = wm / MONTHS_PER_YEAR
    cols: Dict[str, List[Optional[float]]] = {}
    for q in wr_bands.columns:
        cols[_pct_key(q)] = [None if (v is None or (isinstance(v, float) and math.isnan(v))) else round(float(v), 3)
                             for v in wr_bands[q]]
    return {"years": [t_ret + y for y in range(len(wr_bands))], "percentiles": cols,
            "observation_counts": counts or [], "total_paths": int(total_paths)}


def _summary_block(config: Config, wm: int, estimated: bool, success_pct: float, median_start: float,
                   median_final_ok: float, swr: float, final_pcts: Dict[str, float]) -> dict:
    return {
        "required_working_months": wm,
        "required_working_years": round(wm / MONTHS_PER_YEAR, 1),
        "working_period_is_estimate": estimated,
        "retirement_age": round(retirement_age(config.current_age, wm), 1),
        "success_probability": round(float(success_pct), 2),
        "target_probability": config.target_probability,
        "median_start_balance": round(median_start, 2),
        "median_final_balance_successful": round(median_final_ok, 2),
        "swr": _finite_or_none(swr),
        "final_balance_percentiles": final_pcts,
    }


def assemble_result(config: Config, required_w_months: int, outputs, search_curve: Optional[List[dict]] = None) -> dict:
    """The response document (``SimulationResponse``) from the 7-tuple of run_monte_carlo_simulations —
    everything ``_build_result`` does after the final run (server.py:436-562)."""
    summary, bands, samples, wr_bands, real_bands, real_samples, wr_counts = outputs
    if summary.empty:
        raise ValueError(f"Simulation for '{config.Nickname}' yielded no results.")
    wm = required_w_months
    ok = summary["Success"].astype(bool) if "Success" in summary.columns else summary["Final Balance"] > SMALL_EPSILON
    finals_ok = summary.loc[ok, "Final Balance"]
    final_q = summary["Final Balance"].quantile(list(FINAL_BALANCE_PERCENTILES))
    years = trajectory_time_points(wm, config.retirement_years)
    doc = {
        "scenario": config.Nickname,
        "summary": _summary_block(
            config, wm, bool(search_curve), ok.mean() * 100.0, float(summary["Start Balance"].median()),
            float(finals_ok.median()) if not finals_ok.empty else 0.0, median_first_year_withdrawal_rate(summary),
            {_pct_key(q): round(max(0.0, float(v)), 2) for q, v in final_q.items()}),
        "trajectory": _band_payload(bands, samples, years),
        "trajectory_real": _band_payload(real_bands, real_samples, years),
        "withdrawal_rate": _withdrawal_rate_payload(wr_bands, wr_counts, wm, len(summary)),
        "search_curve": None,
        "ruin_histogram": None,
        "histogram": {
            "final_balances": [_r2(v) for v in summary["Final Balance"]],
            "start_balances": [_r2(v) for v in summary["Start Balance"]],
            "success_flags": [bool(v) for v in summary["Success"]],
        },
        "reference_lines": _reference_lines(config, wm),
    }
    if search_curve:
        doc["search_curve"] = {"points": dedupe_search_curve(search_curve), "target_probability": config.target_probability,
                               "selected_working_months": wm}
    if "YearsToRuin" in summary.columns:
        ruined = summary.loc[~ok, "YearsToRuin"].dropna()
        doc["ruin_histogram"] = {"years_to_ruin": [round(float(v), 1) for v in ruined], "failure_count": int(len(ruined)),
                                 "total_paths": int(len(summary))}
    return doc


def build_result(config: Config, simulator: RetirementMonteCarloSimulator, required_w_months: int,
                 search_curve: Optional[List[dict]] = None) -> dict:
    """``_build_result`` (server.py:416-562): final run on the simulator's active stream + document."""
    outputs = simulator.run_monte_carlo_simulations(working_months=required_w_months,
                                                    num_simulations=config.num_simulations_main)
    return assemble_result(config, required_w_months, outputs, search_curve)


# ---------------------------------------------------------------------------------------------------
# large batches: the same document from device-side aggregates
# ---------------------------------------------------------------------------------------------------
def compact_result(config: Config, simulator: RetirementMonteCarloSimulator, required_w_months: int,
                   search_curve: Optional[List[dict]] = None, num_simulations: Optional[int] = None,
                   n_bins: int = 60) -> dict:
    """The response document for batches much larger than the UI's (10^6..10^8 paths): every number of
    ``summary``, the trajectory / withdrawal-rate bands and the sample paths are the SAME values
    ``build_result`` reports (same kernels, same interpolation, same rounding), but they are computed on
    the device — success share from the kernel's counters, medians and final-balance percentiles by the
    radix select over ``mcr_summary_stat_rows`` — and the three per-path lists of ``histogram`` (and the
    per-path ``years_to_ruin`` list) are left empty; ``histogram_binned`` / ``ruin_histogram.bins`` carry
    the binned equivalents (``np.histogram`` semantics; K1's ruin-year counters).  Under a process group the
    path range is sharded over the ranks (below)."""
    wm = int(required_w_months)
    n = int(config.num_simulations_main if num_simulations is None else num_simulations)
    if n <= 0:
        raise ValueError(f"Simulation for '{config.Nickname}' yielded no results.")
    dev = simulator._local_device()
    # One process per GPU (torch.distributed initialised): every rank simulates its shard of the global path range and
    # keeps it in its own HBM; what crosses ranks is the counter block, the digit histograms of the exact distributed
    # quantiles, the histogram range / bins and the five sampled columns.  Every rank returns the same document, equal
    # to the single-GPU one (the Philox counter carries the global path index).
    sharded = D.is_active()
    if sharded:
        import torch.distributed as dist

        world = dist.get_world_size()
        if D.shard_range(n, world - 1, world)[1] <= 0:   # (ceil(n / world)-sized shards: the last ranks go empty for tiny n)
            raise ValueError(f"compact_result under a process group: {n} paths leave a rank of {world} without a shard; "
                             "run so small a batch inside distributed.local_only() instead")
        begin, count = D.shard_range(n, dist.get_rank(), world)
    else:
        begin, count = 0, n
    batch = E.DeviceBatch(simulator._current_params(), wm, max(count, 1), want="full", device=dev)
    if count > 0:
        batch.launch(simulator._batch_rng(n), simulator._stream_id, begin, count)
    # host-only work under the asynchronous launch: RandomState.choice(n, 5, replace=False) permutes all n indices
    # (7 ms at 1e6 paths, 0.7 s at 1e8 — as long as the kernel); None (logged) where the reference returns no samples
    picked = simulator._sample_columns(n)
    reduced = batch.reduce_vec.clone()
    if sharded:
        if count == 0:
            reduced.zero_()
        reduced = reduced.to(D._comm_device())
        D.all_reduce_sum_(reduced)
    reduced = reduced.cpu().numpy()
    ok_count = int(reduced[0])
    # rows: start | final | final of successful paths | first-year real withdrawal rate (%), NaN = not in the cohort
    stat_rows = A.summary_stat_rows(batch, count)
    if sharded:
        sq, _ = D.sharded_row_quantiles(stat_rows, count, FINAL_BALANCE_PERCENTILES)
        traj_q, real_q, wr_q, wr_counts = D.sharded_band_quantiles(batch, count)
    else:
        sq, _ = A.row_quantiles(stat_rows, n, FINAL_BALANCE_PERCENTILES)
        traj_q, real_q, wr_q, wr_counts = A.band_quantiles(batch, n)
    median = FINAL_BALANCE_PERCENTILES.index(0.50)
    med_final_ok = float(sq[2, median])
    swr = float(sq[3, median])
    qcols = pd.Index(list(A.TRAJECTORY_QUANTILES), dtype="float64")
    samples = real_samples = None
    from .simulation import _gather_columns

    # (`picked` is None on every rank or on none — same seed, same n; the collective below runs unconditionally, on a zero
    #  buffer when sampling failed, so the ranks cannot part ways here)
    k = min(n, 5)
    both = np.zeros((2, k, batch.sizes.trajectory_len))
    if picked is not None:
        mine = [(j, int(g) - begin) for j, g in enumerate(picked) if begin <= g < begin + count]
        if mine:
            cols = [c for _, c in mine]
            both[0, [j for j, _ in mine]] = _gather_columns(batch.trajectory, cols)
            both[1, [j for j, _ in mine]] = _gather_columns(batch.real_trajectory, cols)
    if sharded:   # owner ranks fill their columns, the rest stays 0: a sum is the gather
        import torch

        buf = torch.as_tensor(both, device=D._comm_device())
        D.all_reduce_sum_(buf)
        both = buf.cpu().numpy()
    if picked is not None:
        samples, real_samples = both[0].tolist(), both[1].tolist()
    years = trajectory_time_points(wm, config.retirement_years)
    bins, edges = A.success_histogram(batch.summary["final_balance"][:count], batch.success[:count], n_bins,
                                      reduce_range=D.all_reduce_minmax_ if sharded else None,
                                      reduce_bins=D.all_reduce_sum_ if sharded else None)
    ruin_bins = reduced[N_COUNTERS + config.retirement_years:]
    doc = {
        "scenario": config.Nickname,
        "summary": _summary_block(
            config, wm, bool(search_curve), float(np.float64(ok_count) / np.float64(n) * 100.0), float(sq[0, median]),
            med_final_ok if ok_count else 0.0, swr,
            {_pct_key(q): round(max(0.0, float(v)), 2) for q, v in zip(FINAL_BALANCE_PERCENTILES, sq[1])}),
        "trajectory": _band_payload(pd.DataFrame(traj_q, columns=qcols), samples, years),
        "trajectory_real": _band_payload(pd.DataFrame(real_q, columns=qcols), real_samples, years),
        "withdrawal_rate": _withdrawal_rate_payload(
            pd.DataFrame(wr_q, columns=pd.Index(list(A.WR_QUANTILES), dtype="float64")),
            [int(v) for v in wr_counts.tolist()], wm, n),
        "search_curve": None,
        "ruin_histogram": {"years_to_ruin": [], "failure_count": int(ruin_bins.sum()), "total_paths": n,
                           "bins": [int(v) for v in ruin_bins.tolist()]},
        "histogram": {"final_balances": [], "start_balances": [], "success_flags": []},
        "histogram_binned": {"edges": [float(e) for e in edges], "success_counts": [int(b) for b in bins.tolist()],
                             "successful_paths": ok_count, "total_paths": n},
        "reference_lines": _reference_lines(config, wm),
    }
    if search_curve:
        doc["search_curve"] = {"points": dedupe_search_curve(search_curve), "target_probability": config.target_probability,
                               "selected_working_months": wm}
    return doc


# ---------------------------------------------------------------------------------------------------
# the run flow (search -> final run -> document) with the reference's progress events
# ---------------------------------------------------------------------------------------------------
def run_scenario(config: Config, working_months_override: Optional[int] = None,
                 emit: Optional[Callable[[dict], None]] = None, result_builder: Callable[..., dict] = build_result,
                 **simulator_kwargs: Any) -> Optional[dict]:
    """Search (unless overridden), final run, document.

    Without ``emit`` this is ``_run_simulation`` (server.py:231-266): returns the document, raises
    ``ValueError`` when the target cannot be met.  With ``emit`` it is the body of the SSE endpoint
    (server.py:341-394): the same ``phase`` / ``search_iter`` / ``search_refining`` / ``search_complete`` /
    ``result`` / ``error`` events, in the same order, go to ``emit`` (called on this thread), and an
    unreachable target is reported as an ``error`` event (returns None) instead of raised."""
    if emit is None:
        return _run_flow(config, working_months_override, None, result_builder, simulator_kwargs)
    try:
        return _run_flow(config, working_months_override, emit, result_builder, simulator_kwargs)
    except Exception as exc:  # the SSE body reports every failure as an event (server.py:395-396)
        emit({"type": "error", "message": str(exc)})
        return None


def _run_flow(config, working_months_override, emit, result_builder, simulator_kwargs) -> Optional[dict]:
    simulator = RetirementMonteCarloSimulator(config, **simulator_kwargs)
    curve: List[dict] = []
    if working_months_override is not None:
        wm = working_months_override
        if emit:
            emit({"type": "phase", "phase": "final_sim", "message": f"Using override: {wm} months"})
        else:
            logger.info(f"Using working-months override: {wm} ({wm / MONTHS_PER_YEAR:.1f} yrs)")
    else:
        if emit:
            emit({"type": "phase", "phase": "search", "message": "Estimating required working months…"})
        else:
            logger.info(f"Estimating required working months for '{config.Nickname}'")
        wm, achieved, curve = simulator.find_minimum_working_months(verbose=True, progress_callback=emit)
        if wm == -1:
            if emit:
                emit({"type": "error",
                      "message": f"Target {config.target_probability:.1f}% not met. Highest: {achieved:.1f}%"})
                return None
            raise ValueError(f"Target probability of {config.target_probability:.2f}% could not be met. "
                             f"Highest achieved: {achieved:.2f}%")
        if emit:
            emit({"type": "search_complete", "working_months": wm, "working_years": round(wm / MONTHS_PER_YEAR, 1),
                  "probability": round(achieved, 2)})
    if emit:
        emit({"type": "phase", "phase": "final_sim",
              "message": f"Running {config.num_simulations_main} final simulations with {wm} working months…"})
    else:
        logger.info(f"Running final simulation for '{config.Nickname}' "
                    f"({config.num_simulations_main} sims, {wm} working months)")
    simulator.use_final_seeds()
    doc = result_builder(config, simulator, wm, search_curve=curve)
    if emit:
        emit({"type": "result", "data": doc})
    return doc
