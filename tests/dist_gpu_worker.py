"""Worker for tests/test_distributed_gpu.py: one rank of a gloo group; all ranks share the box's GPU.
Checks the product's sharded paths on real kernels: distributed exact quantiles vs pandas on the
unsharded slab, and run_sharded_bands vs a single-process run of the whole path range."""

from __future__ import annotations

import json
import os
import sys

import numpy as np
import pandas as pd
import torch
import torch.distributed as dist

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from monte_carlo_retirement_amd import Config, params_from_config  # noqa: E402
from monte_carlo_retirement_amd import aggregation as A  # noqa: E402
from monte_carlo_retirement_amd import distributed as D  # noqa: E402
from monte_carlo_retirement_amd import engine as E  # noqa: E402


def main():
    out_path = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    res = {"rank": rank}

    # ---- (1) distributed quantiles on a synthetic slab (same data on every rank, each takes its shard)
    rng = np.random.default_rng(77)
    n_total = 60_001
    slab = np.empty((6, n_total))
    slab[0] = rng.lognormal(13, 1, n_total)
    slab[1] = np.where(rng.random(n_total) < 0.4, np.nan, rng.normal(5, 2, n_total))
    slab[2] = np.where(rng.random(n_total) < 0.65, 0.0, rng.lognormal(10, 2, n_total))  # giant tie: overflow path
    slab[3] = 60000.0                                                                     # all equal
    slab[4] = np.nan                                                                      # all NaN
    slab[5] = -rng.lognormal(3, 2, n_total)
    begin, count = D.shard_range(n_total, rank, world)
    stride = (count + 63) // 64 * 64
    local = torch.full((6, max(stride, 64)), 7.0, dtype=torch.float64, device="cuda")
    local[:, :count] = torch.as_tensor(slab[:, begin:begin + count], device="cuda")
    got, counts = D.sharded_row_quantiles(local, count, A.TRAJECTORY_QUANTILES)
    exp = pd.DataFrame(slab.T).quantile(list(A.TRAJECTORY_QUANTILES), axis=0).T.to_numpy()
    res["quantiles_equal"] = bool(np.array_equal(got, exp, equal_nan=True))
    res["counts_equal"] = counts.tolist() == (~np.isnan(slab)).sum(axis=1).tolist()

    # ---- (1b) the same through the sharded BRACKETED route: rows of 2^22 entries in total (the default threshold) and, with
    # the threshold lowered, rows with ties / NaNs / constants that must fall back together on every rank
    n_big = 1 << 22
    rb = np.random.default_rng(5)
    big = np.empty((4, n_big))
    big[0] = rb.lognormal(14, 1.2, n_big)
    big[1] = np.where(rb.random(n_big) < 0.2, np.nan, rb.normal(4.0, 1.5, n_big))
    big[2] = 1.0e6
    big[3] = np.where(rb.random(n_big) < 0.015, 0.0, rb.lognormal(13, 0.8, n_big))
    b0, bc = D.shard_range(n_big, rank, world)
    bstride = (bc + 63) // 64 * 64
    blocal = torch.full((4, bstride), 7.0, dtype=torch.float64, device="cuda")
    blocal[:, :bc] = torch.as_tensor(big[:, b0:b0 + bc], device="cuda")
    gb, cb_ = D.sharded_row_quantiles(blocal, bc, A.TRAJECTORY_QUANTILES)
    res["bracket_route_taken"] = A.last_fallback_rows() == 0
    res["bracket_quantiles_equal"] = bool(all(
        np.array_equal(gb[r], np.quantile(big[r][~np.isnan(big[r])], A.TRAJECTORY_QUANTILES)) for r in range(4)))
    res["bracket_counts_equal"] = cb_.tolist() == (~np.isnan(big)).sum(axis=1).tolist()
    del blocal
    # rows of exactly the DEFAULT threshold (2^18 entries in total, no override): the bracketed route must be the one taken
    n_thr = A._SHARDED_BRACKET_MIN_TOTAL
    rt = np.random.default_rng(11)
    thr = np.stack([rt.lognormal(12, 1.5, n_thr), np.where(rt.random(n_thr) < 0.3, np.nan, rt.normal(0.0, 3.0, n_thr))])
    t0, tc = D.shard_range(n_thr, rank, world)
    tlocal = torch.full((2, (tc + 63) // 64 * 64), 7.0, dtype=torch.float64, device="cuda")
    tlocal[:, :tc] = torch.as_tensor(thr[:, t0:t0 + tc], device="cuda")
    gt, ct = D.sharded_row_quantiles(tlocal, tc, A.TRAJECTORY_QUANTILES)
    res["bracket_at_threshold"] = bool(A.last_fallback_rows() >= 0 and ct.tolist() == (~np.isnan(thr)).sum(axis=1).tolist() and all(
        np.array_equal(gt[r], np.quantile(thr[r][~np.isnan(thr[r])], A.TRAJECTORY_QUANTILES)) for r in range(2)))
    gt2, _ = D.sharded_row_quantiles(tlocal[:, :tc - 40], tc - 40, A.TRAJECTORY_QUANTILES)      # just below it: the stepwise radix route
    res["radix_below_threshold"] = bool(all(
        np.array_equal(gt2[r], np.nanquantile(np.concatenate([thr[r, D.shard_range(n_thr, k, world)[0]:][:D.shard_range(n_thr, k, world)[1] - 40]
                                                              for k in range(world)]), A.TRAJECTORY_QUANTILES)) for r in range(2)))
    del tlocal
    os.environ["MCR_RQ_SHARDED_BRACKET_MIN_N"] = "100000"
    n_mid = 400_003
    rm = np.random.default_rng(9)
    mid = np.empty((7, n_mid))
    mid[0] = rm.lognormal(13, 1, n_mid)
    mid[1] = np.where(rm.random(n_mid) < 0.4, np.nan, rm.normal(5, 2, n_mid))
    mid[2] = np.where(rm.random(n_mid) < 0.65, 0.0, rm.lognormal(10, 2, n_mid))    # giant tie + tail
    mid[3] = 60000.0
    mid[4] = np.nan
    mid[5] = rm.choice([1.0, 2.0, 3.0, 5.0, 8.0], n_mid)                            # few distinct values
    mid[6] = np.arange(n_mid, dtype=float)                                          # sorted: the prefix samples are useless
    m0, mc = D.shard_range(n_mid, rank, world)
    mstride = (mc + 63) // 64 * 64
    mlocal = torch.full((7, mstride), 7.0, dtype=torch.float64, device="cuda")
    mlocal[:, :mc] = torch.as_tensor(mid[:, m0:m0 + mc], device="cuda")
    gm, cm = D.sharded_row_quantiles(mlocal, mc, A.TRAJECTORY_QUANTILES)
    res["bracket_fallback_rows"] = A.last_fallback_rows()
    expm = pd.DataFrame(mid.T).quantile(list(A.TRAJECTORY_QUANTILES), axis=0).T.to_numpy()
    res["bracket_mid_equal"] = bool(np.array_equal(gm, expm, equal_nan=True))
    res["bracket_mid_counts"] = cm.tolist() == (~np.isnan(mid)).sum(axis=1).tolist()
    del os.environ["MCR_RQ_SHARDED_BRACKET_MIN_N"]

    # ---- (1c) randomised shapes through the sharded routes (every rank draws the same slab from the same seed and takes
    # its shard): total length from just above the bracketed route's minimum shard to ~1.2e6, 1-8 rows from the menu of
    # tests/test_simulator_gpu.py::_random_row, 1-15 quantiles; must equal pandas on the whole slab bit for bit.
    # MCR_SHARD_FUZZ_SEED / MCR_SHARD_FUZZ_ROUNDS lengthen it for soaks by hand.
    from test_simulator_gpu import _random_row

    os.environ["MCR_RQ_SHARDED_BRACKET_MIN_N"] = "1"
    fr = np.random.default_rng(int(os.environ.get("MCR_SHARD_FUZZ_SEED", "20260104")))
    fuzz_bad, fuzz_bracket_calls = [], 0
    for it in range(int(os.environ.get("MCR_SHARD_FUZZ_ROUNDS", "6"))):
        n_f = int(fr.choice([world * 65536, world * 65536 + 1, int(fr.integers(world * 65536, 1_200_000)), int(fr.integers(1000, world * 65536))]))
        rows_f = int(fr.integers(1, 9))
        slab_f = np.empty((rows_f, n_f))
        for r in range(rows_f):
            slab_f[r] = _random_row(fr, n_f)
        k = int(fr.integers(1, 16))
        qs = tuple(sorted(set(np.round(fr.uniform(0, 1, k), int(fr.integers(1, 6))).tolist() + ([0.0] if fr.random() < 0.2 else []) + ([1.0] if fr.random() < 0.2 else []))))[:15]
        f0, fc = D.shard_range(n_f, rank, world)
        fstride = max(64, (fc + int(fr.choice([0, 1, 63]))))
        flocal = torch.full((rows_f, fstride), -7.0, dtype=torch.float64, device="cuda")
        flocal[:, :fc] = torch.as_tensor(slab_f[:, f0:f0 + fc], device="cuda")
        gf, cf = D.sharded_row_quantiles(flocal, fc, qs)
        fuzz_bracket_calls += A.last_fallback_rows() >= 0
        expf = pd.DataFrame(slab_f.T).quantile(list(qs), axis=0).T.to_numpy()
        if not np.array_equal(gf, expf, equal_nan=True) or cf.tolist() != (~np.isnan(slab_f)).sum(axis=1).tolist():
            fuzz_bad.append((it, n_f, rows_f, qs))
        del flocal
    del os.environ["MCR_RQ_SHARDED_BRACKET_MIN_N"]
    res["fuzz_bad"] = [str(b) for b in fuzz_bad]
    res["fuzz_bracket_calls"] = int(fuzz_bracket_calls)

    # ---- (2) sharded bands vs one process over the whole range
    with open(os.path.join(REPO, "scenarios", "jorge.json")) as fh:
        cfg = Config(**dict(json.load(fh), equity_inflation_correlation=0.3, initial_balance=20000.0, monthly_contribution=3000.0))
    p = params_from_config(cfg)
    n_paths, wm = 5003, 40
    sh = D.run_sharded_bands(p, 2024, 1, n_paths, wm, n_bins=60)
    whole = E.DeviceBatch(p, wm, n_paths, want="full", device=0)
    whole.launch(2024, 1, 0)
    tq, _ = A.row_quantiles(whole.trajectory, n_paths, A.TRAJECTORY_QUANTILES)
    rq, _ = A.row_quantiles(whole.real_trajectory, n_paths, A.TRAJECTORY_QUANTILES)
    wq, wc = A.row_quantiles(whole.withdrawal_rate_trajectory, n_paths, A.WR_QUANTILES)
    hb, he = A.success_histogram(whole.summary["final_balance"], whole.success, 60)
    res["bands_equal"] = bool(np.array_equal(sh["trajectory_q"], tq, equal_nan=True) and np.array_equal(sh["real_trajectory_q"], rq, equal_nan=True)
                              and np.array_equal(sh["wr_q"], wq, equal_nan=True) and sh["wr_counts"].tolist() == wc.tolist())
    res["hist_equal"] = bool(sh["hist_bins"].tolist() == hb.tolist() and np.array_equal(sh["hist_edges"], he))
    res["counts_ok"] = bool(sh["counts"].success == int(whole.counters[0].item()) and sh["counts"].paths == n_paths
                            and sh["counts"].ruin_year_bins.tolist() == whole.ruin_year_bins.cpu().tolist())
    res["success"] = sh["counts"].success
    # ---- (2b) counts + histogram only (no trajectories): data-ranged and fixed-range bins
    hh = D.run_sharded_histogram(p, 2024, 1, n_paths, wm, n_bins=60)
    res["hist_only_equal"] = bool(hh["hist_bins"].tolist() == hb.tolist() and np.array_equal(hh["hist_edges"], he)
                                  and hh["counts"].success == sh["counts"].success and hh["counts"].paths == n_paths)
    fixed = (0.0, float(he[-1]) * 1.5)
    hf = D.run_sharded_histogram(p, 2024, 1, n_paths, wm, n_bins=60, value_range=fixed)
    hb_f, he_f = A.success_histogram(whole.summary["final_balance"], whole.success, 60, value_range=fixed)
    res["hist_fixed_equal"] = bool(hf["hist_bins"].tolist() == hb_f.tolist() and np.array_equal(hf["hist_edges"], he_f))
    # ---- (3) the drop-in class, transparently sharded: search probe + final run
    from monte_carlo_retirement_amd.simulation import RetirementMonteCarloSimulator

    sim = RetirementMonteCarloSimulator(cfg, main_seed_override=2024)
    sim.shard_min_paths = 0  # force sharding even for this small batch
    sim.use_final_seeds()
    summary, tdf, samples, wdf, rdf, rsamples, wcounts = sim.run_monte_carlo_simulations(wm, n_paths)
    res["class_summary_equal"] = bool(
        np.array_equal(summary["Success"].to_numpy(), whole.success.cpu().numpy().astype(bool))
        and np.array_equal(summary["Final Balance"].to_numpy(), whole.summary["final_balance"].cpu().numpy())
        and np.array_equal(summary["YearsToRuin"].to_numpy(), whole.summary["years_to_ruin"].cpu().numpy(), equal_nan=True))
    res["class_bands_equal"] = bool(np.array_equal(tdf.to_numpy(), tq, equal_nan=True) and np.array_equal(rdf.to_numpy(), rq, equal_nan=True)
                                    and np.array_equal(wdf.to_numpy(), wq, equal_nan=True) and wcounts == wc.tolist())
    picked = np.random.RandomState(2024).choice(n_paths, size=5, replace=False)
    res["class_samples_equal"] = bool(np.array_equal(np.array(samples), whole.trajectory[:, :n_paths].cpu().numpy()[:, picked].T)
                                      and np.array_equal(np.array(rsamples), whole.real_trajectory[:, :n_paths].cpu().numpy()[:, picked].T))
    sim.use_search_seeds()
    one = E.DeviceBatch(p, 25, 3000, want="count", device=0)
    one.launch(2024, 0, 0)
    expect = float(np.float64(int(one.counters[0].item())) / np.float64(3000) * 100.0)
    res["class_probe_equal"] = sim._probe_success_probability(25, 3000) == expect
    sim.shard_min_paths = 1_000_000  # default: small batches are replicated on every rank, no communication
    res["class_probe_equal"] = res["class_probe_equal"] and sim._probe_success_probability(25, 3000) == expect
    # several candidates in one call: split by candidate (small batch) and by path range (forced) agree with
    # the single-GPU counts
    cands = [0, 13, 25, 31, 60]
    solo = E.probe_months(p, 2024, 0, 0, 3000, cands, device=0)[:, 0].cpu().numpy()
    expect_many = {m: float(np.float64(int(solo[i])) / np.float64(3000) * 100.0) for i, m in enumerate(cands)}
    res["class_probe_many_by_candidate"] = sim._probe_many(cands, 3000) == expect_many
    sim.shard_min_paths = 0
    res["class_probe_many_by_range"] = sim._probe_many(cands, 3000) == expect_many
    sim.shard_min_paths = 1_000_000
    res["class_speculation_slots"] = sim._speculation_slots(50_000) == world and sim._speculation_slots(10**7) == 1
    # the whole search under the group (candidates split across ranks, bracket/bisection points evaluated
    # ahead): every rank must replay the reference's recorded search exactly
    g = json.load(open(os.path.join(REPO, "tests", "golden", "search.json")))[0]
    s_sim = RetirementMonteCarloSimulator(Config(**g["cfg"]), main_seed_override=g["seed"])
    s_events, s_calls = [], []
    s_many = s_sim._probe_many
    s_sim._probe_many = lambda months, n: (s_calls.append(len(months)), s_many(months, n))[1]
    s_res = s_sim.find_minimum_working_months(verbose=False, progress_callback=s_events.append)
    res["class_search_replays_reference"] = bool(
        s_res[0] == g["months"] and s_res[1] == g["probability"] and s_res[2] == g["search_curve"] and s_events == g["events"])
    res["class_search_batched"] = bool(len(s_calls) < len(g["search_curve"]))
    # large-n guard: above gather_all_max_paths the per-path frame goes to rank 0 only; the rest of the tuple is unchanged
    sim.shard_min_paths = 0
    sim.use_final_seeds()
    sim.gather_all_max_paths = 1000
    guarded = sim.run_monte_carlo_simulations(wm, n_paths)
    sim.gather_all_max_paths = 20_000_000
    res["guard_frame_rows"] = int(len(guarded[0]))
    res["guard_frame_columns_ok"] = list(guarded[0].columns) == list(summary.columns) and str(guarded[0]["Success"].dtype) == "bool"
    res["guard_rank0_frame_equal"] = bool(rank != 0 or guarded[0].equals(summary))
    res["guard_rest_equal"] = bool(np.array_equal(guarded[1].to_numpy(), tdf.to_numpy(), equal_nan=True) and guarded[2] == samples
                                   and np.array_equal(guarded[3].to_numpy(), wdf.to_numpy(), equal_nan=True) and guarded[5] == rsamples
                                   and guarded[6] == wcounts)
    res["guard_success_probability"] = float(sim._success_probability(guarded[0])) if rank == 0 else None
    sim.use_search_seeds()
    sim.shard_min_paths = 1_000_000
    rep = sim.run_monte_carlo_simulations(wm, 700)
    sim2 = RetirementMonteCarloSimulator(cfg, main_seed_override=2024)
    sim2.shard_min_paths = 0
    sim2.use_search_seeds()
    shd = sim2.run_monte_carlo_simulations(wm, 700)
    res["class_replicated_equals_sharded"] = bool(rep[0].equals(shd[0]) and np.array_equal(rep[1].to_numpy(), shd[1].to_numpy(), equal_nan=True)
                                                  and rep[6] == shd[6] and rep[2] == shd[2])
    # ---- (3b) no seed configured: rank 0's time-derived seed is used by every rank, so the shards belong to ONE run and
    # the sampled columns are the ones that run's seed picks (advisor r1)
    from monte_carlo_retirement_amd.simulation import _fold_seed_u64

    ns = RetirementMonteCarloSimulator(Config(**dict(cfg.model_dump(by_alias=True), seed=None)))
    ns.shard_min_paths = 0
    ns.use_final_seeds()
    ns_out = ns.run_monte_carlo_simulations(wm, 2000)
    ns_whole = E.DeviceBatch(p, wm, 2000, want="full", device=0)
    ns_whole.launch(_fold_seed_u64(ns.main_seed), 1, 0)
    ns_pick = np.random.RandomState(ns.main_seed).choice(2000, size=5, replace=False)
    res["unseeded_seed"] = int(ns.main_seed)
    res["unseeded_samples_equal"] = bool(
        np.array_equal(np.array(ns_out[2]), ns_whole.trajectory[:, :2000].cpu().numpy()[:, ns_pick].T)
        and np.array_equal(ns_out[0]["Final Balance"].to_numpy(), ns_whole.summary["final_balance"].cpu().numpy()))

    # ---- (4) the compact response document of a large batch, from sharded batches: equal to one process's document
    from monte_carlo_retirement_amd import results as R

    c_sim = RetirementMonteCarloSimulator(cfg, main_seed_override=2024)
    c_sim.use_final_seeds()
    doc = R.compact_result(cfg, c_sim, wm, num_simulations=n_paths)
    res["compact_doc"] = doc
    with open(f"{out_path}.{rank}", "w") as fh:
        json.dump(res, fh)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
