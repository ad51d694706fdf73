"""bench.py as the driver runs it: `python bench.py --gpus N` must launch its own N ranks.
The box has one GPU, so the rehearsals use the gloo backend with all ranks on the card: 2 ranks, and 5 — this pool lets one
job hold a card open from at most 6 processes, the test runner being one of them (its process guard kills a job with more:
the world size 8 of the driver's scaling run cannot be rehearsed on a one-GPU box; the 8-rank logic — ragged shards, candidates split 8 ways with ranks sitting a round
out — runs on the CPU with gloo in tests/test_distributed_cpu.py).  The RCCL run on 8 GPUs is the driver's."""

from __future__ import annotations

import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def _run(*args, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("world,n,s60_paths", [(2, 200_000, 300_001), (5, 50_000, 800_001)])
def test_bench_self_launches_its_ranks(world, n, s60_paths):
    steps = 3
    out = _run("--gpus", str(world), "--backend", "gloo", "--steps", str(steps), "--warmup", "1", "--paths", str(n), "--s60-paths", str(s60_paths),
               "--cpu-threads", "2", "--cpu-paths-per-thread", "4000", "--cpu-single-thread-paths", "4000", "--cpu-all-cores-seconds", "0",
               timeout=900)
    assert out["n_gpus"] == world and out["steps"] == steps and out["scaling"] == "weak" and out["unit"] == "paths/s"
    assert "all-reduce" in out["config"]["parallelism"] and f"x{world}" in out["config"]["parallelism"]
    assert out["paths_counted"] == world * steps * n        # the exchange sums every rank's steps exactly once
    assert 0.9 < out["success_probability"] <= 1.0
    assert out["value"] == pytest.approx(world * steps * n / (out["ms_per_step"] * 1e-3 * steps), rel=1e-6)
    assert out["roofline"]["bound"] == "valu_fp64" and 0.0 < out["roofline"]["frac"] < 1.0
    # what the process group contained: every rank took part in a collective, each reports its device
    c = out["config"]
    assert c["ranks_seen"] == world and c["backend"] == "gloo" and [d["rank"] for d in c["devices"]] == list(range(world))
    assert all(d["gcn_arch"].startswith("gfx950") and d["compute_units"] == 256 for d in c["devices"])
    for key in ("s60", "s60_data_ranged"):
        s60 = out[key]
        assert "error" not in s60, s60
        assert s60["paths_counted"] == s60_paths and s60["n_gpus"] == world and 0.5 < s60["success_probability"] < 1.0   # (ragged last shard)
        assert s60["hist_total"] + s60["hist_outside_edges"] == round(s60["success_probability"] * s60_paths)
    assert out["s60"]["exchange"].startswith("1 all-reduce(sum)")        # fixed edges: ONE collective
    assert out["s60_data_ranged"]["hist_outside_edges"] == 0 and "min,max" in out["s60_data_ranged"]["exchange"]
    assert out["s60"]["success_probability"] == out["s60_data_ranged"]["success_probability"]
    # BASELINE configs[4]: the candidate-split search replays the single-GPU search probe for probe
    sr = out["search"]
    assert "error" not in sr, sr
    assert sr["n_gpus"] == world and sr["equals_single_gpu_search"] is True and "by candidate" in sr["probe_split"]
    assert sr["months_found"] == 232 and sr["probes"] == 17 and sr["paths_per_probe"] == 50_000
    assert sr["final_run_paths"] == 1_000_000 and 90.0 < sr["final_success_probability_pct"] < 100.0
    # rank 0 times the CPU baseline under N > 1 too
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["few_threads"]["cores"] == 2 and cb["host_cores"] >= 1 and cb["single_thread"]["cores"] == 1


def test_bench_single_gpu_line_has_the_contract_keys():
    out = _run("--steps", "3", "--warmup", "1", "--paths", "200000", "--aux-paths", "2200000", "--s60-paths", "400000",
               "--cpu-threads", "4", "--cpu-paths-per-thread", "2000", "--cpu-single-thread-paths", "2000", "--cpu-all-cores-seconds", "1.5")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "hbm_kernels", "hbm_kernels_rho0", "numpy_stream", "class_api_1e7",
              "s60", "s60_data_ranged", "search"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["dtype"] == "f64" and out["vs_baseline"] is None and out["paths_counted"] == 3 * 200000
    # the CPU side "single thread and all cores"; `value` is the all-core figure
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["few_threads"]["cores"] == 4 and cb["few_threads"]["value"] > cb["single_thread"]["value"] > 0
    assert cb["host_cores"] >= cb["usable_cores"] >= 4 and isinstance(cb["cpu_model"], str) and cb["cpu_model"]
    assert cb["all_cores"]["cores"] == cb["usable_cores"] and cb["all_cores"]["value"] > 0
    assert cb["value"] == max(cb["all_cores"]["value"], cb["few_threads"]["value"]) > cb["single_thread"]["value"]
    assert "cgroup_cpu_quota_cores" in cb and cb["quota_limited"] == (cb["cgroup_cpu_quota_cores"] is not None and cb["cgroup_cpu_quota_cores"] < cb["usable_cores"])
    # jorge.json with rho = 0.3 AND as shipped (rho = 0); the literal-seed stream has a timed figure
    assert out["hbm_kernels"]["equity_inflation_correlation"] == 0.3 and out["hbm_kernels_rho0"]["equity_inflation_correlation"] == 0.0
    for blk in ("hbm_kernels", "hbm_kernels_rho0"):
        assert "error" not in out[blk], out[blk]
        assert out[blk]["K1_full_output"]["paths_per_s"] > 0 and out[blk]["K3_row_quantiles"]["rows"] == 136
    ns = out["numpy_stream"]
    assert "error" not in ns, ns
    assert ns["paths_counted"] == 200000 and ns["paths_per_s"] > 0 and 0.9 < ns["success_probability"] <= 1.0
    ca = out["class_api_1e7"]
    assert "error" not in ca, ca
    assert ca["paths"] == 2200000 and ca["seconds"] > ca["kernel_seconds"] > 0 and ca["summary_rows"] == 2200000
    assert set(out["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "frac_of_measured_issue_ceiling"}
    assert out["s60"]["paths_counted"] == 400000 and out["hbm_kernels"]["K3_row_quantiles"]["rows"] == 136
    k3 = out["hbm_kernels"]["K3_row_quantiles"]
    assert len(k3["by_allocation_ms"]) == 3 and k3["ms"] == sorted(k3["by_allocation_ms"])[1] and k3["fallback_rows"] == 0
    assert out["s60"]["exchange"] == "none (1 GPU)" and "inside the count-only path kernel" in out["s60"]["workload"]
    assert out["s60"]["success_probability"] == out["s60_data_ranged"]["success_probability"]
    c = out["config"]
    assert c["ranks_seen"] == 1 and c["backend"] is None and len(c["devices"]) == 1 and c["devices"][0]["device_index"] == 0
    # BASELINE configs[4] through the class API.  (The survey's reference run found 233 months on this scenario with its
    # NumPy stream and 300 paths per probe; the engine's stream at 50 000 paths per probe: 232 months, 17 probes.  The
    # search LOGIC is pinned against six searches recorded from the reference in tests/test_simulator_gpu.py.)
    sr = out["search"]
    assert "error" not in sr, sr
    assert sr["months_found"] == 232 and sr["probability_pct"] == pytest.approx(97.184) and sr["probes"] == 17
    assert sr["probe_rounds"] < sr["probes"] <= sr["months_evaluated"] and sr["largest_round"] >= 3   # rounds of up to 3 months at the cost of ~one
    assert sr["search_seconds"] > 0 and sr["ms_per_probe"] == pytest.approx(sr["search_seconds"] / sr["probes"] * 1e3)
    assert sr["final_run_paths"] == 1_000_000 and 97.0 < sr["final_success_probability_pct"] < 99.5
    assert "equals_single_gpu_search" not in sr                       # (only meaningful under a process group)
    acc = out["accuracy_10k"]      # BASELINE's "success-prob abs error vs CPU ref, 10k-path config"
    assert "error" not in acc, acc
    assert acc["abs_error"] <= 1e-4 and acc["flipped_success_flags"] == 0


def test_bench_group_code_path_over_rccl_with_one_rank():
    """The N > 1 code path of bench.py — process group with device_id, asynchronous per-step all-reduce on alternating
    copies, barrier, max-over-ranks timing, the s60 block's barrier/all-reduce, teardown — with the `nccl` (= RCCL)
    backend.  One rank is all a one-GPU box can give RCCL, but it is the real library and the real calls."""
    n, steps = 200_000, 4
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MCR_BENCH_FORCE_GROUP="1", MASTER_PORT=str(29600 + os.getpid() % 300))
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", str(steps), "--warmup", "2", "--paths", str(n),
                        "--s60-paths", "300001", "--no-aux", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["paths_counted"] == steps * n            # the last exchange holds the totals
    assert "all-reduce(sum)" in out["config"]["parallelism"] and "nccl" in out["config"]["parallelism"]
    assert "error" not in out["s60"] and out["s60"]["paths_counted"] == 300001
    assert out["config"]["ranks_seen"] == 1 and out["config"]["backend"] == "nccl"
    assert "error" not in out["search"] and out["search"]["months_found"] == 232


def test_library_collectives_over_rccl_with_one_rank(tmp_path):
    """distributed.py's wrappers (sum / min-max all-reduce, broadcast) and the sharded bracketed quantile route — whose
    reduce callback crosses the C boundary and all-reduces slices of the scratch block — on the `nccl` backend."""
    script = tmp_path / "rccl_one_rank.py"
    script.write_text(
        "import os, sys\n"
        f"sys.path.insert(0, {REPO!r})\n"
        "import numpy as np, torch, torch.distributed as dist\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))\n"
        "from monte_carlo_retirement_amd import aggregation as A, distributed as D\n"
        "t = torch.arange(10, dtype=torch.int64, device='cuda'); D.all_reduce_sum_(t); assert t.tolist() == list(range(10))\n"
        "m = torch.tensor([3.5, -2.0], dtype=torch.float64, device='cuda'); D.all_reduce_minmax_(m); assert m.tolist() == [3.5, -2.0]\n"
        "b = torch.tensor([2**62 + 5], dtype=torch.int64, device='cuda'); dist.broadcast(b, src=0); assert int(b.item()) == 2**62 + 5\n"
        "w = dist.all_reduce(torch.ones(4, dtype=torch.int64, device='cuda'), async_op=True); w.wait()\n"
        "n = 1 << 22\n"
        "rng = np.random.default_rng(4)\n"
        "rows = np.stack([rng.lognormal(12, 1, n), np.where(rng.random(n) < 0.3, np.nan, rng.normal(0, 5, n)), np.full(n, 7.0)])\n"
        "dev = torch.as_tensor(rows, device='cuda')\n"
        "got, counts = A.row_quantiles(dev, n, A.TRAJECTORY_QUANTILES, reduce_counts=D.all_reduce_sum_, n_total=n)\n"
        "assert A.last_fallback_rows() == 0, A.last_fallback_rows()      # the sharded BRACKETED route ran (callback -> RCCL)\n"
        "plain, pc = A.row_quantiles(dev, n, A.TRAJECTORY_QUANTILES)\n"
        "assert np.array_equal(got, plain, equal_nan=True) and counts.tolist() == pc.tolist()\n"
        "dist.barrier(); dist.destroy_process_group(); print('rccl ok')\n")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29300 + os.getpid() % 300))
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_launcher_deadline_kills_hung_ranks_and_reports(tmp_path):
    """launch_ranks cannot wait forever: with a 1-second overall deadline the two children (still importing torch) are
    killed by PID, the exit code is non-zero and the parent says which ranks it killed."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MCR_BENCH_DEADLINE_S"] = "1"
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "200", "--paths", "4000000"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 124, (r.returncode, r.stderr[-1500:])
    assert "overall deadline reached" in r.stderr and "killing ranks [0, 1]" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.parametrize("world", [2, 5])
def test_bench_under_the_drivers_launcher(world):
    """The driver's own command for N > 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W` (ranks from the environment, no self-launch) — here
    with N = 2 and N = 5 on the box's one GPU and the gloo backend."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    port = 29700 + (os.getpid() + world) % 200
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1",
           "--backend", "gloo", "--paths", "200000", "--s60-paths", "300001", "--no-search", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                       # rank 0 prints, rank 1 does not
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["config"]["ranks_seen"] == world and out["paths_counted"] == world * 3 * 200000
    assert out["s60"]["exchange"].startswith("1 all-reduce(sum)") and "search" not in out and "cpu_baseline" not in out
