"""bench.py as the driver runs it: `python bench.py --gpus N` must launch its own N ranks.
The box has one GPU, so the 2-rank rehearsal uses the gloo backend with both ranks on the card; the RCCL run on
8 GPUs is the driver's."""

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


def test_bench_self_launches_two_ranks():
    n, steps = 200_000, 3
    out = _run("--gpus", "2", "--backend", "gloo", "--steps", str(steps), "--warmup", "1", "--paths", str(n), "--s60-paths", "300001")
    assert out["n_gpus"] == 2 and out["steps"] == steps and out["scaling"] == "weak" and out["unit"] == "paths/s"
    assert "all-reduce" in out["config"]["parallelism"] and "x2" in out["config"]["parallelism"]
    assert out["paths_counted"] == 2 * steps * n            # the exchange sums every rank's steps exactly once
    assert 0.9 < out["success_probability"] <= 1.0
    assert out["value"] == pytest.approx(2 * steps * n / (out["ms_per_step"] * 1e-3 * steps), rel=1e-6)
    assert out["roofline"]["bound"] == "valu_fp64" and 0.0 < out["roofline"]["frac"] < 1.0
    s60 = out["s60"]
    assert "error" not in s60, s60
    assert s60["paths_counted"] == 300001 and s60["n_gpus"] == 2 and 0.5 < s60["success_probability"] < 1.0
    assert s60["hist_total"] == round(s60["success_probability"] * 300001)


def test_bench_single_gpu_line_has_the_contract_keys():
    out = _run("--steps", "3", "--warmup", "1", "--paths", "200000", "--aux-paths", "2200000", "--s60-paths", "400000",
               "--cpu-threads", "4", "--cpu-paths-per-thread", "2000")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "hbm_kernels", "s60"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["dtype"] == "f64" and out["vs_baseline"] is None and out["paths_counted"] == 3 * 200000
    assert out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["cores"] == 4
    assert set(out["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert out["s60"]["paths_counted"] == 400000 and out["hbm_kernels"]["K3_row_quantiles"]["rows"] == 136
    acc = out["accuracy_10k"]      # BASELINE's "success-prob abs error vs CPU ref, 10k-path config"
    assert "error" not in acc, acc
    assert acc["abs_error"] <= 1e-4 and acc["flipped_success_flags"] == 0
