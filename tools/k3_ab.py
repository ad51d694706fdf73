"""Interleaved A/B timing of K3 (bands of the [136, n] slab, configs[2] shape) under different MCR_RQ_* knobs, all in
ONE process on one box: variants alternate round after round, so clock / thermal drift hits them alike.

    python tools/k3_ab.py 10000000 "" "MCR_RQ_COLLECT_PER_ROW=4" "MCR_RQ_SAMPLE_WGS=680 MCR_RQ_COLLECT_PER_ROW=3"
"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from monte_carlo_retirement_amd import Config, params_from_config
from monte_carlo_retirement_amd import aggregation as A, engine as E

n = int(sys.argv[1])
variants = sys.argv[2:] or [""]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = Config(**dict(json.load(open(os.path.join(root, "scenarios/jorge.json"))), equity_inflation_correlation=0.3, seed=12345))
b = E.DeviceBatch(params_from_config(cfg), 75, n, want="full")
b.launch(12345, 1, 0)
knobs = sorted({kv.split("=")[0] for v in variants for kv in v.split() if kv})


def timed(reps=5):
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(); A.band_quantiles(b, n); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


for _ in range(3):
    A.band_quantiles(b, n)          # warm-up: scratch, clocks
res = {v: [] for v in variants}
for rnd in range(int(os.environ.get("K3_AB_ROUNDS", "7"))):
    for v in variants:
        for k in knobs:
            os.environ.pop(k, None)
        for kv in v.split():
            k, val = kv.split("=")
            os.environ[k] = val
        res[v].append(timed())
gb = 8 * n * b.slab.shape[0] / 1e9
for v in variants:
    ms = float(np.median(res[v]))
    print(f"{v or '(default)':60s} {ms:.3f} ms  (min {min(res[v]):.3f})  {gb/ms:.3f} TB/s  frac {gb/ms/8:.4f}")
