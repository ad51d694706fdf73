"""K3 timing on the configs[2] shape (jorge rho=0.3, wm=75): bands of the [136, n] slab, median of 5 event-timed calls."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from monte_carlo_retirement_amd import Config, params_from_config
from monte_carlo_retirement_amd import aggregation as A, engine as E
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
cfg = Config(**dict(json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenarios/jorge.json"))), equity_inflation_correlation=0.3, seed=12345))
b = E.DeviceBatch(params_from_config(cfg), 75, n, want="full")
b.launch(12345, 1, 0)
ts = []
for _ in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record(); A.band_quantiles(b, n); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ms = float(np.median(ts[1:]))
gb = 8 * n * b.slab.shape[0] / 1e9
print(f"n={n} rows={b.slab.shape[0]} K3 {ms:.3f} ms  {gb/ms:.3f} TB/s  frac {gb/ms/8:.4f}  fallback {A.last_fallback_rows()}")
