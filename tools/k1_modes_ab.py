"""K1 variants with per-path outputs: summary output on the S60 shape (720 months) and full trajectories on the jorge shape
(555 months, one non-indexed stream = one LDS lock column), median of event-timed launches.  For A/B of library builds
(MCR_HIP_LIBRARY)."""
import json, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from monte_carlo_retirement_amd import Config, params_from_config
from monte_carlo_retirement_amd import engine as E
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def cfg(name, **over):
    return params_from_config(Config(**dict(json.load(open(os.path.join(root, "scenarios", name))), seed=12345, **over)))
cases = [("summary S60 2e7", cfg("config.json", initial_balance=2.0e6, inv1_returns_volatility=0.15, equity_inflation_correlation=0.3), 120, 20_000_000, "summary"),
         ("full jorge 1e7", cfg("jorge.json", equity_inflation_correlation=0.3), 75, 10_000_000, "full"),
         ("full config.json 4e6", cfg("config.json"), 233, 4_000_000, "full"),
         ("count config.json 1e6", cfg("config.json"), 233, 1_000_000, "count")]
for name, p, wm, n, want in cases:
    b = E.DeviceBatch(p, wm, n, want=want)
    ts = []
    for i in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(); b.launch(12345, 1, 0); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = statistics.median(ts[2:])
    print(f"{name:24s} {ms:9.3f} ms  {n / ms / 1e3:8.2f} M paths/s  success {int(b.counters[0]) / n:.6f}")
    del b
    torch.cuda.empty_cache()
