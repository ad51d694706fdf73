"""Where the host time of run_monte_carlo_simulations goes at 10^7 paths (beyond the ~71 ms of kernels)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from monte_carlo_retirement_amd import Config, params_from_config, engine as E, aggregation as A
from monte_carlo_retirement_amd import simulation as S
cfg = Config(**dict(json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenarios/jorge.json"))), seed=12345, equity_inflation_correlation=0.3))
n = 10_000_000
def T(label, fn, reps=3):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"{label:50s} {min(ts)*1e3:8.1f} ms"); return r
b = T("DeviceBatch alloc (full)", lambda: E.DeviceBatch(params_from_config(cfg), 75, n, want="full"), 2)
T("K1 launch", lambda: b.launch(12345, 1, 0))
T("band_quantiles", lambda: A.band_quantiles(b, n))
T("_summary_frame", lambda: S._summary_frame(b, n))
T("  stack 6 columns on device", lambda: torch.stack([b.summary[f][:n] for f in S._FIELD_OF.values()]))
packed = torch.stack([b.summary[f][:n] for f in S._FIELD_OF.values()])
T("  pinned alloc [6,n] f64", lambda: torch.empty(packed.shape, dtype=packed.dtype, pin_memory=True))
host = torch.empty(packed.shape, dtype=packed.dtype, pin_memory=True)
T("  D2H into pinned", lambda: host.copy_(packed))
pg = torch.empty(packed.shape, dtype=packed.dtype)
T("  D2H into pageable", lambda: pg.copy_(packed))
T("  packed.cpu()", lambda: packed.cpu())
