import json, os, sys, statistics
sys.path.insert(0, os.getcwd())
import torch
from monte_carlo_retirement_amd import Config, params_from_config
from monte_carlo_retirement_amd import aggregation as A, engine as E
n = 10_000_000
cfg = Config(**dict(json.load(open("scenarios/jorge.json")), equity_inflation_correlation=0.3, seed=12345))
p = params_from_config(cfg)
b = E.DeviceBatch(p, 75, n, want="full")
b.launch(12345, 1, 0)
def timed():
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(); A.band_quantiles(b, n); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return statistics.median(ts[6:])
print("== same slab, scratch re-allocated each cycle (fresh hipMalloc)")
for c in range(8):
    b.release_scratch(); torch.cuda.empty_cache()
    junk = torch.empty((c * 37 + 1) * (1 << 20), dtype=torch.uint8, device="cuda")   # shift what the driver hands out next
    t = timed()
    print(f"cycle {c}: scratch at {b._rq_scratch.data_ptr():#x}  {t:.3f} ms", flush=True)
    del junk
print("== same slab, same scratch, K1 relaunched in between (data rewritten)")
for c in range(4):
    b.launch(12345, 1, 0)
    print(f"cycle {c}: {timed():.3f} ms", flush=True)
