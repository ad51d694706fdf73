"""Does what was allocated BEFORE the slab decide the K3 mode?  A dummy tensor of varying size is allocated first (and kept),
then the batch; the bands are timed.  Each case in a fresh state of the caching allocator."""
import json, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from monte_carlo_retirement_amd import Config, params_from_config
from monte_carlo_retirement_amd import aggregation as A, engine as E
n = 10_000_000
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = Config(**dict(json.load(open(os.path.join(root, "scenarios/jorge.json"))), equity_inflation_correlation=0.3, seed=12345))
p = params_from_config(cfg)
MiB = 1 << 20
for dummy_mib in [0, 2, 4, 8, 16, 64, 256, 1024, 4096, 0, 2, 4096]:
    torch.cuda.empty_cache()
    dummy = torch.empty(dummy_mib * MiB, dtype=torch.uint8, device="cuda") if dummy_mib else None
    b = E.DeviceBatch(p, 75, n, want="full")
    b.launch(12345, 1, 0)
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(); A.band_quantiles(b, n); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"dummy {dummy_mib:5d} MiB: slab at {b.slab.data_ptr():#x}  steady median {statistics.median(ts[8:]):.3f} ms", flush=True)
    del b, dummy
