"""Is the K3 time of a process decided by where the slab lands?  Re-allocate the full-output batch a few times in ONE
process (free, empty the caching allocator, allocate again) and time the bands each time (steady state)."""
import json, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from monte_carlo_retirement_amd import Config, params_from_config
from monte_carlo_retirement_amd import aggregation as A, engine as E
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = Config(**dict(json.load(open(os.path.join(root, "scenarios/jorge.json"))), equity_inflation_correlation=0.3, seed=12345))
p = params_from_config(cfg)
for cycle in range(int(sys.argv[2]) if len(sys.argv) > 2 else 5):
    b = E.DeviceBatch(p, 75, n, want="full")
    b.launch(12345, 1, 0)
    ts = []
    for _ in range(24):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(); A.band_quantiles(b, n); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    # a plain streaming read of the same slab on the same allocation (no scratch, no candidates, torch's own reduction kernel)
    ss = []
    for _ in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(); b.slab[:, :n].sum(); e1.record(); torch.cuda.synchronize()
        ss.append(e0.elapsed_time(e1))
    gb = 8 * n * b.slab.shape[0] / 1e9
    print(f"cycle {cycle}: slab at {b.slab.data_ptr():#x} scratch at {b._rq_scratch.data_ptr():#x}  steady median {statistics.median(ts[8:]):.3f} ms  min {min(ts):.3f}"
          f"  | torch.sum of the slab {statistics.median(ss[4:]):.3f} ms = {gb / statistics.median(ss[4:]):.2f} TB/s", flush=True)
    del b
    torch.cuda.empty_cache()
