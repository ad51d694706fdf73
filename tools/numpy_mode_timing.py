import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from monte_carlo_retirement_amd import Config, params_from_config
from monte_carlo_retirement_amd import _native as N, engine as E
p = params_from_config(Config(**json.load(open("scenarios/config.json"))))
for name, rng in (("philox", 12345), ("numpy", N.numpy_rng(12345))):
    b = E.DeviceBatch(p, 233, 1_000_000, want="count")
    b.launch(rng, 1, 0); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        b.zero_counters()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); b.launch(rng, 1, 0); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    print(name, round(float(np.median(ts)), 2), "ms per 1e6 paths; success", int(b.counters[0].item()))
