"""K1 count-only timing over batch sizes (GPU box): median of 5 launches per size, config.json wm=233."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from monte_carlo_retirement_amd import Config, params_from_config
from monte_carlo_retirement_amd import engine as E
cfg = Config(**json.load(open("scenarios/config.json")))
p = params_from_config(cfg)
sizes = [int(a) for a in sys.argv[1:]] or [262144, 524288, 1000000, 1048576, 2000000, 4000000, 8000000]
for n in sizes:
    b = E.DeviceBatch(p, 233, n, want="count")
    b.launch(12345, 1, 0); torch.cuda.synchronize()
    ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); b.launch(12345, 1, 0); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ms = float(np.median(ts))
    print(f"n={n:8d} blocks={n//256:6d} per-CU={n/256/256:6.2f}  {ms:8.3f} ms  {n/ms/1e3:7.2f} Mpaths/s", flush=True)
