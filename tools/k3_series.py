import json, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from monte_carlo_retirement_amd import Config, params_from_config
from monte_carlo_retirement_amd import aggregation as A, engine as E
n = 10_000_000
cfg = Config(**dict(json.load(open("scenarios/jorge.json")), equity_inflation_correlation=0.3, seed=12345))
b = E.DeviceBatch(params_from_config(cfg), 75, n, want="full")
b.launch(12345, 1, 0)
ts = []
for _ in range(40):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record(); A.band_quantiles(b, n); e1.record(); torch.cuda.synchronize()
    ts.append(round(e0.elapsed_time(e1), 3))
print(ts)
import time
time.sleep(2.0)
ts = []
for _ in range(10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record(); A.band_quantiles(b, n); e1.record(); torch.cuda.synchronize()
    ts.append(round(e0.elapsed_time(e1), 3))
print("after 2 s idle", ts)
