import json, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from monte_carlo_retirement_amd import Config, params_from_config, engine as E
p = params_from_config(Config(**dict(json.load(open("scenarios/config.json")), seed=12345)))
for n in (64, 4096, 50000):
    for months in ([233], [231, 232, 233], list(range(217, 234))):
        E.probe_months(p, 12345, 0, 0, n, months).cpu()
        ts = []
        for _ in range(15):
            t0 = time.perf_counter(); c = E.probe_months(p, 12345, 0, 0, n, months); t1 = time.perf_counter(); c = c.cpu(); t2 = time.perf_counter()
            ts.append((t2 - t0, t1 - t0))
        ts.sort()
        print(f"n={n:6d} candidates={len(months):2d}: total {ts[7][0]*1e3:.3f} ms, enqueue {ts[7][1]*1e3:.3f} ms")
