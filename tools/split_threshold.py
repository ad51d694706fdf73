"""Where the producer / consumer form of small launches stops paying (MCR_K1_SPLIT_MAX_WAVES): ms per call of 1 and 3 search
candidates over n paths with the split forced on / off."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from monte_carlo_retirement_amd import Config, params_from_config, engine as E
p = params_from_config(Config(**dict(json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenarios", "config.json"))), seed=12345)))
for n in (20000, 50000, 100000, 150000, 200000, 300000):
    row = []
    for months in ([233], [231, 232, 233]):
        for limit in ("1000000", "0"):
            os.environ["MCR_K1_SPLIT_MAX_WAVES"] = limit
            E.probe_months(p, 12345, 0, 0, n, months).cpu()
            ts = []
            for _ in range(11):
                t0 = time.perf_counter(); E.probe_months(p, 12345, 0, 0, n, months).cpu(); ts.append(time.perf_counter() - t0)
            row.append(f"{len(months)} cand {'split' if limit != '0' else 'plain'} {sorted(ts)[5]*1e3:6.3f}")
    print(f"n={n:7d} ({(n + 63) // 64:5d} path-waves): " + " | ".join(row), flush=True)
os.environ.pop("MCR_K1_SPLIT_MAX_WAVES", None)
