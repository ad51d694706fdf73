"""Time-sliced path blocks (PHASE 3 of path_kernel, MCR_K1_SEGMENTS) against the plain launch: identical counters / year bins /
histogram bins, and ms per launch, over batch sizes around the chip's resident capacity.
    python tools/k1_segments_ab.py [--segments 0,2,4,8]"""
import argparse, json, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from monte_carlo_retirement_amd import Config, params_from_config
from monte_carlo_retirement_amd import engine as E

ap = argparse.ArgumentParser()
ap.add_argument("--segments", default="0,4")
ap.add_argument("--sizes", default="500000,786432,1000000,1179648,2000000,4000000")
args = ap.parse_args()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
scen = [("config.json wm=233", Config(**dict(json.load(open(os.path.join(root, "scenarios", "config.json"))), seed=12345)), 233),
        ("S60 wm=120", Config(**dict(json.load(open(os.path.join(root, "scenarios", "config.json"))), seed=12345, initial_balance=2.0e6,
                                      inv1_returns_volatility=0.15, equity_inflation_correlation=0.3)), 120)]
edges = np.geomspace(1.0, 1e12, 101)
for name, cfg, wm in scen:
    p = params_from_config(cfg)
    for n in [int(x) for x in args.sizes.split(",")]:
        ref, row = None, []
        for q in [int(x) for x in args.segments.split(",")]:
            os.environ["MCR_K1_SEGMENTS"] = str(q)
            b = E.DeviceBatch(p, wm, n, want="count", hist_edges=edges)
            ts = []
            for i in range(9):
                b.zero_counters()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(); e0.record(); b.launch(12345, 1, 7 * n); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            vec = b.reduce_vec.cpu().numpy().copy()
            if ref is None:
                ref = vec
            same = bool(np.array_equal(vec, ref))
            row.append(f"q={q}: {statistics.median(ts[2:]):7.3f} ms {'same' if same else 'DIFFERENT'}")
            del b
        print(f"{name:20s} n={n:8d}  " + " | ".join(row) + f"   success {int(ref[0])}", flush=True)
os.environ.pop("MCR_K1_SEGMENTS", None)
