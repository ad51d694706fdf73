"""K1 time vs the length and kind of other_income_streams (ABI v7: the list has any length), count-only and full output:
k frozen (non-indexed) streams with their lock columns in LDS (default) or forced into the global overflow block
(MCR_K1_LDS_LOCK_SLOTS), and lists beyond the 16-record by-value block (device table, XS variants).
    python tools/streams_timing.py [--paths 1000000]"""
import argparse, json, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from monte_carlo_retirement_amd import Config, params_from_config
from monte_carlo_retirement_amd import engine as E

ap = argparse.ArgumentParser()
ap.add_argument("--paths", type=int, default=1_000_000)
args = ap.parse_args()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = dict(json.load(open(os.path.join(root, "scenarios", "config.json"))), seed=12345)
base["other_income_streams"] = []


def stream(i, indexed):
    return {"name": f"s{i}", "monthly_amount_today": 150.0 + 10 * i, "start_at_age": 58.0 + 0.5 * i, "duration_years": None if i % 3 else 20,
            "inflation_indexed": indexed, "tax_rate": 0.1}


def timed(p, want, n, wm=233):
    b = E.DeviceBatch(p, wm, n, want=want)
    ts = []
    for _ in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(); b.launch(12345, 1, 0); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ok = int(b.counters[0])
    del b
    torch.cuda.empty_cache()
    return statistics.median(ts[2:]), ok


n = args.paths
print(f"config.json scenario, wm=233, {n} paths; ms per launch (median of 7)")
for label, streams in [("0 streams", []),
                       ("2 indexed", [stream(i, True) for i in range(2)]),
                       ("16 indexed", [stream(i, True) for i in range(16)]),
                       ("17 indexed (XS)", [stream(i, True) for i in range(17)]),
                       ("40 indexed (XS)", [stream(i, True) for i in range(40)]),
                       ("2 frozen", [stream(i, False) for i in range(2)]),
                       ("4 frozen", [stream(i, False) for i in range(4)]),
                       ("8 frozen", [stream(i, False) for i in range(8)]),
                       ("16 frozen", [stream(i, False) for i in range(16)]),
                       ("40 frozen (XS)", [stream(i, False) for i in range(40)])]:
    p = params_from_config(Config(**dict(base, other_income_streams=streams)))
    row = []
    n_frozen = sum(not s["inflation_indexed"] for s in streams)
    caps = [None] if n_frozen == 0 else [None, 64, 0, 4, 8]
    for cap in caps:
        if cap is None:
            os.environ.pop("MCR_K1_LDS_LOCK_SLOTS", None)
        else:
            os.environ["MCR_K1_LDS_LOCK_SLOTS"] = str(cap)
        c, ok = timed(p, "count", n)
        f, _ = timed(p, "full", n)
        row.append(f"lds<={'auto' if cap is None else cap}: count {c:7.3f} full {f:7.3f}")
    os.environ.pop("MCR_K1_LDS_LOCK_SLOTS", None)
    print(f"{label:18s} ok={ok:8d}  " + " | ".join(row))
