"""Count-only K1 on the reference's DEFAULT tax configuration (inv1 on the annual-gains system, inv2 taxed on realized
gains: one taxed asset) and on config.json (both taxed): for A/B of the per-asset tax variants (MCR_HIP_LIBRARY)."""
import json, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from monte_carlo_retirement_amd import Config, params_from_config
from monte_carlo_retirement_amd import engine as E
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = dict(json.load(open(os.path.join(root, "scenarios", "config.json"))), seed=12345)
cases = [("both assets taxed (config.json)", base),
         ("inv2 only (the Config defaults)", dict(base, inv1_use_realized_gains_tax_system=False)),
         ("inv1 only", dict(base, inv2_use_realized_gains_tax_system=False)),
         ("inv2 realized + inv1 annual 15 %", dict(base, inv1_use_realized_gains_tax_system=False, inv1_annual_tax_on_gains_rate=0.15))]
for name, c in cases:
    p = params_from_config(Config(**c))
    b = E.DeviceBatch(p, 233, 4_000_000, want="count")
    ts = []
    for i in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); b.launch(12345, 1, i * 4_000_000); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = statistics.median(ts[3:])
    print(f"{name:36s} {ms:8.3f} ms per 4e6 paths  {4e3 / ms:7.2f} M paths/s  success {int(b.counters[0]) / int(b.counters[1]):.5f}")
