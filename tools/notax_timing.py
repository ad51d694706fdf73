import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from monte_carlo_retirement_amd import Config, params_from_config
from monte_carlo_retirement_amd import engine as E
base = json.load(open("scenarios/config.json"))
for name, over in (("config.json (realized 10%/10%)", {}),
                   ("no realized-gains tax (reference defaults)", dict(inv1_use_realized_gains_tax_system=False, inv2_realized_gains_tax_rate=0.0)),
                   ("annual tax 15%/15%, no realized", dict(inv1_use_realized_gains_tax_system=False, inv1_annual_tax_on_gains_rate=0.15,
                                                             inv2_use_realized_gains_tax_system=False, inv2_annual_tax_on_gains_rate=0.15))):
    p = params_from_config(Config(**dict(base, **over)))
    n = 4_000_000
    b = E.DeviceBatch(p, 233, n, want="count")
    b.launch(12345, 1, 0); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        b.zero_counters(); e0.record(); b.launch(12345, 1, 0); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ms = float(np.median(ts))
    print(f"{name:48s} {ms:8.2f} ms  {n/ms/1e3:7.1f} Mpaths/s  success {int(b.counters[0].item())/n:.4f}")
