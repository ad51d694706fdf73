"""Path-level accuracy of the kernel's random-number / exp arithmetic against the oracle (libm evaluation of the same
stream) and count-only timing, for A/B of library builds (MCR_HIP_LIBRARY):  max over paths and summary fields of
|gpu - oracle| / max(|oracle|, the path's balance at retirement), flipped Success flags, ms per 1e6 count-only paths."""
import json, os, sys, statistics, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from monte_carlo_retirement_amd import Config, params_from_config
from monte_carlo_retirement_amd import engine as E
from oracle import oracle as O
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
golden = json.load(open(os.path.join(root, "tests", "golden", "paths_injected.json")))
n = int(os.environ.get("K1_ACC_PATHS", "200000"))
for name in os.environ.get("K1_ACC_SCENARIOS", "C1_config_json_wm233,S60_wm120,ANNUAL_wm50,MIXED_wm36,FAILING_wm24").split(","):
    g = [x for x in golden if x["name"] == name][0]
    p = params_from_config(Config(**g["cfg"]))
    sid = {"search": 0, "final": 1}[g["stream"]]
    gpu = E.run_batch_host(p, g["seed"], sid, 0, n, g["working_months"], want_trajectories=False)
    T = 16; per = (n + T - 1) // T; parts = [None] * T
    def work(t):
        parts[t] = O.run_batch(p, g["seed"], sid, t * per, max(0, min(per, n - t * per)), g["working_months"], want_trajectories=False)
    ths = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    [t.start() for t in ths]; [t.join() for t in ths]
    cpu = {k: np.concatenate([q[k] for q in parts]) for k in E.SUMMARY_FIELDS + ("success",)}
    scale = np.maximum(np.abs(cpu["start_balance"]), 1.0)
    worst, med = 0.0, 0.0
    for k in E.SUMMARY_FIELDS:
        both = np.isnan(gpu[k]) & np.isnan(cpu[k])
        err = np.where(both, 0.0, np.abs(gpu[k] - cpu[k])) / np.maximum(np.abs(np.nan_to_num(cpu[k])), scale)
        worst = max(worst, float(np.nanmax(err)))
        med = max(med, float(np.nanmedian(err)))
    print(f"{name:24s} {n} paths: worst scaled error {worst:.3e}  (largest per-field median {med:.1e})  flipped flags {int((gpu['success'] != cpu['success']).sum())}"
          f"  success {int(cpu['success'].sum())}", flush=True)
cfg = Config(**dict(json.load(open(os.path.join(root, "scenarios", "config.json"))), seed=12345))
b = E.DeviceBatch(params_from_config(cfg), 233, 1_000_000, want="count")
ts = []
for i in range(40):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); b.launch(12345, 1, i * 1_000_000); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print(f"count-only 1e6 paths x 833 months: median {statistics.median(ts[10:]):.3f} ms")
