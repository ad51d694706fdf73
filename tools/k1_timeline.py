"""Diagnostic: per-wave start/end wall-clock stamps and SIMD placement of one count-only K1 launch.
Needs the instrumented build:  python -c "from monte_carlo_retirement_amd.csrc import build; build.build(force=True, extra_flags=['-DMCR_K1_TIMELINE'])"
(then rebuild without the flag).  Output kept under profiles/r02/k1_timeline_*.txt."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from monte_carlo_retirement_amd import Config, params_from_config
from monte_carlo_retirement_amd import engine as E, _native as N
cfg = Config(**json.load(open("scenarios/config.json")))
p = params_from_config(cfg)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
b = E.DeviceBatch(p, 233, n, want="count")
nw = (n + 255) // 256 * 4
buf = torch.zeros(nw * 3, dtype=torch.int64, device="cuda")
rng = N.philox_rng(12345); rng.path_seeds = buf.data_ptr()
for rep in range(3):
    buf.zero_(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); b.launch(rng, 1, 0); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
t = buf.cpu().numpy().reshape(nw, 3)
t0, t1, hw = t[:, 0].astype(np.float64), t[:, 1].astype(np.float64), t[:, 2]
base = t0.min(); t0 = (t0 - base) / 100e6 * 1e3; t1 = (t1 - base) / 100e6 * 1e3   # 100 MHz wall clock -> ms
xcc = (hw >> 32) & 0xF; hwid = hw & 0xFFFFFFFF
cu = (hwid >> 8) & 0xF; se = (hwid >> 13) & 0x7; sh = (hwid >> 12) & 1; simd = (hwid >> 4) & 3
cuid = xcc * 64 + se * 16 + sh * 8 + cu   # unique-ish CU key
print(f"n={n} waves={nw} kernel {ms:.3f} ms; stamps span {t1.max():.3f} ms")
print("start-time histogram (ms):", np.histogram(t0, bins=np.arange(0, t1.max() + 0.5, 0.5))[0].tolist())
print("end-time histogram (ms):  ", np.histogram(t1, bins=np.arange(0, t1.max() + 0.5, 0.5))[0].tolist())
dur = t1 - t0
print("wave duration ms: min %.3f med %.3f max %.3f" % (dur.min(), np.median(dur), dur.max()))
ts = np.arange(0, t1.max(), 0.25)
running = [(int(((t0 <= x) & (t1 > x)).sum())) for x in ts]
print("waves running at t (every 0.25 ms):", running)
keys, counts = np.unique(cuid * 4 + simd, return_counts=True)
print("distinct SIMDs seen:", len(keys), "waves per SIMD: min %d max %d" % (counts.min(), counts.max()), "hist:", np.bincount(counts).tolist())
last = t1 > t1.max() - 1.0
print("waves ending in the last 1 ms:", int(last.sum()), "on SIMDs:", len(np.unique((cuid * 4 + simd)[last])))
