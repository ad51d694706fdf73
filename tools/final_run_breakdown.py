"""Where the time of RetirementMonteCarloSimulator.run_monte_carlo_simulations goes at 10^6 paths (the search block's final run):
the function's own steps timed one by one, host clock, stream synchronised after each."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd, torch
from monte_carlo_retirement_amd import Config, engine as E, aggregation as A
from monte_carlo_retirement_amd import simulation as S
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = Config(**dict(json.load(open(os.path.join(root, "scenarios/config.json"))), seed=12345))
sim = S.RetirementMonteCarloSimulator(cfg); sim.use_final_seeds()
n, wm = 1_000_000, 232
for _ in range(3):
    sim.run_monte_carlo_simulations(wm, n)
marks = []
def mark(label):
    torch.cuda.synchronize(); marks.append((label, time.perf_counter()))
for rep in range(3):
    marks.clear()
    mark("start")
    batch = E.DeviceBatch(sim._current_params(), wm, n, want="full", device=0); mark("DeviceBatch (params + alloc)")
    batch.launch(sim._batch_rng(n), sim._stream_id, 0); mark("K1 full output")
    df = S._summary_frame(batch, n); mark("_summary_frame")
    q = A.band_quantiles(batch, n); mark("band_quantiles")
    frames = [pd.DataFrame(q[0]), pd.DataFrame(q[1]), pd.DataFrame(q[2])]; mark("band frames")
    picked = np.random.RandomState(sim.main_seed).choice(n, size=5, replace=False); mark("RandomState.choice(n, 5, replace=False)")
    s1 = S._gather_columns(batch.trajectory, picked).tolist(); s2 = S._gather_columns(batch.real_trajectory, picked).tolist(); mark("gather samples")
    del batch; mark("free the batch")
    del df, frames; mark("free the frames")
print("\n".join(f"{b[0]:45s} {(b[1] - a[1]) * 1e3:7.2f} ms" for a, b in zip(marks, marks[1:])))
t0 = time.perf_counter(); r = sim.run_monte_carlo_simulations(wm, n); torch.cuda.synchronize(); print(f"{'whole call':45s} {(time.perf_counter() - t0) * 1e3:7.2f} ms")
