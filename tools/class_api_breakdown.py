"""Where the wall time of RetirementMonteCarloSimulator.run_monte_carlo_simulations goes at the BASELINE configs[2] size
(jorge.json rho = 0.3, wm = 75, 10^7 paths): the function's own steps, host clock, WITHOUT extra synchronisations (a step that
only enqueues shows its enqueue cost; the waits show up where the function itself waits)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd, torch
from monte_carlo_retirement_amd import Config, engine as E, aggregation as A
from monte_carlo_retirement_amd import simulation as S
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = Config(**dict(json.load(open(os.path.join(root, "scenarios/jorge.json"))), seed=12345, equity_inflation_correlation=0.3))
sim = S.RetirementMonteCarloSimulator(cfg); sim.use_final_seeds()
n, wm = int(os.environ.get("N", "10000000")), 75
r = sim.run_monte_carlo_simulations(wm, n); del r
for rep in range(3):
    marks = [("start", time.perf_counter())]
    def mark(label): marks.append((label, time.perf_counter()))
    torch.cuda.synchronize()
    marks = [("start", time.perf_counter())]
    batch = E.DeviceBatch(sim._current_params(), wm, n, want="full", device=0); mark("DeviceBatch (params + alloc)")
    batch.launch(sim._batch_rng(n), sim._stream_id, 0); mark("K1 enqueue")
    sampler = S._BackgroundCall(sim._sample_columns, n); mark("sample_columns thread start")
    dl = S._SummaryDownload(batch, n); mark("_SummaryDownload.start (pinned alloc + enqueue)")
    q = A.band_quantiles(batch, n); mark("band_quantiles (waits for K1 + K3)")
    frames = [pd.DataFrame(q[0]), pd.DataFrame(q[1]), pd.DataFrame(q[2])]; mark("band frames")
    picked = sampler.result(); mark("join the sample_columns thread"); s1 = S._gather_columns(batch.trajectory, picked).tolist(); s2 = S._gather_columns(batch.real_trajectory, picked).tolist(); mark("gather samples")
    df = dl.frame(); mark("_SummaryDownload.frame (waits for the copies)")
    del batch; mark("free the batch")
    total = marks[-1][1] - marks[0][1]
    del df, frames, dl; mark("free the frames")
    print(f"--- rep {rep}: {total * 1e3:.1f} ms before the frees")
    print("\n".join(f"{b[0]:55s} {(b[1] - a[1]) * 1e3:8.2f} ms" for a, b in zip(marks, marks[1:])))
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = sim.run_monte_carlo_simulations(wm, n); torch.cuda.synchronize(); t1 = time.perf_counter(); del r; t2 = time.perf_counter()
    print(f"{'whole call':55s} {(t1 - t0) * 1e3:8.2f} ms  (+ {1e3 * (t2 - t1):.2f} ms to drop the result)")
