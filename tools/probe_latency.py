import json, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from monte_carlo_retirement_amd import Config
from monte_carlo_retirement_amd.simulation import RetirementMonteCarloSimulator
cfg = Config(**dict(json.load(open("scenarios/config.json")), num_simulations_search=50000, seed=12345))
sim = RetirementMonteCarloSimulator(cfg)
sim.use_search_seeds()
t0 = time.perf_counter(); sim._probe_success_probability(0, 50000); print("first probe (init)", round(time.perf_counter() - t0, 3), "s")
for wm in (0, 120, 233):
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); p = sim._probe_success_probability(wm, 50000); ts.append(time.perf_counter() - t0)
    print(f"wm={wm}: prob={p:.2f}% median {sorted(ts)[5]*1e3:.2f} ms/probe")
t0 = time.perf_counter(); r = sim.find_minimum_working_months(verbose=False); print("full search", r[0], r[1], len(r[2]), "probes", round((time.perf_counter() - t0) * 1e3, 1), "ms")
