import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
for k in (1, 2, 4, 5, 8, 17):
    months = list(range(217, 217 + k))
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); sim._probe_many(months, 50000); ts.append(time.perf_counter() - t0)
    print(f"{k:2d} candidates in one call: median {sorted(ts)[3]*1e3:.2f} ms  ({sorted(ts)[3]*1e3/k:.2f} ms/candidate)")
for slots in (1, None):
    s2 = RetirementMonteCarloSimulator(cfg)
    if slots:
        s2._speculation_slots = lambda n: 1
        one = s2._probe_many
        s2._probe_many = lambda months, n: {m: one([m], n)[m] for m in months}   # strictly one launch at a time
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); r = s2.find_minimum_working_months(verbose=False); ts.append(time.perf_counter() - t0)
    print("search,", "one probe at a time:" if slots else "batched candidates: ", r[0], r[1], len(r[2]), "probes", round(sorted(ts)[2] * 1e3, 1), "ms")
