import json, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from monte_carlo_retirement_amd import Config
from monte_carlo_retirement_amd.simulation import RetirementMonteCarloSimulator
cfg = Config(**dict(json.load(open("scenarios/config.json")), num_simulations_search=50000, seed=12345))
for env in ({}, {"MCR_K1_SEGMENTS": "0"}):
    os.environ.pop("MCR_K1_SEGMENTS", None); os.environ.update(env)
    sim = RetirementMonteCarloSimulator(cfg)
    sim.find_minimum_working_months(verbose=False)
    rounds = []
    inner = sim._probe_many
    def timed(months, n):
        t0 = time.perf_counter(); r = inner(months, n); rounds.append((len(months), (time.perf_counter() - t0) * 1e3, list(months)[:3])); return r
    sim._probe_many = timed
    t0 = time.perf_counter(); sim.find_minimum_working_months(verbose=False); tot = (time.perf_counter() - t0) * 1e3
    print(env, f"total {tot:.2f} ms", [(k, round(ms, 2)) for k, ms, _ in rounds])
