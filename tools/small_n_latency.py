import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monte_carlo_retirement_amd import Config
from monte_carlo_retirement_amd.simulation import RetirementMonteCarloSimulator
cfg = Config(**dict(json.load(open("scenarios/config.json")), seed=12345))
sim = RetirementMonteCarloSimulator(cfg); sim.use_final_seeds()
sim.run_monte_carlo_simulations(233, 1000)
for n in (1000, 10000, 100000):
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); sim.run_monte_carlo_simulations(233, n); ts.append(time.perf_counter() - t0)
    print(f"run_monte_carlo_simulations(233, {n}): median {sorted(ts)[3]*1e3:.2f} ms end to end (7-tuple with pandas frames)")
