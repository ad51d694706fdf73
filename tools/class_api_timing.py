import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from monte_carlo_retirement_amd import Config
from monte_carlo_retirement_amd.simulation import RetirementMonteCarloSimulator
cfg = Config(**dict(json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenarios/jorge.json"))), seed=12345, equity_inflation_correlation=0.3))
sim = RetirementMonteCarloSimulator(cfg); sim.use_final_seeds()
sim.run_monte_carlo_simulations(75, 100000)
for n in (1_000_000, 10_000_000):
    t0 = time.perf_counter(); r = sim.run_monte_carlo_simulations(75, n); dt = time.perf_counter() - t0
    ok = r[0]["Success"].mean() * 100
    print(f"class API, n={n}: {dt*1e3:.0f} ms end to end (7-tuple incl. {n}-row summary_df); success {ok:.2f}%; peak GPU mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
