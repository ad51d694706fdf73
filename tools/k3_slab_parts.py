"""Where the K3 slab pass spends its time: the EXPERIMENT build of the library (csrc/build.py build(variant="exp",
extra_flags=["-DMCR_RQ_EXPERIMENT"])) switches parts of rq_slab_kernel's element loop off through MCR_RQ_DBG (results are
invalid then: the call returns after the slab pass).  Run under the kernel trace; tools/k3_slab_parts.sh prints the slab
kernel's duration per setting, interleaved over rounds on ONE allocation of the slab.

    MCR_HIP_LIBRARY=.../libmcr_hip_exp.so python tools/k3_slab_parts.py 10000000 0 1 2 4 8 16 ...
bits: 1 no staging of bracket members, 2 flush without the global atomic + candidate stores, 4 flush without the
sub-histogram tally, 8 no position counters (ds_add), 16 no bucket table / bound pair (position = one compare)
"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from monte_carlo_retirement_amd import Config, params_from_config
from monte_carlo_retirement_amd import aggregation as A, engine as E

n = int(sys.argv[1])
settings = [int(x) for x in sys.argv[2:]] or [0]
rounds = int(os.environ.get("K3_PARTS_ROUNDS", "5"))
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = Config(**dict(json.load(open(os.path.join(root, "scenarios/jorge.json"))), equity_inflation_correlation=0.3, seed=12345))
b = E.DeviceBatch(params_from_config(cfg), 75, n, want="full")
b.launch(12345, 1, 0)
os.environ["MCR_RQ_DBG"] = "0"
for _ in range(6):
    A.band_quantiles(b, n)          # warm-up: scratch, clocks
torch.cuda.synchronize()
print("order", " ".join(str(s) for _ in range(rounds) for s in settings), flush=True)
for _ in range(rounds):
    for s in settings:
        os.environ["MCR_RQ_DBG"] = str(s)
        A.band_quantiles(b, n)
torch.cuda.synchronize()
