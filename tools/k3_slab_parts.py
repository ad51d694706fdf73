"""Where the K3 slab pass spends its time: the EXPERIMENT build of the library (csrc/build.py build(variant="exp",
extra_flags=["-DMCR_RQ_EXPERIMENT"])) switches parts of rq_slab_kernel's element loop off through MCR_RQ_DBG (results are
invalid then: the call returns after the slab pass).  Run under the kernel trace; tools/k3_slab_parts.sh prints the slab
kernel's duration per setting, interleaved over rounds on ONE allocation of the slab.

    MCR_HIP_LIBRARY=.../libmcr_hip_exp.so python tools/k3_slab_parts.py 10000000 0 1 2 4 8 16 ...
bits: 1 no staging of bracket members, 2 flush without the global atomic + candidate stores, 4 flush without the
sub-histogram tally, 32 flush without the stores, 64 flush without the atomic, 128 nontemporal candidate stores (a valid pass),
256 nothing switched off, but the call returns after the slab pass (use with narrowed brackets, e.g. 256:1.6)
"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from monte_carlo_retirement_amd import Config, params_from_config
from monte_carlo_retirement_amd import aggregation as A, engine as E

n = int(sys.argv[1])
settings = sys.argv[2:] or ["0"]      # "dbg" or "dbg:sigmas" (MCR_RQ_SIGMAS: half-width of the fine brackets, default 4.5)
rounds = int(os.environ.get("K3_PARTS_ROUNDS", "5"))
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = Config(**dict(json.load(open(os.path.join(root, "scenarios/jorge.json"))), equity_inflation_correlation=0.3, seed=12345))
b = E.DeviceBatch(params_from_config(cfg), 75, n, want="full")
b.launch(12345, 1, 0)
os.environ["MCR_RQ_DBG"] = "0"
for _ in range(6):
    A.band_quantiles(b, n)          # warm-up: scratch, clocks
torch.cuda.synchronize()
print("order", " ".join(s for _ in range(rounds) for s in settings), flush=True)
for _ in range(rounds):
    for s in settings:
        dbg, _, sig = s.partition(":")
        os.environ["MCR_RQ_DBG"] = dbg
        os.environ["MCR_RQ_SIGMAS"] = sig or "4.5"
        A.band_quantiles(b, n)
torch.cuda.synchronize()
