"""Which rows of the bracket test fall back (one row per call)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, pandas as pd, torch
os.environ["MCR_RQ_BRACKET_MIN_N"] = "1"
from monte_carlo_retirement_amd import aggregation as A
from test_simulator_gpu import _rows_for_bracket_test
for n, stride in [(300_001, 300_032), (4096, 4096), (65536, 65600)]:
    rng = np.random.default_rng(n)
    rows = _rows_for_bracket_test(rng, n, stride)
    for qs in (A.TRAJECTORY_QUANTILES, tuple(np.linspace(0.02, 0.98, 15))):
        fb = []
        for r in range(rows.shape[0]):
            d = torch.as_tensor(rows[r:r + 1], device="cuda")
            got, counts = A.row_quantiles(d, n, qs)
            exp = pd.DataFrame(rows[r:r + 1, :n].T).quantile(list(qs), axis=0).T.to_numpy()
            fb.append((r, A.last_fallback_rows(), bool(np.array_equal(got, exp, equal_nan=True))))
        print(n, len(qs), fb)
# the production shape: lognormal rows, NaN-tailed WR rows
n = 4_000_000
rng = np.random.default_rng(1)
rows = np.empty((6, n))
rows[0] = rng.lognormal(14, 0.05, n); rows[1] = rng.lognormal(14, 1.0, n); rows[2] = 240000.0
rows[3] = np.where(rng.random(n) < 0.02, np.nan, rng.normal(4, 0.5, n)); rows[4] = np.where(rng.random(n) < 0.02, 0.0, rng.lognormal(15, 1.5, n))
rows[5] = rng.normal(0, 1, n)
d = torch.as_tensor(rows, device="cuda")
got, counts = A.row_quantiles(d, n, A.TRAJECTORY_QUANTILES)
print("4e6 fallback rows:", A.last_fallback_rows(), [bool(np.array_equal(got[r], np.nanquantile(rows[r], A.TRAJECTORY_QUANTILES))) for r in range(6)])
