import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, pandas as pd, torch
os.environ["MCR_RQ_BRACKET_MIN_N"] = "1"
from monte_carlo_retirement_amd import aggregation as A
from test_simulator_gpu import _rows_for_bracket_test
for n, stride in [(65536, 65600), (300_001, 300_032)]:
    rows = _rows_for_bracket_test(np.random.default_rng(n), n, stride)
    for qs in (A.TRAJECTORY_QUANTILES, A.WR_QUANTILES, (0.0, 1.0, 0.5, 0.999, 0.001)):
        fb, bad = [], []
        for r in range(14):
            got, _ = A.row_quantiles(torch.as_tensor(rows[r:r + 1], device="cuda"), n, qs)
            exp = pd.DataFrame(rows[r:r + 1, :n].T).quantile(list(qs), axis=0).T.to_numpy()
            if A.last_fallback_rows(): fb.append(r)
            if not np.array_equal(got, exp, equal_nan=True): bad.append(r)
        print(n, qs[:3], "fallback rows", fb, "WRONG" if bad else "", bad)
