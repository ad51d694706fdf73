import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
os.environ["MCR_RQ_BRACKET_MIN_N"] = "1"
from monte_carlo_retirement_amd import aggregation as A
from test_simulator_gpu import _rows_for_bracket_test
n, stride = 300_001, 300_032
rows = _rows_for_bracket_test(np.random.default_rng(n), n, stride)
qs = A.TRAJECTORY_QUANTILES
good = [0, 2, 4, 5, 8, 10]
for sub in ([0], [0, 2], good, [0, 1], [0, 3], good + [3], good + [7], list(range(14)), list(range(14))):
    d = torch.as_tensor(rows[sub], device="cuda")
    A.row_quantiles(d, n, qs)
    print(sub, "->", A.last_fallback_rows())
