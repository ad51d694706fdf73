"""Soak test of the bracketed row quantiles far beyond the product sizes: 3 rows x 2e8 entries, checked against a full sort."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from monte_carlo_retirement_amd import aggregation as A
n = 200_000_000
g = torch.Generator(device="cuda"); g.manual_seed(1)
rows = torch.empty((3, n), dtype=torch.float64, device="cuda")
rows[0].normal_(1e6, 3e5, generator=g)
rows[1].log_normal_(10.0, 1.5, generator=g)
rows[2].uniform_(0, 1, generator=g); rows[2][rows[2] < 0.25] = float("nan")
torch.cuda.synchronize(); t0 = time.perf_counter()
q, c = A.row_quantiles(rows, n, A.TRAJECTORY_QUANTILES)
dt = time.perf_counter() - t0
print("bracket route: fallback rows", A.last_fallback_rows(), "first call %.1f ms (allocates the scratch)" % (dt * 1e3))
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    q, c = A.row_quantiles(rows, n, A.TRAJECTORY_QUANTILES)
    dt = time.perf_counter() - t0
    print("  again: %.2f ms = %.0f GB/s of slab" % (dt * 1e3, 3 * n * 8 / dt / 1e9))
# check against torch.kthvalue-based exact order statistics on row 0 (no NaN) and NaN-filtered row 2
for r in (0, 2):
    v = rows[r][~torch.isnan(rows[r])]
    m = v.numel()
    assert int(c[r]) == m
    vs, _ = torch.sort(v)
    for j, qq in enumerate(A.TRAJECTORY_QUANTILES):
        vi = (m - 1) * qq
        lo = int(np.floor(vi)); hi = min(lo + 1, m - 1); gm = vi - np.floor(vi)
        a, b = float(vs[lo]), float(vs[hi])
        ref = a + (b - a) * gm if gm < 0.5 else b - (b - a) * (1 - gm)
        assert q[r, j] == ref, (r, qq, q[r, j], ref)
    del vs, v
print("exact at n =", n)
