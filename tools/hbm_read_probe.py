"""What a plain streaming read achieves on this box: torch reductions over a slab of the K3 size (136 x 1e7 doubles),
next to K3's own time on the same box (tools/k3_time.py).  Box-to-box spread is ~5 %."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
n, rows = 10_000_000, 136
x = torch.rand((rows, n), dtype=torch.float64, device="cuda")
gb = x.numel() * 8 / 1e9
for name, fn in (("sum", lambda: x.sum()), ("max", lambda: x.max()), ("row sums", lambda: x.sum(dim=1)), ("copy (r+w)", lambda: x.clone())):
    ts = []
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = float(np.median(ts[1:]))
    traffic = gb * (2 if "copy" in name else 1)
    print(f"{name:12s} {ms:7.3f} ms  {traffic / ms:6.3f} TB/s  ({traffic / ms / 8:.3f} of 8 TB/s)")
