import json, os, sys, statistics
sys.path.insert(0, os.getcwd())
import torch
from monte_carlo_retirement_amd import Config, params_from_config
from monte_carlo_retirement_amd import engine as E
cfg = Config(**dict(json.load(open("scenarios/config.json")), seed=12345))
p = params_from_config(cfg)
for want, seg in (("full", "0"), ("full", ""), ("summary", "0"), ("summary", "")):
    os.environ.pop("MCR_K1_SEGMENTS", None)
    if seg:
        os.environ["MCR_K1_SEGMENTS"] = seg
    for n in (655360, 983040, 1000000, 1310720, 2000000):
        b = E.DeviceBatch(p, 232, n, want=want)
        ts = []
        for i in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record(); b.launch(12345, 1, 0); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms = statistics.median(ts[2:])
        print(f"{want:8s} {'plain ' if seg == '0' else 'sliced'} n={n:8d} blocks/1280={n/256/1280:6.2f}  {ms:7.3f} ms  {ms/n*1e6:6.3f} ms per 1e6", flush=True)
        del b; torch.cuda.empty_cache()
