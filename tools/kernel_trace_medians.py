"""Per-kernel duration statistics of a rocprofv3 --kernel-trace CSV, computed ON the GPU box before the trace is deleted
(tools/collect_profiles.sh; the merge-back limit of gpurun_out/ is 64 MiB): calls, mean, median, min, max, and the same
over the TIMED launches only — rocprof's own --stats summary averages every launch of a kernel, warm-up included (round 3:
7.21 ms mean over 12 launches, max 8.39, against 6.8-7.0 ms in every bench run), so `median_timed_ms` is what bench.py's
HIP-event figure has to be compared with.

    python tools/kernel_trace_medians.py <dir with *_kernel_trace.csv> <out.json> [--skip-first SUBSTR=N ...]

--skip-first "path_kernel<0=5": the first 5 launches of kernels whose name contains the substring are warm-up."""
import collections
import csv
import glob
import json
import os
import statistics
import sys


def main():
    src, out = sys.argv[1], sys.argv[2]
    skip = {}
    args = sys.argv[3:]
    for i, a in enumerate(args):
        if a == "--skip-first":
            k, v = args[i + 1].rsplit("=", 1)
            skip[k] = int(v)
    f = max(glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        per[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    res = {}
    for name, rows in per.items():
        rows.sort()
        d = [x[1] / 1e6 for x in rows]
        n_skip = next((v for k, v in skip.items() if k in name), 0)
        t = d[n_skip:] if len(d) > n_skip else d
        res[name] = {"calls": len(d), "mean_ms": statistics.fmean(d), "median_ms": statistics.median(d), "min_ms": min(d), "max_ms": max(d),
                     "timed_calls": len(t), "skipped_warmup_calls": len(d) - len(t), "median_timed_ms": statistics.median(t),
                     "mean_timed_ms": statistics.fmean(t), "min_timed_ms": min(t), "max_timed_ms": max(t)}
    json.dump({"source": os.path.basename(f), "kernels": res}, open(out, "w"), indent=1)
    for name, v in sorted(res.items(), key=lambda kv: -kv[1]["mean_ms"] * kv[1]["calls"])[:8]:
        print(f"{v['calls']:5d} x  median {v['median_timed_ms']:9.4f} ms (timed {v['timed_calls']}, min {v['min_timed_ms']:.4f}, max {v['max_timed_ms']:.4f})  {name[:110]}")


if __name__ == "__main__":
    main()
