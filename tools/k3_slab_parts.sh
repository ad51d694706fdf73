#!/bin/bash
# slab-pass duration per MCR_RQ_DBG setting (tools/k3_slab_parts.py) from the kernel trace
#   bash tools/k3_slab_parts.sh gpurun_out/<dir> 10000000 0 1 2 4 8 16
set -o pipefail
OUT=${1:?output directory}; shift
export TMPDIR=/tmp
export MCR_HIP_LIBRARY=$PWD/monte_carlo_retirement_amd/csrc/libmcr_hip_exp.so
mkdir -p "$OUT"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/kt" -- python3 tools/k3_slab_parts.py "$@" > "$OUT/kt.log" 2>&1 || { tail -5 "$OUT/kt.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, statistics, collections
out = sys.argv[1]
order = [l.split()[1:] for l in open(os.path.join(out, "kt.log")) if l.startswith("order")][0]
f = max(glob.glob(os.path.join(out, "kt", "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
slab = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "rq_slab_kernel" in r["Kernel_Name"]]
slab = slab[-len(order):]
per = collections.defaultdict(list)
for s, t in zip(order, slab):
    per[s].append(t)
for s in dict.fromkeys(order):
    print(f"dbg {s:>8s}: median {statistics.median(per[s]):8.1f} us   min {min(per[s]):8.1f}   ({10.88e3 / statistics.median(per[s]):.3f} TB/s)   all {[round(x) for x in per[s]]}")
PY
find "$OUT" -name '*_kernel_trace.csv' -delete
