#!/bin/bash
# per-dispatch durations of the K3 kernels over several re-allocations of the slab (tools/k3_alloc_modes.py), grouped by cycle
set -o pipefail
OUT=${1:?output directory}
export TMPDIR=/tmp
mkdir -p "$OUT"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/kt" -- python3 tools/k3_alloc_modes.py 10000000 ${CYCLES:-6} > "$OUT/kt.log" 2>&1 || { tail -5 "$OUT/kt.log"; exit 1; }
grep cycle "$OUT/kt.log" | cut -c60-
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, statistics, collections
f = max(glob.glob(os.path.join(sys.argv[1], "kt", "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = ("rq_slab_kernel", "rq_count_kernel<16, false>", "rq_collect", "rq_tiny", "rq_select", "rq_refine")
per = collections.defaultdict(list)
cycle = -1
for r in rows:
    k = r["Kernel_Name"]
    if "path_kernel" in k:
        cycle += 1
    for nm in names:
        if nm in k:
            per[(cycle, nm)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for c in range(cycle + 1):
    print("cycle", c, "  ".join(f"{nm.split('<')[0][3:]} {statistics.median(per[(c, nm)][8:]):.1f}" for nm in names if per.get((c, nm))))
PY
find "$OUT" -name '*_kernel_trace.csv' -delete
