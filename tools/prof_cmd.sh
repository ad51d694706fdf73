#!/bin/bash
# rocprofv3 --kernel-trace --stats of one command; prints the per-kernel summary (top 25 by total time).
#   bash tools/prof_cmd.sh gpurun_out/<dir> python3 tools/k3_time.py 10000000
set -o pipefail
OUT=${1:?output directory}; shift
export TMPDIR=/tmp
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- "$@" > "$OUT/kt.log" 2>&1 || { tail -5 "$OUT/kt.log"; exit 1; }
tail -2 "$OUT/kt.log"
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
fs = glob.glob(os.path.join(sys.argv[1], "kt", "**", "*_kernel_stats.csv"), recursive=True)
if not fs:
    sys.exit("no kernel_stats.csv")
f = max(fs, key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:25]:
    print(f'{r["Name"][:90]:90s} calls {int(r["Calls"]):5d} avg {float(r["AverageNs"])/1e3:10.1f} us  total {float(r["TotalDurationNs"])/1e6:9.3f} ms')
import shutil
shutil.copy(f, os.path.join(sys.argv[1], "kernel_stats.csv"))
PY
find "$OUT" -name '*_kernel_trace.csv' -delete
