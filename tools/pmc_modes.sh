#!/bin/bash
# PMC counters of the K3 slab kernel over several re-allocations of the slab (tools/k3_alloc_modes.py): what differs between
# the fast and the slow allocations?  Separate --pmc passes (never combined with a trace domain other than the kernel trace).
set -o pipefail
OUT=${1:?output directory}
export TMPDIR=/tmp
mkdir -p "$OUT"
i=0
for set in "GRBM_GUI_ACTIVE TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum GRBM_UTCL2_BUSY" \
           "GRBM_GUI_ACTIVE TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_TAG_STALL_sum TCP_PENDING_STALL_CYCLES_sum" \
           "GRBM_GUI_ACTIVE TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc$i" -- python3 tools/k3_alloc_modes.py 10000000 ${CYCLES:-6} > "$OUT/pmc$i.log" 2>&1 || { tail -5 "$OUT/pmc$i.log"; exit 1; }
  python3 - "$OUT/pmc$i" <<'PY'
import csv, glob, os, sys, statistics, collections
f = max(glob.glob(os.path.join(sys.argv[1], "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
# dispatches in order; a path_kernel dispatch starts a new allocation cycle
order = {}
for r in rows:
    order.setdefault(int(r["Dispatch_Id"]), r["Kernel_Name"])
cycle_of, cycle = {}, -1
for d in sorted(order):
    if "path_kernel" in order[d]:
        cycle += 1
    cycle_of[d] = cycle
per = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if "rq_slab_kernel" in r["Kernel_Name"]:
        per[cycle_of[int(r["Dispatch_Id"])]][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for v in per.values() for c in v})
print("cycle  " + "  ".join(f"{n[:34]:>34s}" for n in names))
for c in sorted(per):
    print(f"{c:5d}  " + "  ".join(f"{statistics.median(per[c][n][6:]):34.4g}" for n in names))
PY
  find "$OUT/pmc$i" -name '*.csv' -size +20M -delete
done
