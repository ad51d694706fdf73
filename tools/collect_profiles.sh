#!/bin/bash
# Collect the per-round measurement evidence on a GPU box (run from the repo root):
#   bash tools/collect_profiles.sh gpurun_out/<dir>      then, back in the container,
#   python tools/summarize_profiles.py gpurun_out/<dir> profiles/<round_dir>
# One --kernel-trace --stats pass, then FOUR separate --pmc passes (counters are never combined with
# any trace domain other than the kernel trace), then a default bench.py run.
set -o pipefail
OUT=${1:?output directory}
export TMPDIR=/tmp
mkdir -p "$OUT"
WARMUP=5
B="python3 bench.py --steps 50 --warmup $WARMUP --no-cpu-baseline --no-s60 --no-search"
# which box the figures belong to (the commit is recorded by the summariser: the snapshot on the box has no .git)
{ hostname; date -u +%FT%TZ; rocm-smi --showuniqueid --showproductname 2>/dev/null | grep -i "GPU\[" ; } > "$OUT/box.txt" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- $B > "$OUT/kt.log" 2>&1 &&
# medians over the TIMED launches, computed here because the trace itself is too large to travel back
python3 tools/kernel_trace_medians.py "$OUT/kt" "$OUT/kt_medians.json" --skip-first "path_kernel<0, 0=$WARMUP" > "$OUT/kt_medians.txt" &&
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/pmc_sq" -- $B > "$OUT/pmc_sq.log" 2>&1 &&
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU \
    --output-format csv -d "$OUT/pmc_mix" -- $B > "$OUT/pmc_mix.log" 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $B > "$OUT/pmc_fetch.log" 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $B > "$OUT/pmc_write.log" 2>&1 &&
python3 bench.py 2> "$OUT/bench_default.err" | grep '"metric"' > "$OUT/bench_default.json" &&
# keep only what the summariser reads (the merge-back limit is 64 MiB)
find "$OUT" -name '*_kernel_trace.csv' -delete &&
echo "profiles collected in $OUT"
