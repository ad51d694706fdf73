#!/bin/bash
# A/B of compile-time variants of the library (csrc/libmcr_hip_<variant>.so, built with csrc/build.py build(variant=...)):
# K3 timing of every variant (steady state: tools/k3_series.py), interleaved over a few rounds on one box.
#   bash tools/k3_ab_libs.sh "" lut256 lut512
for round in 1 2; do
  for v in "$@"; do
    if [ -z "$v" ]; then lib=""; else lib="$PWD/monte_carlo_retirement_amd/csrc/libmcr_hip_$v.so"; fi
    printf "round $round variant '%s': " "$v"; MCR_HIP_LIBRARY=$lib python tools/k3_series.py 2>/dev/null | head -1 | python3 -c "import sys,ast,statistics; t=ast.literal_eval(sys.stdin.read()); print('steady median %.3f ms  min %.3f' % (statistics.median(t[10:]), min(t)))"
  done
done
