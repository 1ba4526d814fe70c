#!/bin/bash
# small-call (reference-stream, one UTR per call) kernel times for several builds: tools/ab_small.sh <lib.so> ...
for L in "$@"; do
  echo "== $L"
  SCAPE_HIP_LIB=$L python tools/reference_mode_breakdown.py 2>&1 | grep -E "wall per UTR|k2_estep|k2_mstep"
done
