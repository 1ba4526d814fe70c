#!/bin/bash
# per-round M-step launch times (no byte tally) for several builds: tools/r3_plain_rounds.sh <outdir> <label>=<lib.so> ...
OUT=gpurun_out/$1; shift; mkdir -p $OUT
for arm in "$@"; do
  label=${arm%%=*}; lib=${arm#*=}
  SCAPE_HIP_LIB=$PWD/$lib timeout -k 10 200 python tools/trace_rounds.py > $OUT/$label.out 2> $OUT/$label.err
  echo "$label mstep ms per round:"; grep "kind 5" $OUT/$label.err | head -50 | awk '{printf "%s ", $6}'; echo
  echo "$label estep ms per round:"; grep "kind 4" $OUT/$label.err | head -51 | awk '{printf "%s ", $6}'; echo
done
