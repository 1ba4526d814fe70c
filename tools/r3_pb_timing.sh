#!/bin/bash
# Phase A / B time per launch of one or more builds (e.g. timing-only builds, whose results are wrong): tools/r3_pb_timing.sh <outdir> <label>=<lib.so> ...
set -o pipefail
OUT=gpurun_out/$1; shift; mkdir -p $OUT
for arm in "$@"; do
  label=${arm%%=*}; lib=${arm#*=}
  SCAPE_HIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py --e2e-utrs 0 --no-cpu-baseline --no-other-configs --no-cli-leg --steps 2 2>$OUT/bench_$label.err | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['kernels_ms']
print('$label', 'phaseB %.2f ms' % (k['phase_b']['ms_total']/k['phase_b']['launches']), 'phaseA %.2f ms' % (k['phase_a']['ms_total']/k['phase_a']['launches']), 'step %.1f' % d['ms_per_step'])" | tee -a $OUT/summary.txt
done
