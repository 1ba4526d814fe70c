#!/bin/bash
# round 3 A/B on one box: tools/r3_ab.sh <outdir> <label>=<lib.so>[,ENV=VAL...] ...   (each arm twice, alternating)
set -o pipefail
OUT=gpurun_out/$1; shift; mkdir -p $OUT
for rep in 1 2; do
for arm in "$@"; do
  label=${arm%%=*}; rest=${arm#*=}; lib=${rest%%,*}; envs=""
  if [[ "$rest" == *,* ]]; then envs=$(echo "${rest#*,}" | tr ',' ' '); fi
  env SCAPE_HIP_LIB=$lib $envs timeout -k 10 300 python bench.py --e2e-utrs 0 --no-cpu-baseline --steps 2 2>$OUT/bench_$label.err | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];k=d['kernels_ms']
print('$label', 'value %.0f' % d['value'], 'step %.1f ms' % d['ms_per_step'], 'sweep %.1f' % r['em_sweep_ms'], 'mstep %.3f ms/launch' % r['launch_ms'], 'frac %.3f' % r['frac'], 'estep %.1f ms/sweep' % k['k2_estep_profiled_step']['ms_total'], 'phaseB %.1f' % (k['phase_b']['ms_total']/k['phase_b']['launches']))" | tee -a $OUT/summary.txt
done; done
