#!/bin/bash
# E-step / M-step time per sweep at the other BASELINE shapes, two builds side by side: tools/ab_shapes.sh <libA.so> <libB.so>
for SHAPE in "--reads 500 --kcap 5 --utrs 2048" "--reads 5000 --kcap 12 --utrs 256" "--reads 10000 --kcap 10 --utrs 256"; do
  for L in "$@"; do
    SCAPE_HIP_LIB=$L python bench.py --e2e-utrs 0 --no-cpu-baseline --steps 2 $SHAPE 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];k=d['kernels_ms']
jr=r['em_rounds_per_sweep']
print('$SHAPE', '$L', 'value %.0f' % d['value'], 'step %.1f ms' % d['ms_per_step'], 'estep %.1f ms/sweep = %.1f ns per job-round' % (k['k2_estep_profiled_step']['ms_total'], 1e6*k['k2_estep_profiled_step']['ms_total']/jr), 'mstep %.3f ms/launch' % r['launch_ms'], 'frac %.3f' % r['frac'], 'phaseB %.1f' % (k['phase_b']['ms_total']/k['phase_b']['launches']))"
  done
done
