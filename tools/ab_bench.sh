#!/bin/bash
# A/B of several builds of the library on the SAME GPU box: tools/ab_bench.sh <lib.so> [<lib.so> ...]
for L in "$@" "$@"; do
  SCAPE_HIP_LIB=$L python bench.py --e2e-utrs 0 --no-cpu-baseline --steps 2 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];k=d['kernels_ms']
print('$L', 'value %.0f' % d['value'], 'step %.1f ms' % d['ms_per_step'], 'sweep %.1f' % r['em_sweep_ms'], 'mstep %.3f ms/launch' % r['launch_ms'], 'estep %.1f ms/sweep' % k['k2_estep_profiled_step']['ms_total'], 'phaseB %.1f' % (k['phase_b']['ms_total']/k['phase_b']['launches']), 'phaseA %.1f' % (k['phase_a']['ms_total']/k['phase_a']['launches']))"
done
