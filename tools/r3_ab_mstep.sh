#!/bin/bash
# round 3: parity + A/B of the register-staged (v2) and LDS-DMA ring (v3) M-step on one box
set -o pipefail
OUT=gpurun_out/${1:-r3a}; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" | tee -a $OUT/summary.txt
tail -3 $OUT/pytest.log | tee -a $OUT/summary.txt
for rep in 1 2; do
for m in v2 v3; do
  SCAPE_HIP_MSTEP=$m timeout -k 10 300 python bench.py --e2e-utrs 0 --no-cpu-baseline --steps 2 2>$OUT/bench_$m.err | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];k=d['kernels_ms']
print('$m', 'value %.0f' % d['value'], 'step %.1f ms' % d['ms_per_step'], 'sweep %.1f' % r['em_sweep_ms'], 'mstep %.3f ms/launch' % r['launch_ms'], 'frac %.3f' % r['frac'], 'estep %.1f ms/sweep' % k['k2_estep_profiled_step']['ms_total'], 'phaseB %.1f' % (k['phase_b']['ms_total']/k['phase_b']['launches']))" | tee -a $OUT/summary.txt
done; done
python tools/trace_rounds.py > $OUT/trace_plain.out 2> $OUT/trace_plain.err
python tools/trace_rounds.py bytes > $OUT/trace_bytes.out 2> $OUT/trace_bytes.err
python tools/trace_rounds_table.py $OUT/trace_plain.err $OUT/trace_bytes.err > $OUT/mstep_per_round.txt
head -30 $OUT/mstep_per_round.txt
