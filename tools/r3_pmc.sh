#!/bin/bash
# rocprofv3 counter passes on one EM sweep workload: tools/r3_pmc.sh <tag> <lib.so> "<counters pass 1>" ["<counters pass 2>" ...]
set -o pipefail
TAG=$1; LIB=$2; shift 2
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp SCAPE_HIP_LIB=$ROOT/$LIB
i=0
for C in "$@"; do
  i=$((i+1))
  ( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pass$i" -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --e2e-utrs 0 --no-other-configs > "$OUT/pass$i.json" 2> "$OUT/pass$i.err" ) || { echo "pass $i failed"; tail -5 "$OUT/pass$i.err"; }
  python3 tools/pmc_rounds.py "$OUT/pass$i" mstep 50 > "$OUT/pass${i}_mstep.txt"
  python3 tools/pmc_rounds.py "$OUT/pass$i" k2_estep 51 > "$OUT/pass${i}_estep.txt"
  python3 tools/pmc_rounds.py "$OUT/pass$i" k_phase_b 4 > "$OUT/pass${i}_phaseb.txt"
  python3 tools/rocprof_summary.py pmc "$OUT/pass$i" "$OUT/pass${i}_summary.csv" "python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --e2e-utrs 0 --no-other-configs"
  find "$OUT/pass$i" -name "*.csv" -size +20M -delete
done
