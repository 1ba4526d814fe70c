#!/bin/bash
# rocprofv3 kernel-trace summary of the default bench workload: tools/r3_stats.sh <tag> [lib.so]
TAG=$1; ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
[ -n "$2" ] && export SCAPE_HIP_LIB=$ROOT/$2
( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --e2e-utrs 0 --no-other-configs > "$OUT/stats_bench.json" 2> "$OUT/stats.err" )
python3 tools/rocprof_summary.py stats "$OUT/stats" "$OUT/kernel_stats.csv" "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --e2e-utrs 0 --no-other-configs"
find "$OUT/stats" -name "*.csv" -size +20M -delete
head -14 "$OUT/kernel_stats.csv"
