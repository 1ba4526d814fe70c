#!/bin/bash
# rocprofv3 evidence for the bench line (run on the GPU box from the repo root):  tools/collect_profiles.sh <tag> [pmc counters...]
# writes profiles/<tag>_kernel_stats.csv and profiles/<tag>_pmc_<counter>.csv (+ raw output under gpurun_out/<tag>/)
set -e -o pipefail
TAG=${1:?tag}; shift || true
ROOT=$(pwd)   # summaries are also written under gpurun_out/<tag>/ (gpurun merges only gpurun_out/ back)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT" "$ROOT/profiles"
export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --e2e-utrs 0 --no-other-configs"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/stats_bench.json" 2> "$OUT/stats.err" )
python3 tools/rocprof_summary.py stats "$OUT/stats" "profiles/${TAG}_kernel_stats.csv" "python3 bench.py $ARGS"
cp "$OUT/stats_bench.json" "profiles/${TAG}_bench_under_rocprof.json"
ARGS1="--steps 1 --warmup 0 --no-cpu-baseline --e2e-utrs 0 --no-other-configs"
for C in "$@"; do
  N=$(echo "$C" | tr ' ,' '__')
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_$N" -- python3 "$ROOT/bench.py" $ARGS1 > "$OUT/pmc_$N.json" 2> "$OUT/pmc_$N.err" )
  python3 tools/rocprof_summary.py pmc "$OUT/pmc_$N" "profiles/${TAG}_pmc_${N}.csv" "python3 bench.py $ARGS1"
done
cp profiles/${TAG}_* "$OUT/" 2>/dev/null || true
head -8 "profiles/${TAG}_kernel_stats.csv"
