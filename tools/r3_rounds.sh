#!/bin/bash
# per-round M-step table for one library build: tools/r3_rounds.sh <outdir> <lib.so>
OUT=gpurun_out/$1; mkdir -p $OUT
export SCAPE_HIP_LIB=$2
python tools/trace_rounds.py > $OUT/trace_plain.out 2> $OUT/trace_plain.err
python tools/trace_rounds.py bytes > $OUT/trace_bytes.out 2> $OUT/trace_bytes.err
python tools/trace_rounds_table.py $OUT/trace_plain.err $OUT/trace_bytes.err > $OUT/mstep_per_round.txt
