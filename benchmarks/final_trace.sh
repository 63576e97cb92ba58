#!/bin/bash
# Kernel-trace stats of the bench command + the bench line itself (PMC traffic passes: benchmarks/final_profile.sh).
OUT=${1:-gpurun_out/final}; export TMPDIR=/tmp; mkdir -p $OUT
CMD="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-latency"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.json 2> $OUT/trace.err
python bench.py --per-op $OUT/per_op.txt > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
head -12 $OUT/trace/trace_kernel_stats.csv 2>/dev/null | cut -c1-160
