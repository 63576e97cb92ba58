#!/bin/bash
# Round profile set: kernel-trace stats + PMC traffic of the SAME bench command, then the bench line itself.
# Usage (on the GPU box, from the repo root): benchmarks/final_profile.sh gpurun_out/final
OUT=${1:-gpurun_out/final}; export TMPDIR=/tmp; mkdir -p $OUT
CMD="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-latency"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- $CMD > $OUT/fetch.json 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- $CMD > $OUT/write.json 2> $OUT/write.err
python tools/pmc_traffic.py $OUT/fetch/fetch_counter_collection.csv $OUT/write/write_counter_collection.csv conv3x3_halo_kernel --batch 8 --dtype bf16 > $OUT/r1_traffic_conv3x3_halo.json
mkdir -p profiles; cp $OUT/r1_traffic_conv3x3_halo.json profiles/r1_traffic_conv3x3_halo.json
python bench.py --per-op $OUT/per_op.txt > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
ls $OUT/trace | head; head -5 $OUT/trace/trace_kernel_stats.csv 2>/dev/null | cut -c1-200
