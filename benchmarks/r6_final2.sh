#!/bin/bash
# Round 6, after the last KERNEL change (key-split d = 64 attention): the -m gpu suite, the hash-guarded HBM traffic record of the conv3x3_*
# launches (two PMC passes), then the bench lines + kernel statistics (benchmarks/r6_bench_lines.sh).  The SQ-counter summaries and the power
# probe of benchmarks/r6_final.sh are of kernels this change did not touch.
O=gpurun_out; T=${1:-r6}; export TMPDIR=/tmp; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $O/${T}_gputests_final_tree.log 2>&1
tail -14 $O/${T}_gputests_final_tree.log
CMD="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-latency --no-f32 --no-modes --no-power"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${T}_fetch -o fetch -- $CMD > /dev/null 2> $O/${T}_fetch.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${T}_write -o write -- $CMD > /dev/null 2> $O/${T}_write.err
F=$(find $O/${T}_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/${T}_write -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py $F $W conv3x3_ --batch 8 --dtype bf16 --size 512 --source-hash $(python -c "import bench; print(bench.source_hash())") \
    --collected "$T: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of '$CMD'" > $O/${T}_traffic_conv3x3.json
cp $O/${T}_traffic_conv3x3.json profiles/traffic_conv3x3.json
cat $O/${T}_traffic_conv3x3.json
bash benchmarks/r6_bench_lines.sh $T
