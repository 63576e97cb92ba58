#!/bin/bash
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r3a_gputests.log 2>&1; echo "rc $?" >> $O/r3a_gputests.log
timeout 600 python bench.py --per-op $O/r3a_per_op_bs8.txt > $O/r3a_bench_bs8.json 2> $O/r3a_bench_bs8.err
S="vae 128->128@512 gn,vae 256->256@256 gn,vae 512->512@128 gn"
timeout 300 python benchmarks/bench_ops.py --lib img2img-turbo_amd/csrc/libi2i_turbo_trace.so --trace --only "$S" --tiles 0 --iters 5 --out $O/r3a_trace.json > $O/r3a_w32_trace_segments.log 2>&1
timeout 300 python benchmarks/bench_ops.py --lib img2img-turbo_amd/csrc/libi2i_turbo_trace.so --trace --nogn --only "$S" --tiles 0 --iters 5 --out $O/r3a_trace2.json >> $O/r3a_w32_trace_segments.log 2>&1
timeout 300 python benchmarks/bench_ops.py --lib img2img-turbo_amd/csrc/libi2i_turbo_trace.so --trace --splitk 8 --only "$S" --tiles 0 --iters 5 --out $O/r3a_trace3.json >> $O/r3a_w32_trace_segments.log 2>&1
timeout 700 bash benchmarks/pmc_conv.sh $O/r3a_pmc "vae 128->128@512 gn,vae 512->512@128 gn" > /dev/null 2>&1
python tools/pmc_summary.py $(find $O/r3a_pmc/sq1 -name "*counter_collection.csv" | head -1) $(find $O/r3a_pmc/sq2 -name "*counter_collection.csv" | head -1) conv3x3_w32_kernel > $O/r3a_pmc_conv3x3_w32_summary.txt 2>&1
tail -4 $O/r3a_gputests.log; cat $O/r3a_bench_bs8.json; tail -3 $O/r3a_bench_bs8.err; cat $O/r3a_w32_trace_segments.log; cat $O/r3a_pmc_conv3x3_w32_summary.txt
