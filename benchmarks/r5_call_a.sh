#!/bin/bash
# Round-5 first GPU call: this round's same-box baseline (bench line + per-op table), the segment trace of the wide-tile 3x3
# conv (csrc/build.py --tag trace --defs=-DI2I_TRACE=1), one PMC pass over the attention kernels, and where the GPU test
# suite spends its wall time ([phase] lines of tests/test_e2e_gpu.py).
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
python bench.py --per-op $O/r5a_per_op_bs8.txt --no-cpu-baseline > $O/r5a_bench_bs8.json 2> $O/r5a_bench_bs8.err
LIBT=img2img-turbo_amd/csrc/libi2i_turbo_trace.so
if [ -f $LIBT ]; then
  for extra in "" "--res"; do
    timeout 120 python benchmarks/bench_ops.py --lib $LIBT --trace --tiles 0 --iters 3 --only "vae 128->128@512 gn,vae 256->256@256 gn,vae 512->512@128 gn" $extra --out $O/r5a_trace.json
  done > $O/r5a_w32_trace.log 2>&1
fi
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d $O/r5a_pmc_att -o att -- python benchmarks/bench_attention.py > $O/r5a_pmc_att.log 2>&1
python tools/pmc_summary.py $(find $O/r5a_pmc_att -name "*counter_collection.csv" | head -1) attention_ > $O/r5a_pmc_attention_summary.txt 2>&1
timeout 400 python -m pytest tests/test_e2e_gpu.py -m gpu -x -q -s -k "cfg2 or full_sd" --durations=10 > $O/r5a_tests.log 2>&1
tail -5 $O/r5a_tests.log; grep phase $O/r5a_tests.log
cat $O/r5a_bench_bs8.json | cut -c1-600
cat $O/r5a_w32_trace.log
cat $O/r5a_pmc_attention_summary.txt
