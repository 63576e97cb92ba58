#!/bin/bash
# PMC passes over the conv3x3 micro-benchmark (one counter set per pass, --kernel-trace only; see
# MI355X_MICROARCH.md "rocprofv3 PMC slots").  Usage: benchmarks/pmc_conv.sh <outdir> "<shape filter>"
#        benchmarks/pmc_conv.sh <outdir> "<shape filter>" "<extra bench_ops.py flags>"   (e.g. "--geglu" for the ff.net.0 GEMMs)
OUT=${1:-gpurun_out/pmc}; ONLY=${2:-"vae 128->128@512 gn,vae 512->512@128 gn"}; EXTRA=${3:-}
export TMPDIR=/tmp; mkdir -p $OUT
run() { name=$1; shift; timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python benchmarks/bench_ops.py --tiles 0 --iters 3 --only "$ONLY" $EXTRA --out $OUT/$name.json > $OUT/$name.log 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
ls -R $OUT | head -40
