#!/bin/bash
# Round-6 closing call: the whole -m gpu suite (slowest tests listed), the evidence set of the final tree (benchmarks/final_round.sh), and
# the SQ counters of the reworked d = 64 flash attention (two passes: the wave-cycle split, and the matrix-pipe share against GRBM_GUI_ACTIVE).
O=gpurun_out; T=${1:-r6}; export TMPDIR=/tmp; mkdir -p $O
python -c "import importlib.util as u; print('diffusers importable on this box:', u.find_spec('diffusers') is not None, ' peft:', u.find_spec('peft') is not None)" > $O/${T}_reference_packages.log 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 > $O/${T}_gputests_final_tree.log 2>&1
tail -25 $O/${T}_gputests_final_tree.log
bash benchmarks/final_round.sh $T > $O/${T}_final_round.log 2>&1
tail -70 $O/${T}_final_round.log
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \
    --output-format csv -d $O/${T}_pmc_att1 -o att -- python benchmarks/bench_attention.py > $O/${T}_pmc_att.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES \
    --output-format csv -d $O/${T}_pmc_att2 -o att -- python benchmarks/bench_attention.py >> $O/${T}_pmc_att.log 2>&1
python tools/pmc_summary.py $(find $O/${T}_pmc_att1 $O/${T}_pmc_att2 -name "*counter_collection.csv" | sort | tr '\n' ' ') attention_dma > $O/${T}_pmc_attention_summary.txt 2>&1
cat $O/${T}_pmc_attention_summary.txt
# gpurun merges at most 64 MiB back: the raw rocprofv3 directories (kernel traces, counter csv files of whole bench runs) stay on the box,
# their summaries above are what is kept
find $O -mindepth 1 -maxdepth 1 -type d -name "${T}_*" -exec rm -rf {} +
find $O -type f -size +3M -delete
du -sh $O
