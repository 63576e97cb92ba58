#!/bin/bash
# round-4 final call: the whole -m gpu suite on the final tree, then the evidence set (benchmarks/final_round.sh r4)
O=gpurun_out; export TMPDIR=/tmp; mkdir -p $O
timeout 1100 python -m pytest tests -m gpu -x -q --durations=15 > $O/r4_gputests_full.log 2>&1; tail -22 $O/r4_gputests_full.log
bash benchmarks/final_round.sh r4 > $O/r4_final_round.log 2>&1; tail -40 $O/r4_final_round.log
