#!/bin/bash
# Round-5 closing call: the whole -m gpu suite (with the slowest tests listed), then the evidence set of the final tree.
O=gpurun_out; mkdir -p $O
timeout 1150 python -m pytest tests -m gpu -x -q --durations=12 > $O/r5_gputests_final_tree.log 2>&1
tail -25 $O/r5_gputests_final_tree.log
bash benchmarks/final_round.sh r5 > $O/r5_final_round.log 2>&1
tail -60 $O/r5_final_round.log
