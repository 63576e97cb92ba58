#!/bin/bash
# A/B: persistent K-step stream in igemm_dma (I2I_PERSIST_WGS=0 disables it)
mkdir -p gpurun_out
O=gpurun_out/ab_persist.txt
: > $O
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -k "dma or igemm or geglu or linear" >> $O 2>&1
for tiles in 20; do
for w in 0 default 512; do
  echo "== tiles=$tiles PERSIST_WGS=$w" >> $O
  if [ $w = default ]; then unset I2I_PERSIST_WGS; else export I2I_PERSIST_WGS=$w; fi
  timeout 200 python benchmarks/bench_ops.py --only "lin,skip,conv_in,down" --tiles $tiles --nogn --out gpurun_out/ab_$w.json >> $O 2>&1
done
done
export I2I_PERSIST_WGS=0
echo "== bench PERSIST_WGS=0" >> $O
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline >> $O 2>&1
