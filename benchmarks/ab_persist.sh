#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/ab_persist.txt
: > $O
timeout 400 python -m pytest tests/test_ops_gpu.py -x -q -k "dma or igemm or geglu or linear" >> $O 2>&1
for w in 0 default; do
  echo "== tiles=20 PERSIST_WGS=$w" >> $O
  if [ $w = default ]; then unset I2I_PERSIST_WGS; else export I2I_PERSIST_WGS=$w; fi
  timeout 200 python benchmarks/bench_ops.py --only "lin,skip,conv_in,down" --tiles 20 --nogn --out gpurun_out/ab_$w.json >> $O 2>&1
done
unset I2I_PERSIST_WGS
echo "== bench default" >> $O
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline >> $O 2>&1
