#!/bin/bash
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python benchmarks/check_w32_gpu.py > $O/r3j_w32_parity.log 2>&1; echo "parity rc $?" >> $O/r3j_w32_parity.log
S="vae 512->512@128 gn,vae 512->512@64 gn"
L=$O/r3j_xcdtn_ab.log; : > $L
for rep in 1 2 3; do
for X in 1 0; do
echo "== I2I_W32_XCDTN=$X rep $rep" >> $L
I2I_W32_XCDTN=$X timeout 300 python benchmarks/bench_ops.py --only "$S" --tiles 0 --iters 9 --out $O/r3j_tmp.json >> $L 2>&1
done; done
tail -2 $O/r3j_w32_parity.log; grep -v "amdgpu.ids\|n/a" $L
