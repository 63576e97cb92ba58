#!/bin/bash
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python benchmarks/check_w32_gpu.py > $O/r3f_w32_parity.log 2>&1; echo "parity rc $?" >> $O/r3f_w32_parity.log
S="vae 128->128@512 gn,vae 512->512@128 gn,vae 256->256@256 gn,vae 256->128@512 gn"
timeout 300 python benchmarks/bench_ops.py --only "$S" --tiles 0,13 --iters 7 --out $O/r3f_tmp.json > $O/r3f_ab.log 2>&1
timeout 300 python benchmarks/bench_ops.py --only "$S" --nogn --tiles 0,13 --iters 7 --out $O/r3f_tmp.json >> $O/r3f_ab.log 2>&1
tail -2 $O/r3f_w32_parity.log; grep -v amdgpu.ids $O/r3f_ab.log
