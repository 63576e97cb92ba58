#!/bin/bash
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
S="vae 128->128@512 gn,vae 512->512@128 gn,vae 256->256@256 gn"
L=$O/r3h_v2_vs_v3.log; : > $L
for rep in 1 2; do
echo "== v3 (product) rep $rep" >> $L
timeout 300 python benchmarks/bench_ops.py --only "$S" --tiles 0,13 --iters 7 --out $O/r3h_tmp.json >> $L 2>&1
echo "== v3 nopin rep $rep" >> $L
timeout 300 python benchmarks/bench_ops.py --lib img2img-turbo_amd/csrc/libi2i_turbo_nopin.so --only "$S" --tiles 0 --iters 7 --out $O/r3h_tmp.json >> $L 2>&1
echo "== v2 (non-persistent, commit 15e7833) rep $rep" >> $L
timeout 300 python benchmarks/bench_ops.py --lib img2img-turbo_amd/csrc/libi2i_turbo_w32v2.so --only "$S" --tiles 41,42 --iters 7 --out $O/r3h_tmp.json >> $L 2>&1
done
echo "== nogn: v3, v2" >> $L
timeout 300 python benchmarks/bench_ops.py --only "$S" --nogn --tiles 0 --iters 7 --out $O/r3h_tmp.json >> $L 2>&1
timeout 300 python benchmarks/bench_ops.py --lib img2img-turbo_amd/csrc/libi2i_turbo_w32v2.so --only "$S" --nogn --tiles 41,42 --iters 7 --out $O/r3h_tmp.json >> $L 2>&1
grep -v "amdgpu.ids\|n/a" $L
