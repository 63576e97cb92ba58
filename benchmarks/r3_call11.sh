#!/bin/bash
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python benchmarks/check_w32_gpu.py > $O/r3i_w32_parity.log 2>&1; echo "parity rc $?" >> $O/r3i_w32_parity.log
S="vae 128->128@512 gn,vae 512->512@128 gn,vae 256->256@256 gn,vae 256->128@512 gn,vae 512->512@64 gn"
L=$O/r3i_ab.log; : > $L
for rep in 1 2; do
echo "== simple + staged epilogue (product) rep $rep" >> $L
timeout 300 python benchmarks/bench_ops.py --only "$S" --tiles 0,13 --iters 7 --out $O/r3i_tmp.json >> $L 2>&1
echo "== v2 rep $rep" >> $L
timeout 300 python benchmarks/bench_ops.py --lib img2img-turbo_amd/csrc/libi2i_turbo_w32v2.so --only "$S" --tiles 40 --iters 7 --out $O/r3i_tmp.json >> $L 2>&1
done
timeout 600 python bench.py --per-op $O/r3i_per_op_bs8.txt --no-cpu-baseline > $O/r3i_bench_bs8.json 2> $O/r3i_bench_bs8.err
tail -2 $O/r3i_w32_parity.log; grep -v "amdgpu.ids\|n/a" $L
python - <<'PY'
import json
r=json.load(open('gpurun_out/r3i_bench_bs8.json'))
print(r['value'],'img/s',r['ms_per_step'],'ms frac',r['roofline']['frac'],r['roofline']['per_kernel'],r.get('parity_max_abs'))
print({k:v['ms'] for k,v in r['kernel_breakdown_ms'].items()})
PY
