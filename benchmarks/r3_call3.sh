#!/bin/bash
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python benchmarks/check_w32_gpu.py > $O/r3d_w32_parity.log 2>&1; echo "parity rc $?" >> $O/r3d_w32_parity.log
S="vae 128->128@512,vae 256->256@256,vae 512->512@128,vae 512->512@64,vae 256->128@512"
timeout 600 python benchmarks/bench_ops.py --only "$S" --tiles 0,13 --iters 7 --out $O/r3d_ab_gn.json > $O/r3d_w32_ab_gn.log 2>&1
timeout 600 python bench.py --per-op $O/r3d_per_op_bs8.txt --no-cpu-baseline > $O/r3d_bench_bs8.json 2> $O/r3d_bench_bs8.err
S2="vae 128->128@512 gn,vae 256->256@256 gn,vae 512->512@128 gn"
timeout 300 python benchmarks/bench_ops.py --lib img2img-turbo_amd/csrc/libi2i_turbo_trace.so --trace --only "$S2" --tiles 0 --iters 5 --out $O/r3d_trace.json > $O/r3d_w32_trace_segments.log 2>&1
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_e2e_gpu.py -x -q -k "w32 or cfg2" > $O/r3d_e2e.log 2>&1
tail -3 $O/r3d_w32_parity.log; cat $O/r3d_w32_ab_gn.log; python - <<'PY'
import json
r=json.load(open('gpurun_out/r3d_bench_bs8.json'))
print(r['value'],'img/s',r['ms_per_step'],'ms frac',r['roofline']['frac'],r['roofline']['per_kernel'],r['parity_max_abs'] if 'parity_max_abs' in r else '')
print({k:v['ms'] for k,v in r['kernel_breakdown_ms'].items()})
PY
tail -2 $O/r3d_bench_bs8.err; cat $O/r3d_w32_trace_segments.log; tail -3 $O/r3d_e2e.log
