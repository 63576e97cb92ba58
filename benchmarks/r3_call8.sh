#!/bin/bash
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
S="vae 128->128@512 gn,vae 512->512@128 gn"
timeout 600 python benchmarks/bench_ops.py --lib img2img-turbo_amd/csrc/libi2i_turbo_trace.so --trace --splitk 0,4,16,32,64,8,12,60,124 --only "$S" --tiles 0 --iters 5 --out $O/r3g_trace.json > $O/r3g_w32_ablation.log 2>&1
grep -v amdgpu.ids $O/r3g_w32_ablation.log | cut -c1-330
