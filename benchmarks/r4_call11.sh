#!/bin/bash
# round-4 call 11: would the UNet's 64x64 / 32x32-plane 3x3 convs (halo conv, 650-750 TFLOP/s) do better on the wide GEMM?  + bs=1 A/B of the K-slice route
O=gpurun_out; export TMPDIR=/tmp; mkdir -p $O
SH="unet 320->320@64 gn,unet 640->640@32 gn,unet 960->320@64 gn,unet 1280->640@32 gn"
python benchmarks/bench_ops.py --tiles 0 --only "$SH" --out $O/r4k_halo.json 2>&1 | grep -v "amdgpu\|n/a" > $O/r4k_bench_ops_unet_planes_halo.log; cat $O/r4k_bench_ops_unet_planes_halo.log
python benchmarks/bench_ops.py --nogn --tiles 51,52,53,54 --splitk 0,2,4 --only "$SH" --out $O/r4k_w32.json 2>&1 | grep -v "amdgpu\|n/a" > $O/r4k_bench_ops_unet_planes_w32.log; cat $O/r4k_bench_ops_unet_planes_w32.log
python benchmarks/ab.py --arms "I2I_W32_SPLITK=7" "I2I_W32_SPLITK=0" --batch 1 --repeats 6 --steps 20 --out $O/r4k_ab_bs1.json > $O/r4k_ab_bs1.log 2>&1; grep -v amdgpu $O/r4k_ab_bs1.log | tail -3
