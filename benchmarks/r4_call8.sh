#!/bin/bash
# round-4 call 8: split-K on the wide GEMM for the UNet's small-plane 3x3 convolutions: parity on hardware + (tile, split) sweep
O=gpurun_out; export TMPDIR=/tmp; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "gemm_w32_conv3x3" > $O/r4h_gputests_ops.log 2>&1; tail -3 $O/r4h_gputests_ops.log
python benchmarks/bench_ops.py --nogn --tiles 20 --splitk 0,4,6,8 --only "unet 1280->1280@8,unet 2560->1280@8,unet 1280->1280@16,unet down" --out $O/r4h_dma.json 2>&1 | grep -v "amdgpu\|n/a" > $O/r4h_bench_ops_splitk_dma.log; cat $O/r4h_bench_ops_splitk_dma.log
python benchmarks/bench_ops.py --nogn --tiles 51,52,53,54 --splitk 0,4,6,8,12,16 --only "unet 1280->1280@8,unet 2560->1280@8,unet 1280->1280@16,unet down" --out $O/r4h_w32.json 2>&1 | grep -v "amdgpu\|n/a" > $O/r4h_bench_ops_splitk_w32.log; cat $O/r4h_bench_ops_splitk_w32.log
