#!/bin/bash
# Round-6 call L: batch 1 -- the VAE's 64x64-plane 512 -> 512 convolutions on the wide GEMM (128 tiles x 72 K stages, one slice today) in two slices.
O=gpurun_out; T=r6l; export TMPDIR=/tmp; mkdir -p $O
python benchmarks/ab.py --arms - I2I_W32_SPLITK_LONGK=32 I2I_W32_SPLITK_LONGK=16 --repeats 6 --steps 20 --batch 1 --out $O/${T}_ab_bs1.json 2>&1 | grep -v amdgpu.ids | tee $O/${T}_ab_bs1_w32_longk_two_slices.log
I2I_W32_SPLITK_LONGK=32 python bench.py --batch 1 --steps 50 --warmup 10 --no-cpu-baseline --no-f32 --no-modes --no-latency --no-calib --per-op $O/${T}_per_op_bs1_longk32.txt > $O/${T}_bench_bs1_longk32.json 2> $O/${T}_bench.err
grep -E "mid_block.resnets.0.conv|down_blocks.3.resnets.1.conv" $O/${T}_per_op_bs1_longk32.txt
