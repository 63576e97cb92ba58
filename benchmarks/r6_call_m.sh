#!/bin/bash
# Round-6 call M: batch 1 -- (1) more slices for the 128-tile long-K wide-GEMM convs, (2) one 256-row tile per column block for the 256-row
# weight-streaming 3x3 convs of the LDS-DMA igemm, (3) the 64 x 32 tile up to K = 1280 where it gives >= 512 workgroups.
O=gpurun_out; T=r6m; export TMPDIR=/tmp; mkdir -p $O
python benchmarks/ab.py --arms - I2I_W32_SPLITK_LONGK_SK=3 I2I_W32_SPLITK_LONGK_SK=4 I2I_DMA_TALL_ROWS=256 I2I_SMALL_TILE_K2=1280 --repeats 6 --steps 20 --batch 1 --out $O/${T}_ab_bs1.json 2>&1 | grep -v amdgpu.ids | tee $O/${T}_ab_bs1_longk_sk_dma_tall_small_tile_k2.log
I2I_DMA_TALL_ROWS=256 I2I_SMALL_TILE_K2=1280 python bench.py --batch 1 --steps 50 --warmup 10 --no-cpu-baseline --no-f32 --no-modes --no-latency --no-calib --per-op $O/${T}_per_op_bs1_tall_k2.txt > $O/${T}_bench_bs1_tall_k2.json 2> $O/${T}_bench.err
grep -E "up_blocks.1.resnets.1.conv|down_blocks.2.resnets.1.conv|up_blocks.3.attentions.1.*ff.net.2" $O/${T}_per_op_bs1_tall_k2.txt
