#!/bin/bash
# Round-6 call K: (1) both sources of a concatenated resnet input through ONE gn_apply launch (ABI v10); (2) at batch 1, which kernel should take
# the UNet's 3x3 convolutions?  I2I_W32_SPLITK mask bits: 1 stride-2 on the wide GEMM, 2 16x16 planes, 4 planes under halo_min_tiles, 8 the
# 32x32 / 64x64 planes (instead of the halo conv, which applies GroupNorm + SiLU itself).
O=gpurun_out; T=r6k; export TMPDIR=/tmp; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gn_apply or gn_stats" 2>&1 | tail -4 | tee $O/${T}_op_tests.log
python benchmarks/ab.py --arms I2I_GN_APPLY_ONE=0 - --repeats 6 --steps 10 --batch 8 --out $O/${T}_ab_bs8.json 2>&1 | grep -v amdgpu.ids | tee $O/${T}_ab_bs8_gn_apply_one.log
python benchmarks/ab.py --arms I2I_GN_APPLY_ONE=0 - I2I_W32_SPLITK=7 I2I_W32_SPLITK=13 I2I_W32_SPLITK=5 I2I_W32_SPLITK=11 --repeats 6 --steps 20 --batch 1 --out $O/${T}_ab_bs1.json 2>&1 | grep -v amdgpu.ids | tee $O/${T}_ab_bs1_gn_apply_one_w32_splitk_mask.log
timeout 600 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu -k "full_sd_turbo or plan_file or cfg4" 2>&1 | tail -4 | tee $O/${T}_e2e_subset.log
