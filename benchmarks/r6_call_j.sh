#!/bin/bash
# Round-6 call J: the plain short-K projections of the wide GEMM on its two-workgroups-per-CU form (I2I_G32_SHORTK = K limit), batch 8 and 1.
O=gpurun_out; T=r6j; export TMPDIR=/tmp; mkdir -p $O
python benchmarks/ab.py --arms - I2I_G32_SHORTK=320 I2I_G32_SHORTK=640 I2I_G32_SHORTK=1280 --repeats 6 --steps 10 --batch 8 --out $O/${T}_ab_bs8.json 2>&1 | grep -v amdgpu.ids | tee $O/${T}_ab_bs8_g32_shortk.log
python benchmarks/ab.py --arms - I2I_G32_SHORTK=640 --repeats 6 --steps 20 --batch 1 --out $O/${T}_ab_bs1.json 2>&1 | grep -v amdgpu.ids | tee $O/${T}_ab_bs1_g32_shortk.log
I2I_G32_SHORTK=640 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32 --no-modes --no-latency --per-op $O/${T}_per_op_bs8_shortk640.txt > $O/${T}_bench_bs8_shortk640.json 2> $O/${T}_bench.err
grep -E "up_blocks.3.attentions.1.*(proj_in|proj_out|to_out|to_q)|up_blocks.2.attentions.1.*(proj_in|proj_out|to_out|to_q)" $O/${T}_per_op_bs8_shortk640.txt
