#!/bin/bash
# Round-6 call P: key-split d = 64 flash attention at batch 1 (I2I_ATT_KSPLIT64 = splits; 0 = off): op tests, A/B, per-op.
O=gpurun_out; T=r6p; export TMPDIR=/tmp; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_text_encoder.py -x -q -m gpu -k "attention or text or clip" 2>&1 | tail -4 | tee $O/${T}_op_tests.log
python benchmarks/ab.py --arms I2I_ATT_KSPLIT64=0 I2I_ATT_KSPLIT64=2 - I2I_ATT_KSPLIT64=8 --repeats 6 --steps 20 --batch 1 --out $O/${T}_ab_bs1.json 2>&1 | grep -v amdgpu.ids | tee $O/${T}_ab_bs1_attention_ksplit64.log
python benchmarks/ab.py --arms I2I_ATT_KSPLIT64=0 - --repeats 5 --steps 20 --batch 2 --out $O/${T}_ab_bs2.json 2>&1 | grep -v amdgpu.ids | tee $O/${T}_ab_bs2_attention_ksplit64.log
python bench.py --batch 1 --steps 50 --warmup 10 --no-cpu-baseline --no-f32 --no-modes --no-latency --no-calib --per-op $O/${T}_per_op_bs1.txt > $O/${T}_bench_bs1.json 2> $O/${T}_bench.err
grep sdpa $O/${T}_per_op_bs1.txt | sort -rn | head -8
timeout 600 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu -k "full_sd_turbo or cfg2 or odd_size" 2>&1 | tail -3 | tee $O/${T}_e2e_subset.log
