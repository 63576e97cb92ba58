#!/bin/bash
# Round-6 call I: tile 27 (64 x 32 behind an 8-stage ring) for the un-sliced 1280-wide projections at batch 1; V^T fragment reads ahead of the
# exponentials in the d = 64 flash kernel (build variants).
O=gpurun_out; T=r6i; export TMPDIR=/tmp; mkdir -p $O
C=img2img-turbo_amd/csrc
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "small_tile or dma_igemm or narrow" 2>&1 | tail -6 | tee $O/${T}_op_tests.log
python benchmarks/ab.py --arms I2I_SMALL_TILE_K_DEEP=0 - I2I_SMALL_TILE_K_DEEP=1280 I2I_SMALL_TILE_K_DEEP=5120 --repeats 6 --steps 20 --batch 1 --out $O/${T}_ab_bs1.json 2>&1 | grep -v amdgpu.ids | tee $O/${T}_ab_bs1_deep_ring.log
python benchmarks/ab.py --arms - I2I_LIB=$C/libi2i_turbo_attv4.so I2I_LIB=$C/libi2i_turbo_attv8.so --repeats 6 --steps 10 --batch 8 --out $O/${T}_ab_bs8.json 2>&1 | grep -v amdgpu.ids | tee $O/${T}_ab_bs8_attention_vpre.log
python bench.py --batch 1 --steps 50 --warmup 10 --no-cpu-baseline --no-f32 --no-modes --no-latency --per-op $O/${T}_per_op_bs1.txt > $O/${T}_bench_bs1.json 2> $O/${T}_bench.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/r6i_bench_bs1.json"))
print(r["value"], r["ms_per_step"], {k: v for k, v in r["calib"].items() if k in ("mfma_tflops", "hbm_tbytes_per_s", "graph_node_us", "value_normalised")})
for k, v in r["kernel_breakdown_ms"].items(): print(k, v["ms"], v["launches"])
PY
grep -E "up_blocks.1.attentions.1.*(proj_in|proj_out|to_out|ff.net.2)|up_blocks.1.resnets.1.conv_shortcut" $O/${T}_per_op_bs1.txt
