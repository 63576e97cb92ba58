#!/bin/bash
# Round-6 call G: batch-1 work -- the 64 x 32 tile of the LDS-DMA igemm for the un-sliced small linears, 64-query attention workgroups
# for the small grids: op tests, same-box A/Bs at batch 1 and 8, per-op profile at batch 1.
O=gpurun_out; T=r6g; export TMPDIR=/tmp; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attention or small_tile or dma_igemm" 2>&1 | tail -6 | tee $O/${T}_op_tests.log
python benchmarks/ab.py --arms I2I_SMALL_TILE_ROWS=0,I2I_ATT_QF=2 I2I_SMALL_TILE_ROWS=0 I2I_SMALL_TILE_ROWS=256 - I2I_SMALL_TILE_ROWS=4096 --repeats 6 --steps 20 --batch 1 --out $O/${T}_ab_bs1.json 2>&1 | grep -v amdgpu.ids | tee $O/${T}_ab_bs1_small_tile_att_qf.log
python benchmarks/ab.py --arms I2I_SMALL_TILE_ROWS=0 - --repeats 6 --steps 10 --batch 8 --out $O/${T}_ab_bs8.json 2>&1 | grep -v amdgpu.ids | tee $O/${T}_ab_bs8_small_tile.log
python bench.py --batch 1 --steps 50 --warmup 10 --no-cpu-baseline --no-f32 --no-modes --no-latency --per-op $O/${T}_per_op_bs1.txt > $O/${T}_bench_bs1.json 2> $O/${T}_bench.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/r6g_bench_bs1.json"))
print(r["value"], r["ms_per_step"], {k: v for k, v in r["calib"].items() if k in ("mfma_tflops", "hbm_tbytes_per_s", "graph_node_us", "value_normalised")})
for k, v in r["kernel_breakdown_ms"].items(): print(k, v["ms"], v["launches"])
PY
timeout 900 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu -k "full_sd_turbo or odd_size or cfg3" 2>&1 | tail -5 | tee $O/${T}_e2e_subset.log
