#!/bin/bash
# Round-6 call H: the narrow-output conv (VAE decoder conv_out as a read-bound kernel of its own), the small-tile rule restricted to short K.
O=gpurun_out; T=r6h; export TMPDIR=/tmp; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "narrow or small_tile or conv3x3 or halo" 2>&1 | tail -6 | tee $O/${T}_op_tests.log
python benchmarks/ab.py --arms I2I_CONV_NARROW=0 - --repeats 6 --steps 10 --batch 8 --out $O/${T}_ab_bs8.json 2>&1 | grep -v amdgpu.ids | tee $O/${T}_ab_bs8_conv_narrow_in_out.log
python benchmarks/ab.py --arms I2I_SMALL_TILE_ROWS=0 - I2I_SMALL_TILE_K=1280 --repeats 6 --steps 20 --batch 1 --out $O/${T}_ab_bs1.json 2>&1 | grep -v amdgpu.ids | tee $O/${T}_ab_bs1_small_tile_short_k.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32 --no-modes --no-latency --per-op $O/${T}_per_op_bs8.txt > $O/${T}_bench_bs8.json 2> $O/${T}_bench.err
grep -h "conv_in\|conv_out" $O/${T}_per_op_bs8.txt
python - <<'PY'
import json
r = json.load(open("gpurun_out/r6h_bench_bs8.json"))
print(r["value"], r["ms_per_step"], {k: v for k, v in r["calib"].items() if k in ("mfma_tflops", "hbm_tbytes_per_s", "graph_node_us", "value_normalised")})
for k, v in r["kernel_breakdown_ms"].items(): print(k, v)
PY
timeout 900 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu -k "cfg2 or full_sd_turbo or u8_pipeline or cfg5" 2>&1 | tail -5 | tee $O/${T}_e2e_subset.log
