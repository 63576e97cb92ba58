#!/bin/bash
# Round-6 call E: the narrow-input conv (VAE conv_in as a write-bound kernel of its own): op tests, same-box A/B against the LDS-DMA igemm route.
O=gpurun_out; T=r6e; export TMPDIR=/tmp; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "narrow_input or conv_gn_part or conv3x3" 2>&1 | tail -6 | tee $O/${T}_op_tests.log
python benchmarks/ab.py --arms I2I_CONV_NARROW=0 I2I_CONV_NARROW=1 --repeats 6 --steps 10 --batch 8 --out $O/${T}_ab_bs8.json 2>&1 | grep -v amdgpu.ids | tee $O/${T}_ab_bs8_conv_narrow.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32 --no-modes --no-latency --per-op $O/${T}_per_op_bs8.txt > $O/${T}_bench_bs8.json 2> $O/${T}_bench.err
grep -h "conv_in\|conv_out" $O/${T}_per_op_bs8.txt
python - <<'PY'
import json
r = json.load(open("gpurun_out/r6e_bench_bs8.json"))
print(r["value"], r["ms_per_step"], {k: v for k, v in r["calib"].items() if k in ("mfma_tflops", "hbm_tbytes_per_s", "graph_node_us", "value_normalised")})
PY
timeout 600 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu -k "cfg2 or full_sd_turbo or checkpoint_files or u8_pipeline" 2>&1 | tail -5 | tee $O/${T}_e2e_subset.log
