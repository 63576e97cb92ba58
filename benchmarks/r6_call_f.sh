#!/bin/bash
# Round-6 call F: the reworked d = 64 flash attention (lazy reference in the accumulator, row sums on the matrix pipe, permuted keys):
# op tests, same-box A/B against the previous kernel (library built with the old attention.hip), per-op profile, end-to-end gates.
O=gpurun_out; T=r6f; export TMPDIR=/tmp; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attention" 2>&1 | tail -6 | tee $O/${T}_op_tests.log
OLD=img2img-turbo_amd/csrc/libi2i_turbo_oldatt.so
python benchmarks/ab.py --arms I2I_LIB=$OLD - I2I_ATT_Q_LOG2=0 --repeats 6 --steps 10 --batch 8 --out $O/${T}_ab_bs8.json 2>&1 | grep -v amdgpu.ids | tee $O/${T}_ab_bs8_attention.log
python benchmarks/ab.py --arms I2I_LIB=$OLD - --repeats 6 --steps 20 --batch 1 --out $O/${T}_ab_bs1.json 2>&1 | grep -v amdgpu.ids | tee $O/${T}_ab_bs1_attention.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32 --no-latency --per-op $O/${T}_per_op_bs8.txt > $O/${T}_bench_bs8.json 2> $O/${T}_bench.err
grep -h "sdpa" $O/${T}_per_op_bs8.txt | head -8
python - <<'PY'
import json
r = json.load(open("gpurun_out/r6f_bench_bs8.json"))
print(r["value"], r["ms_per_step"], {k: v for k, v in r["calib"].items() if k in ("mfma_tflops", "hbm_tbytes_per_s", "graph_node_us", "value_normalised")})
print(r["kernel_breakdown_ms"]["attention_dma_kernel"])
print(r.get("precision_modes"))
PY
timeout 900 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu -k "cfg2 or full_sd_turbo or cfg3 or text" 2>&1 | tail -5 | tee $O/${T}_e2e_subset.log
