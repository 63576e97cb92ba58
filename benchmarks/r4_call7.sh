#!/bin/bash
# round-4 call 7: wide GEMM with the 3x3 stride-2 gather + GroupNorm partials, one-launch self-attention V^T, key-split VAE attention
O=gpurun_out; export TMPDIR=/tmp; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "gemm_w32 or attention_wide_head or w32" > $O/r4g_gputests_ops.log 2>&1; tail -3 $O/r4g_gputests_ops.log
timeout 900 python -m pytest tests/test_e2e_gpu.py -m gpu -x -q -k "test_full_sd_turbo_512 or test_cfg2" > $O/r4g_gputests_e2e.log 2>&1; tail -3 $O/r4g_gputests_e2e.log
python benchmarks/bench_ops.py --nogn --tiles 20,53,54 --only "vae down" --out $O/r4g_down.json 2>&1 | grep -v "amdgpu\|n/a" > $O/r4g_bench_ops_downsamplers.log; cat $O/r4g_bench_ops_downsamplers.log
python benchmarks/bench_ops.py --nogn --tiles 20,53,54 --only "vae lin" --out $O/r4g_vaelin.json 2>&1 | grep -v "amdgpu\|n/a" > $O/r4g_bench_ops_vae_lin.log; cat $O/r4g_bench_ops_vae_lin.log
for L in "" "--lib img2img-turbo_amd/csrc/libi2i_turbo_nopre.so" "" "--lib img2img-turbo_amd/csrc/libi2i_turbo_nopre.so"; do
  echo "== residual rows fetched by the last slab (default) vs by the epilogue (nopre): bench_ops --res $L"; python benchmarks/bench_ops.py --res --tiles 0 --iters 9 --only "vae 128->128@512 gn,vae 256->256@256 gn,vae 512->512@128 gn" $L --out $O/r4g_rd.json 2>&1 | grep -v "amdgpu\|n/a"
done > $O/r4g_ab_w32_residual_prefetch.log 2>&1; cat $O/r4g_ab_w32_residual_prefetch.log
python benchmarks/ab.py --arms "-" "I2I_GEMM_W32_CONV=0" "I2I_VT_ONE_LAUNCH=0" --repeats 6 --steps 10 --out $O/r4g_ab_bs8.json > $O/r4g_ab_bs8.log 2>&1; grep -v amdgpu $O/r4g_ab_bs8.log | tail -4
python benchmarks/ab.py --arms "-" "I2I_ATT_KSPLIT=0" "I2I_VT_ONE_LAUNCH=0" --batch 1 --repeats 6 --steps 20 --out $O/r4g_ab_bs1.json > $O/r4g_ab_bs1.log 2>&1; grep -v amdgpu $O/r4g_ab_bs1.log | tail -4
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --per-op $O/r4g_per_op_bs8.txt > $O/r4g_bench_bs8.json 2> $O/r4g_bench_bs8.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/r4g_bench_bs8.json"))
print(r["value"], "img/s", r["ms_per_step"], "ms", "frac", r["roofline"]["frac"], "lat1", r.get("latency_bs1_ms_p50"))
print({k: (v["ms"], v["launches"], v["tflops"]) for k, v in r["kernel_breakdown_ms"].items()})
PY
