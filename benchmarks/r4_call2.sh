#!/bin/bash
# round-4 call 2: gemm_w32 on hardware -- parity, per-shape A/B against the LDS-DMA igemm, whole-step A/B with error bars
O=gpurun_out; export TMPDIR=/tmp; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "gemm_w32" > $O/r4b_gputests_gemm_w32.log 2>&1; tail -5 $O/r4b_gputests_gemm_w32.log
SH="unet lin 320->2560 T4096,unet lin 1280->320 T4096,unet lin 320->320 T4096,unet lin 640->5120 T1024,unet lin 2560->640 T1024,unet lin 640->640 T1024,unet lin 1280->10240 T256,unet lin 5120->1280 T256,unet lin 1280->1280 T256,vae lin 512->1024 T4096,vae lin 512->512 T4096,vae sc 256->128@512 1x1,vae sc 512->256@256 1x1,vae skip 128->256@512 1x1"
python benchmarks/bench_ops.py --tiles 20,51,52,53,54 --only "$SH" --out $O/r4b_bench_ops_gemm.json > $O/r4b_bench_ops_gemm.log 2>&1
python benchmarks/bench_ops.py --tiles 20,51,52 --geglu --only "unet lin 320->2560 T4096,unet lin 640->5120 T1024,unet lin 1280->10240 T256" --out $O/r4b_bench_ops_geglu.json > $O/r4b_bench_ops_geglu.log 2>&1
python benchmarks/bench_ops.py --tiles 20,51,52,53 --res --only "unet lin 1280->320 T4096,unet lin 320->320 T4096,vae lin 512->512 T4096" --out $O/r4b_bench_ops_res.json > $O/r4b_bench_ops_res.log 2>&1
grep -v "n/a" $O/r4b_bench_ops_gemm.log; cat $O/r4b_bench_ops_geglu.log $O/r4b_bench_ops_res.log | grep -v "n/a"
timeout 500 python benchmarks/ab.py --arms "I2I_GEMM_W32=1" "I2I_GEMM_W32=0" --repeats 6 --steps 10 --out $O/r4b_ab_gemm_w32.json > $O/r4b_ab_gemm_w32.log 2>&1; tail -3 $O/r4b_ab_gemm_w32.log
timeout 500 python benchmarks/ab.py --arms "I2I_GEMM_W32=1" "I2I_GEMM_W32=0" --repeats 6 --steps 20 --batch 1 --out $O/r4b_ab_gemm_w32_bs1.json > $O/r4b_ab_gemm_w32_bs1.log 2>&1; tail -3 $O/r4b_ab_gemm_w32_bs1.log
python bench.py --steps 20 --warmup 5 --per-op $O/r4b_per_op_bs8.txt > $O/r4b_bench_bs8.json 2> $O/r4b_bench_bs8.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/r4b_bench_bs8.json"))
print(r["value"], "img/s", r["ms_per_step"], "ms", "frac", r["roofline"]["frac"], "lat1", r.get("latency_bs1_ms_p50"), "parity", r.get("parity_max_abs"), r.get("parity_psnr_db"))
print({k: (v["ms"], v["launches"], v["tflops"]) for k, v in r["kernel_breakdown_ms"].items()})
PY
grep "gemm_w32" $O/r4b_per_op_bs8.txt | sort -k1 -n -r | head -60
