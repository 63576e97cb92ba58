#!/bin/bash
# round-4 call 5: gemm_w32 two-workgroups-per-CU configs, GELU with FMAs, fast gather v2
O=gpurun_out; export TMPDIR=/tmp; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "gemm_w32 or geglu" > $O/r4e_gputests_gemm.log 2>&1; tail -3 $O/r4e_gputests_gemm.log
python benchmarks/bench_ops.py --tiles 20,51,52,55 --geglu --only "unet lin 320->2560 T4096,unet lin 640->5120 T1024,unet lin 1280->10240 T256" --out $O/r4e_geglu.json 2>&1 | grep -v "amdgpu\|n/a" > $O/r4e_bench_ops_geglu.log; cat $O/r4e_bench_ops_geglu.log
python benchmarks/bench_ops.py --tiles 51,52,54,55,56 --res --only "unet lin 320->320 T4096,unet lin 1280->320 T4096,unet lin 2560->640 T1024,unet lin 640->640 T1024,unet lin 5120->1280 T256,unet lin 1280->1280 T256" --out $O/r4e_res.json 2>&1 | grep -v "amdgpu\|n/a" > $O/r4e_bench_ops_res.log; cat $O/r4e_bench_ops_res.log
SH="vae down 128@512 s2,unet 1280->1280@16 gn,unet 1280->1280@8 gn,unet 640->640@32 gn"
for L in "" "--lib img2img-turbo_amd/csrc/libi2i_turbo_nofg.so"; do
  echo "== bench_ops $L"; python benchmarks/bench_ops.py --nogn --tiles 20 --splitk 0,4 --only "$SH" $L --out $O/r4e_fg.json 2>&1 | grep -v amdgpu
done > $O/r4e_ab_fast_gather.log 2>&1; cat $O/r4e_ab_fast_gather.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --per-op $O/r4e_per_op_bs8.txt > $O/r4e_bench_bs8.json 2> $O/r4e_bench_bs8.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/r4e_bench_bs8.json"))
print(r["value"], "img/s", r["ms_per_step"], "ms", "frac", r["roofline"]["frac"], "lat1", r.get("latency_bs1_ms_p50"))
print({k: (v["ms"], v["launches"], v["tflops"]) for k, v in r["kernel_breakdown_ms"].items()})
PY
