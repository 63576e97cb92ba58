#!/bin/bash
# round-4 call 12: every UNet 3x3 conv on the wide GEMM (group 8 of I2I_W32_SPLITK) vs the halo conv: interleaved A/B at bs=8 and bs=1, parity, bench
O=gpurun_out; export TMPDIR=/tmp; mkdir -p $O
python benchmarks/ab.py --arms "I2I_W32_SPLITK=15" "I2I_W32_SPLITK=7" --repeats 6 --steps 10 --out $O/r4l_ab_bs8.json > $O/r4l_ab_bs8.log 2>&1; grep -v amdgpu $O/r4l_ab_bs8.log | tail -3
python benchmarks/ab.py --arms "I2I_W32_SPLITK=15" "I2I_W32_SPLITK=7" "I2I_W32_SPLITK=0" --batch 1 --repeats 6 --steps 20 --out $O/r4l_ab_bs1.json > $O/r4l_ab_bs1.log 2>&1; grep -v amdgpu $O/r4l_ab_bs1.log | tail -4
timeout 900 python -m pytest tests/test_e2e_gpu.py -m gpu -x -q -k "test_cfg2 or test_full_sd_turbo_512 or test_cfg5" > $O/r4l_gputests_e2e.log 2>&1; tail -3 $O/r4l_gputests_e2e.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --per-op $O/r4l_per_op_bs8.txt > $O/r4l_bench_bs8.json 2> $O/r4l_bench_bs8.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/r4l_bench_bs8.json"))
print(r["value"], "img/s", r["ms_per_step"], "ms", "frac", r["roofline"]["frac"], "lat1", r.get("latency_bs1_ms_p50"))
print({k: (v["ms"], v["launches"], v["tflops"]) for k, v in r["kernel_breakdown_ms"].items()})
PY
