#!/bin/bash
# round-4 call 10: K-sliced wide GEMM for the UNet's small planes, per group of ops (bit mask), interleaved A/B; then parity + bench
O=gpurun_out; export TMPDIR=/tmp; mkdir -p $O
python benchmarks/ab.py --arms "I2I_W32_SPLITK=7" "I2I_W32_SPLITK=3" "I2I_W32_SPLITK=2" "I2I_W32_SPLITK=0" --repeats 6 --steps 10 --out $O/r4j_ab_bs8.json > $O/r4j_ab_bs8.log 2>&1; grep -v amdgpu $O/r4j_ab_bs8.log | tail -5
timeout 600 python -m pytest tests/test_e2e_gpu.py -m gpu -x -q -k "test_cfg2" > $O/r4j_gputests_e2e.log 2>&1; tail -3 $O/r4j_gputests_e2e.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --per-op $O/r4j_per_op_bs8.txt > $O/r4j_bench_bs8.json 2> $O/r4j_bench_bs8.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/r4j_bench_bs8.json"))
print(r["value"], "img/s", r["ms_per_step"], "ms", "frac", r["roofline"]["frac"], "lat1", r.get("latency_bs1_ms_p50"))
print({k: (v["ms"], v["launches"], v["tflops"]) for k, v in r["kernel_breakdown_ms"].items()})
PY
