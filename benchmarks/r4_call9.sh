#!/bin/bash
# round-4 call 9: UNet small-plane 3x3 convolutions on the split-K wide GEMM in the step: parity at the BASELINE configs + interleaved A/B
O=gpurun_out; export TMPDIR=/tmp; mkdir -p $O
timeout 1200 python -m pytest tests/test_e2e_gpu.py -m gpu -x -q -k "test_cfg2 or test_cfg3 or test_cfg4" > $O/r4i_gputests_e2e.log 2>&1; tail -4 $O/r4i_gputests_e2e.log
python benchmarks/ab.py --arms "I2I_W32_SPLITK=2" "I2I_W32_SPLITK=1" "I2I_W32_SPLITK=0" --repeats 6 --steps 10 --out $O/r4i_ab_bs8.json > $O/r4i_ab_bs8.log 2>&1; grep -v amdgpu $O/r4i_ab_bs8.log | tail -4
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --per-op $O/r4i_per_op_bs8.txt > $O/r4i_bench_bs8.json 2> $O/r4i_bench_bs8.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/r4i_bench_bs8.json"))
print(r["value"], "img/s", r["ms_per_step"], "ms", "frac", r["roofline"]["frac"], "lat1", r.get("latency_bs1_ms_p50"))
print({k: (v["ms"], v["launches"], v["tflops"]) for k, v in r["kernel_breakdown_ms"].items()})
PY
