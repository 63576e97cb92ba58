#!/bin/bash
# round-4 call 3: shortcut fold + tuned GEMM routing on hardware
O=gpurun_out; export TMPDIR=/tmp; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "second_contraction or w32_conv_every or gemm_w32 or subpixel" > $O/r4c_gputests_ops.log 2>&1; tail -5 $O/r4c_gputests_ops.log
timeout 900 python -m pytest tests/test_e2e_gpu.py -m gpu -x -q -s -k "cfg2 or skip_convs_folded or tiny_pix2pix or cfg3" > $O/r4c_gputests_e2e.log 2>&1; grep -E "parity|passed|failed|Error" $O/r4c_gputests_e2e.log | tail -20
timeout 600 python benchmarks/ab.py --arms "I2I_FUSE_SHORTCUT=1" "I2I_FUSE_SHORTCUT=0" "I2I_GEMM_W32=0" --repeats 6 --steps 10 --out $O/r4c_ab_shortcut_gemm.json > $O/r4c_ab_shortcut_gemm.log 2>&1; tail -4 $O/r4c_ab_shortcut_gemm.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --per-op $O/r4c_per_op_bs8.txt > $O/r4c_bench_bs8.json 2> $O/r4c_bench_bs8.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/r4c_bench_bs8.json"))
print(r["value"], "img/s", r["ms_per_step"], "ms", "frac", r["roofline"]["frac"], "lat1", r.get("latency_bs1_ms_p50"))
print({k: (v["ms"], v["launches"], v["tflops"]) for k, v in r["kernel_breakdown_ms"].items()})
PY
grep -E "conv_shortcut|resnets.0.conv2" $O/r4c_per_op_bs8.txt | head -20
tail -3 $O/r4c_bench_bs8.err
