#!/bin/bash
# round-4 call 6 (session 2): where the step stands after the gemm_w32 / shortcut-fold work: per-op table, bs=1 line
O=gpurun_out; export TMPDIR=/tmp; mkdir -p $O
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --per-op $O/r4f_per_op_bs8.txt > $O/r4f_bench_bs8.json 2> $O/r4f_bench_bs8.err
python bench.py --batch 1 --no-cpu-baseline --per-op $O/r4f_per_op_bs1.txt > $O/r4f_bench_bs1.json 2>> $O/r4f_bench_bs8.err
python - <<'PY'
import json
for f in ("gpurun_out/r4f_bench_bs8.json", "gpurun_out/r4f_bench_bs1.json"):
    r = json.load(open(f))
    print(r["value"], "img/s", r["ms_per_step"], "ms", "frac", r["roofline"]["frac"], "lat1", r.get("latency_bs1_ms_p50"))
    print({k: (v["ms"], v["launches"], v["tflops"]) for k, v in r["kernel_breakdown_ms"].items()})
PY
