#!/bin/bash
# Round-6 baseline on the tree of the round start: the headline line, per-op tables at batch 8 and batch 1 (which launches the
# UNet-side work of the round has to remove), kernel-trace stats of the batch-1 forward.
O=gpurun_out; T=r6a; export TMPDIR=/tmp; mkdir -p $O
python bench.py --steps 20 --warmup 5 --per-op $O/${T}_per_op_bs8.txt > $O/${T}_bench_bs8.json 2> $O/${T}_bench.err
python bench.py --batch 1 --no-cpu-baseline --no-f32 --per-op $O/${T}_per_op_bs1.txt > $O/${T}_bench_bs1.json 2>> $O/${T}_bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_trace_bs1 -o trace -- python bench.py --batch 1 --steps 20 --warmup 3 --no-cpu-baseline --no-latency --no-f32 --no-power > /dev/null 2> $O/${T}_trace_bs1.err
cp $(find $O/${T}_trace_bs1 -name "*kernel_stats.csv" | head -1) $O/${T}_bench_bs1_kernel_stats.csv
rm -rf $O/${T}_trace_bs1
amd-smi static -g 0 --json > $O/${T}_amd_smi_static.json 2>&1
amd-smi metric -g 0 --json > $O/${T}_amd_smi_metric.json 2>&1
lscpu | head -20 > $O/${T}_lscpu.txt
python -c "import diffusers" > $O/${T}_diffusers_import.txt 2>&1
for f in $O/${T}_bench_bs8.json $O/${T}_bench_bs1.json; do python - "$f" <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))
print(sys.argv[1].split("/")[-1], r["value"], "img/s", r["ms_per_step"], "ms/step", "bs1 p50", r.get("latency_bs1_ms_p50"), "frac", (r.get("roofline") or {}).get("frac"))
PY
done
