#!/bin/bash
# Round 6, after the last planner change (kernel sources unchanged: the PMC passes and the traffic record of r6_final.sh stay valid): the bench
# lines of every configuration and the rocprofv3 kernel statistics again, on one box.
TAG=${1:-r6}; O=gpurun_out; export TMPDIR=/tmp; mkdir -p $O
CMD="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-latency --no-f32 --no-modes --no-power"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace -o trace -- $CMD > $O/${TAG}_trace.json 2> $O/${TAG}_trace.err
cp $(find $O/${TAG}_trace -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench_bs8_kernel_stats.csv
t0=$SECONDS
python bench.py --per-op $O/${TAG}_per_op_bs8.txt > $O/${TAG}_bench_bs8.json 2> $O/${TAG}_bench_bs8.err
echo "default bench.py line (what the driver runs): $((SECONDS - t0)) s of wall clock"
python bench.py --activation-offset 30 --no-cpu-baseline --no-f32 --no-modes --no-latency --steps 20 > $O/${TAG}_bench_bs8_offset30.json 2>> $O/${TAG}_bench_bs8.err
python bench.py --activation-offset 30 --dtype f16 --steps 20 --no-f32 --no-latency > $O/${TAG}_bench_bs8_offset30_f16.json 2>> $O/${TAG}_bench_bs8.err
python bench.py --batch 1 --no-cpu-baseline --per-op $O/${TAG}_per_op_bs1.txt > $O/${TAG}_bench_bs1.json 2>> $O/${TAG}_bench_bs8.err
python bench.py --batch 32 --no-cpu-baseline --steps 10 > $O/${TAG}_bench_bs32.json 2>> $O/${TAG}_bench_bs8.err
python bench.py --model cyclegan --batch 4 --no-cpu-baseline > $O/${TAG}_bench_cfg3_cyclegan_bs4.json 2>> $O/${TAG}_bench_bs8.err
python bench.py --stochastic --gamma 0.4 --batch 16 --no-cpu-baseline > $O/${TAG}_bench_cfg4_stochastic_bs16.json 2>> $O/${TAG}_bench_bs8.err
python bench.py --size 1024 --dtype f16 --batch 8 --steps 10 --no-cpu-baseline > $O/${TAG}_bench_cfg5_1024_f16_bs8.json 2>> $O/${TAG}_bench_bs8.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace_bs1 -o trace -- python bench.py --batch 1 --steps 20 --warmup 3 --no-cpu-baseline --no-latency --no-f32 --no-modes > /dev/null 2> $O/${TAG}_trace_bs1.err
cp $(find $O/${TAG}_trace_bs1 -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench_bs1_kernel_stats.csv
for f in $O/${TAG}_bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1]))
    print(sys.argv[1].split("/")[-1], r["value"], "img/s", r["ms_per_step"], "ms/step", "frac", (r.get("roofline") or {}).get("frac"), "parity", r.get("parity_max_abs"), r.get("parity_psnr_db"), "wall", (r.get("phase_s") or {}).get("total"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
python __graft_entry__.py smoke 2>&1 | tail -3
find $O -mindepth 1 -maxdepth 1 -type d -name "${TAG}_*" -exec rm -rf {} +
find $O -type f -size +3M -delete
