#!/bin/bash
# (every rocprofv3 pass runs under `timeout`: a PMC pass once hung a box for 30 minutes)
# Round evidence set, from the repo root on the GPU box:  benchmarks/final_round.sh r2   (writes gpurun_out/<tag>_*)
#   1. the bench line of the headline configuration (cfg 2) with the CPU oracle beside it and the parity of image 0
#   2. rocprofv3 --kernel-trace --stats of the same command (per-kernel calls / total / average)
#   3. HBM traffic of the halo conv launches: two separate PMC passes (FETCH_SIZE, WRITE_SIZE), corrected per
#      MI355X_MICROARCH.md, written to profiles/traffic_conv3x3.json keyed on the kernel-source hash
#   4. one line each for the other BASELINE configurations at N = 1 (bs=1, bs=32, cfg 3, cfg 4, cfg 5)
TAG=${1:-r5}; O=gpurun_out; export TMPDIR=/tmp; mkdir -p $O
CMD="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-latency --no-f32 --no-power"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace -o trace -- $CMD > $O/${TAG}_trace.json 2> $O/${TAG}_trace.err
cp $(find $O/${TAG}_trace -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench_bs8_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_fetch -o fetch -- $CMD > /dev/null 2> $O/${TAG}_fetch.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_write -o write -- $CMD > /dev/null 2> $O/${TAG}_write.err
F=$(find $O/${TAG}_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/${TAG}_write -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py $F $W conv3x3_ --batch 8 --dtype bf16 --size 512 --source-hash $(python -c "import bench; print(bench.source_hash())") \
    --collected "$TAG: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of '$CMD'" > $O/${TAG}_traffic_conv3x3.json
cp $O/${TAG}_traffic_conv3x3.json profiles/traffic_conv3x3.json
t0=$SECONDS
python bench.py --per-op $O/${TAG}_per_op_bs8.txt > $O/${TAG}_bench_bs8.json 2> $O/${TAG}_bench_bs8.err
echo "default bench.py line (what the driver runs): $((SECONDS - t0)) s of wall clock"
# the same step with every VAE resnet's conv2 bias + 30 (residual streams on DC offsets: GroupNorm groups take the second pass)
python bench.py --activation-offset 30 --no-cpu-baseline --no-f32 --no-latency --steps 20 > $O/${TAG}_bench_bs8_offset30.json 2>> $O/${TAG}_bench_bs8.err
python bench.py --activation-offset 30 --dtype f16 --steps 20 --no-f32 --no-latency > $O/${TAG}_bench_bs8_offset30_f16.json 2>> $O/${TAG}_bench_bs8.err
python bench.py --batch 1 --no-cpu-baseline > $O/${TAG}_bench_bs1.json 2>> $O/${TAG}_bench_bs8.err
python bench.py --batch 32 --no-cpu-baseline --steps 10 > $O/${TAG}_bench_bs32.json 2>> $O/${TAG}_bench_bs8.err
python bench.py --model cyclegan --batch 4 --no-cpu-baseline > $O/${TAG}_bench_cfg3_cyclegan_bs4.json 2>> $O/${TAG}_bench_bs8.err
python bench.py --stochastic --gamma 0.4 --batch 16 --no-cpu-baseline > $O/${TAG}_bench_cfg4_stochastic_bs16.json 2>> $O/${TAG}_bench_bs8.err
python bench.py --size 1024 --dtype f16 --batch 8 --steps 10 --no-cpu-baseline > $O/${TAG}_bench_cfg5_1024_f16_bs8.json 2>> $O/${TAG}_bench_bs8.err
for f in $O/${TAG}_bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1]))
    print(sys.argv[1].split("/")[-1], r["value"], "img/s", r["ms_per_step"], "ms/step", "frac", (r.get("roofline") or {}).get("frac"), "parity", r.get("parity_max_abs"), r.get("parity_psnr_db"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
head -8 $O/${TAG}_bench_bs8_kernel_stats.csv | cut -c1-180
cat $O/${TAG}_traffic_conv3x3.json
# 5. bs=1 latency path: kernel-trace stats of the bs=1 bench (560 launches per forward)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace_bs1 -o trace -- python bench.py --batch 1 --steps 20 --warmup 3 --no-cpu-baseline --no-latency > /dev/null 2> $O/${TAG}_trace_bs1.err
cp $(find $O/${TAG}_trace_bs1 -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench_bs1_kernel_stats.csv
# 6. SQ counters of the halo conv on its two characteristic shapes (one counter set per pass)
timeout 600 bash benchmarks/pmc_conv.sh $O/${TAG}_pmc "vae 128->128@512 gn,vae 512->512@128 gn" > /dev/null 2>&1
python tools/pmc_summary.py $(find $O/${TAG}_pmc/sq1 -name "*counter_collection.csv" | head -1) $(find $O/${TAG}_pmc/sq2 -name "*counter_collection.csv" | head -1) conv3x3_ > $O/${TAG}_pmc_conv3x3_summary.txt 2>&1
cat $O/${TAG}_pmc_conv3x3_summary.txt
# 7. SQ + HBM counters of the wide GEMM (gemm_w32.hip) on its characteristic shapes: the GEGLU / ff.net.2 projections (tile 0 = auto) and the UNet's
#    3x3 convolutions through its im2col gather (256 x 160 tiles)
#    (one shape per pass set, so that the fetch of a projection and of a gathered 3x3 conv are attributable)
: > $O/${TAG}_pmc_gemm_w32_summary.txt
for shape in "unet lin 1280->320 T4096" "unet 960->320@64 gn"; do
  d=$O/${TAG}_pmc_g32_$(echo "$shape" | tr -c 'a-zA-Z0-9' '_')
  timeout 300 bash benchmarks/pmc_conv.sh $d "$shape" "--nogn --tiles 51" > /dev/null 2>&1
  echo "== $shape" >> $O/${TAG}_pmc_gemm_w32_summary.txt
  python tools/pmc_summary.py $(find $d -name "*counter_collection.csv" | sort | tr '\n' ' ') gemm_w32_kernel >> $O/${TAG}_pmc_gemm_w32_summary.txt 2>&1
done
cat $O/${TAG}_pmc_gemm_w32_summary.txt
# 8. socket power / shader clock while the dominant launches run back to back, random vs zero operands (benchmarks/power_probe.py)
timeout 200 python benchmarks/power_probe.py --out $O/${TAG}_power_probe.json 2>&1 | grep -v "amdgpu.ids\|smi sample" | cut -c1-220 > $O/${TAG}_power_probe.log
cat $O/${TAG}_power_probe.log
