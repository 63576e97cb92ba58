#!/bin/bash
# Re-collect the hash-keyed PMC traffic record of the conv3x3_* launches (two separate passes) and the headline lines on the current tree:
# what final_round.sh does in its steps 1-3, without the other configurations (used after the last small kernel changes of round 5).
TAG=${1:-r5t}; O=gpurun_out; export TMPDIR=/tmp; mkdir -p $O
CMD="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-latency --no-f32 --no-power"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_fetch -o fetch -- $CMD > /dev/null 2> $O/${TAG}_fetch.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_write -o write -- $CMD > /dev/null 2> $O/${TAG}_write.err
F=$(find $O/${TAG}_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/${TAG}_write -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py $F $W conv3x3_ --batch 8 --dtype bf16 --size 512 --source-hash $(python -c "import bench; print(bench.source_hash())") \
    --collected "$TAG: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of '$CMD'" > $O/${TAG}_traffic_conv3x3.json
cp $O/${TAG}_traffic_conv3x3.json profiles/traffic_conv3x3.json
cat $O/${TAG}_traffic_conv3x3.json
python bench.py --steps 20 --warmup 5 --per-op $O/${TAG}_per_op_bs8.txt > $O/${TAG}_bench_bs8.json 2> $O/${TAG}_bench.err
python bench.py --batch 1 --no-cpu-baseline > $O/${TAG}_bench_bs1.json 2>> $O/${TAG}_bench.err
python - "$O" "$TAG" <<'PY'
import json, sys
for n in ("bs8", "bs1"):
    r = json.load(open("%s/%s_bench_%s.json" % (sys.argv[1], sys.argv[2], n)))
    print(n, r["value"], "img/s", r["ms_per_step"], "ms", "frac", r["roofline"]["frac"], "at clock", r["roofline"].get("frac_at_measured_clock"), "traffic", r["roofline"]["traffic"],
          "p50 bs1", r.get("latency_bs1_ms_p50"), "parity", r.get("parity_max_abs"), r.get("parity_psnr_db"), r.get("power"))
PY
