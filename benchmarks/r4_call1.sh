#!/bin/bash
# round-4 call 1: baseline line with the new fields, f32 line, PMC evidence for igemm_dma, A/B harness dry run
O=gpurun_out; export TMPDIR=/tmp; mkdir -p $O
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --per-op $O/r4a_per_op_bs8.txt > $O/r4a_bench_bs8.json 2> $O/r4a_bench_bs8.err
python bench.py --dtype f32 --batch 8 --steps 4 --warmup 1 --per-op $O/r4_per_op_f32_bs8.txt > $O/r4_bench_f32_bs8.json 2> $O/r4_bench_f32_bs8.err
# PMC: igemm_dma on its characteristic shapes (plain + GEGLU epilogue)
SH="unet lin 320->2560 T4096,unet lin 1280->320 T4096,vae sc 256->128@512 1x1,vae down 128@512 s2,unet lin 320->320 T4096,unet lin 640->5120 T1024,unet lin 1280->10240 T256"
timeout 500 bash benchmarks/pmc_conv.sh $O/r4_pmc_igemm "$SH" > /dev/null 2>&1
timeout 400 bash benchmarks/pmc_conv.sh $O/r4_pmc_geglu "unet lin 320->2560 T4096,unet lin 640->5120 T1024,unet lin 1280->10240 T256" "--geglu" > /dev/null 2>&1
for d in r4_pmc_igemm r4_pmc_geglu; do
  python tools/pmc_summary.py $(find $O/$d/sq1 -name "*counter_collection.csv" | head -1) $(find $O/$d/sq2 -name "*counter_collection.csv" | head -1) \
      $(find $O/$d/fetch -name "*counter_collection.csv" | head -1) $(find $O/$d/write -name "*counter_collection.csv" | head -1) igemm_dma > $O/${d}_summary.txt 2>&1
  cat $O/$d/sq1.log | grep -v "^$" | tail -12 >> $O/${d}_summary.txt
done
# A/B harness: the two round-3 single-shot A/Bs again, with error bars
timeout 400 python benchmarks/ab.py --arms "I2I_W32_XCDTN=1" "I2I_W32_XCDTN=0" --repeats 6 --steps 10 --out $O/r4_ab_xcdtn.json > $O/r4_ab_xcdtn.log 2>&1
timeout 400 python benchmarks/ab.py --arms "I2I_FUSE_SKIP=1" "I2I_FUSE_SKIP=0" --repeats 6 --steps 10 --out $O/r4_ab_fuse_skip.json > $O/r4_ab_fuse_skip.log 2>&1
python - <<'PY'
import json
for f in ["r4a_bench_bs8", "r4_bench_f32_bs8"]:
    try:
        r = json.load(open("gpurun_out/%s.json" % f))
        print(f, r["value"], "img/s", r["ms_per_step"], "ms", "frac", r["roofline"]["frac"], "lat1", r.get("latency_bs1_ms_p50"), "lat8", r.get("latency_bs8_ms_p50"), "lat32", r.get("latency_bs32_ms_p50"), "parity", r.get("parity_max_abs"))
        print({k: (v["ms"], v["launches"], v["tflops"]) for k, v in r["kernel_breakdown_ms"].items()})
    except Exception as e:
        print(f, "FAILED", e)
PY
cat $O/r4_pmc_igemm_summary.txt $O/r4_pmc_geglu_summary.txt | head -120
cat $O/r4_ab_xcdtn.log $O/r4_ab_fuse_skip.log | tail -8
tail -3 $O/*.err
