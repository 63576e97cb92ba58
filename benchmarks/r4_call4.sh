#!/bin/bash
# round-4 call 4: new parity tests (per-configuration floors stage by stage, SD-Turbo-size snapshot files), igemm_dma fast
# gather A/B, w32 stagger experiment
O=gpurun_out; export TMPDIR=/tmp; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "dma or large_offset or conv_variants or stride" > $O/r4d_gputests_ops.log 2>&1; tail -3 $O/r4d_gputests_ops.log
timeout 1500 python -m pytest tests/test_e2e_gpu.py -m gpu -x -q -s -k "stage_by_stage or snapshot_files" > $O/r4d_gputests_floor_files.log 2>&1; grep -E "floor\]|files\]|parity\]|passed|failed|Error" $O/r4d_gputests_floor_files.log | tail -30
SH="vae down 128@512 s2,unet 1280->1280@16 gn,unet 1280->1280@8 gn,unet 2560->1280@8 gn,unet 640->640@32 gn"
for L in "" "--lib img2img-turbo_amd/csrc/libi2i_turbo_nofg.so"; do
  echo "== bench_ops $L"; python benchmarks/bench_ops.py --nogn --tiles 20 --splitk 0,4 --only "$SH" $L --out $O/r4d_fg.json 2>&1 | grep -v amdgpu
done > $O/r4d_ab_fast_gather.log 2>&1; cat $O/r4d_ab_fast_gather.log
timeout 700 python benchmarks/ab.py --arms "I2I_W32_STAGGER=0" "I2I_W32_STAGGER=1" "I2I_W32_STAGGER=2" "I2I_W32_STAGGER=4" --repeats 5 --steps 10 --out $O/r4d_ab_stagger.json > $O/r4d_ab_stagger.log 2>&1; tail -5 $O/r4d_ab_stagger.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --per-op $O/r4d_per_op_bs8.txt > $O/r4d_bench_bs8.json 2> $O/r4d_bench_bs8.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/r4d_bench_bs8.json"))
print(r["value"], "img/s", r["ms_per_step"], "ms", "frac", r["roofline"]["frac"], "lat1", r.get("latency_bs1_ms_p50"))
print({k: (v["ms"], v["launches"], v["tflops"]) for k, v in r["kernel_breakdown_ms"].items()})
PY
tail -3 $O/r4d_bench_bs8.err
