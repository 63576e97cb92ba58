#!/bin/bash
# Round-5 call D: sliced single-launch GroupNorm statistics (ABI v7 counters): parity, then interleaved A/B at batch 8 and 1;
# the wide conv on zero / constant operands (how much of its time is clock rather than cycles).
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q > $O/r5d_tests_ops.log 2>&1
tail -3 $O/r5d_tests_ops.log
python benchmarks/ab.py --arms "I2I_GN_SLICED=1" "I2I_GN_SLICED=0" --repeats 5 --steps 8 --out $O/r5d_ab_gn_sliced_bs8.json > $O/r5d_ab_gn_sliced_bs8.log 2>&1
tail -4 $O/r5d_ab_gn_sliced_bs8.log
python benchmarks/ab.py --arms "I2I_GN_SLICED=1" "I2I_GN_SLICED=0" --batch 1 --repeats 5 --steps 20 --out $O/r5d_ab_gn_sliced_bs1.json > $O/r5d_ab_gn_sliced_bs1.log 2>&1
tail -4 $O/r5d_ab_gn_sliced_bs1.log
for fill in randn const zero; do
  echo "== fill $fill"
  python benchmarks/bench_ops.py --tiles 0 --iters 5 --fill $fill --only "vae 128->128@512 gn,vae 512->512@128 gn,unet lin 1280->10240 T256" --out $O/r5d_ops.json
done 2>&1 | grep -v amdgpu.ids | tee $O/r5d_fill.log
python bench.py --no-cpu-baseline --per-op $O/r5d_per_op_bs8.txt > $O/r5d_bench_bs8.json 2> $O/r5d_bench_bs8.err
cut -c1-300 $O/r5d_bench_bs8.json; python - <<'PY'
import json
r=json.load(open('gpurun_out/r5d_bench_bs8.json'))
print({k:r.get(k) for k in ('value','ms_per_step','latency_bs1_ms_p50','images_per_s_f32','ms_per_step_f32')})
for k,v in list(r['kernel_breakdown_ms'].items())[:14]: print(k,v)
PY
timeout 300 python -m pytest tests/test_e2e_gpu.py -m gpu -x -q -k "cfg2 or tiny_pix2pix or odd or fuzz" > $O/r5d_tests_e2e.log 2>&1
tail -3 $O/r5d_tests_e2e.log
