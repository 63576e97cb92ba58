#!/bin/bash
# Round-5 call B: the wide-tile conv with every MFMA gap written out (staged GroupNorm+SiLU transform) against the round-4 library
# (libi2i_turbo_r4base.so), same box, interleaved; parity first.
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "w32 or conv3x3 or halo or subpix or second_contraction or gn_stats" > $O/r5b_tests_ops.log 2>&1
tail -3 $O/r5b_tests_ops.log
OLD=img2img-turbo_amd/csrc/libi2i_turbo_r4base.so; NEW=img2img-turbo_amd/csrc/libi2i_turbo.so
SH="vae 128->128@512 gn,vae 256->256@256 gn,vae 512->512@128 gn,vae 512->512@64 gn,vae 256->128@512 gn"
for rep in 1 2; do
  for lib in $OLD $NEW; do
    echo "== $lib rep $rep"
    python benchmarks/bench_ops.py --lib $lib --tiles 0 --iters 5 --only "$SH" --out $O/r5b_ops.json
    python benchmarks/bench_ops.py --lib $lib --tiles 0 --iters 5 --only "$SH" --res --out $O/r5b_ops.json | sed 's/$/  [+res]/'
  done
done > $O/r5b_w32_ab.log 2>&1
grep -v amdgpu.ids $O/r5b_w32_ab.log
python bench.py --no-cpu-baseline --no-f32 --per-op $O/r5b_per_op_bs8.txt > $O/r5b_bench_bs8.json 2> $O/r5b_bench_bs8.err
cut -c1-400 $O/r5b_bench_bs8.json
I2I_LIB=$OLD python bench.py --no-cpu-baseline --no-f32 --no-latency > $O/r5b_bench_bs8_r4lib.json 2>> $O/r5b_bench_bs8.err
cut -c1-400 $O/r5b_bench_bs8_r4lib.json
timeout 300 python -m pytest tests/test_e2e_gpu.py -m gpu -x -q -k "cfg2 or tiny_pix2pix" > $O/r5b_tests_e2e.log 2>&1
tail -3 $O/r5b_tests_e2e.log
