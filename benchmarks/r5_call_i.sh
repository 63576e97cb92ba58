#!/bin/bash
# Round-5 call I: the epilogues of the two wide kernels without their serial round trips (conv3x3_w32: bias / residual quads read one
# fragment ahead, accumulators read out of the AGPRs at their use, row chunks read in one batch, predicate-free instantiation for tiles
# inside the plane; gemm_w32: bias + residual of block t+1 requested before block t's stores, predicate-free instantiation) against
# the same sources built with -DW32_EPI_BATCH=0 -DW32_EPI_FULL=0 -DG32_EPI_PIPE=0 (libi2i_turbo_epi0.so), same box, interleaved.
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "w32 or conv3x3 or halo or subpix or second_contraction or gemm or geglu or gather or linear" > $O/r5i_tests_ops.log 2>&1
tail -3 $O/r5i_tests_ops.log
OLD=img2img-turbo_amd/csrc/libi2i_turbo_epi0.so; NEW=img2img-turbo_amd/csrc/libi2i_turbo.so
SH="vae 128->128@512 gn,vae 256->256@256 gn,vae 512->512@128 gn,vae 512->512@64 gn,vae 256->128@512 gn"
for rep in 1 2; do
  for lib in $OLD $NEW; do
    echo "== $lib rep $rep"
    python benchmarks/bench_ops.py --lib $lib --tiles 0 --iters 7 --only "$SH" --out $O/r5i_ops.json
    python benchmarks/bench_ops.py --lib $lib --tiles 0 --iters 7 --only "$SH" --res --out $O/r5i_ops.json | sed 's/$/  [+res]/'
    python benchmarks/bench_ops.py --lib $lib --tiles 0 --iters 7 --only "$SH" --fill zero --out $O/r5i_ops.json | sed 's/$/  [zero operands]/'
  done
done 2>&1 | grep -v amdgpu.ids | tee $O/r5i_w32_epilogue_ab.log
GS="unet lin 320->2560,unet lin 1280->320 T4096,unet lin 320->320 T4096,unet lin 640->5120,unet lin 2560->640,unet lin 1280->10240,unet lin 5120->1280,unet lin 640->640,unet lin 1280->1280 T256,vae down 128@512 s2,vae down 512@128 s2,unet 320->320@64 gn,unet 960->320@64 gn,unet 640->640@32 gn"
for rep in 1 2; do
  for lib in $OLD $NEW; do
    echo "== $lib rep $rep"
    python benchmarks/bench_ops.py --lib $lib --nogn --tiles 50 --iters 7 --only "$GS" --out $O/r5i_ops.json
    python benchmarks/bench_ops.py --lib $lib --nogn --tiles 50 --iters 7 --only "unet lin" --res --out $O/r5i_ops.json | sed 's/$/  [+res]/'
    python benchmarks/bench_ops.py --lib $lib --nogn --tiles 50 --iters 7 --only "unet lin 320->2560,unet lin 640->5120,unet lin 1280->10240" --geglu --out $O/r5i_ops.json
  done
done 2>&1 | grep -v "amdgpu.ids\|n/a" | tee $O/r5i_g32_epilogue_ab.log
for rep in 1 2; do
I2I_LIB=$OLD python bench.py --no-cpu-baseline --no-f32 --no-power --no-latency > $O/r5i_bench_bs8_old_$rep.json 2>> $O/r5i_bench.err
python bench.py --no-cpu-baseline --no-f32 --no-power --no-latency > $O/r5i_bench_bs8_new_$rep.json 2>> $O/r5i_bench.err
done
I2I_LIB=$OLD python bench.py --batch 1 --no-cpu-baseline --no-f32 --no-power > $O/r5i_bench_bs1_old.json 2>> $O/r5i_bench.err
python bench.py --batch 1 --no-cpu-baseline --no-f32 --no-power > $O/r5i_bench_bs1_new.json 2>> $O/r5i_bench.err
python bench.py --no-cpu-baseline --no-f32 --per-op $O/r5i_per_op_bs8.txt > $O/r5i_bench_bs8.json 2>> $O/r5i_bench.err
python - <<'PY'
import json
for n in ("bs8_old_1","bs8_new_1","bs8_old_2","bs8_new_2","bs1_old","bs1_new","bs8"):
    try:
        r=json.load(open('gpurun_out/r5i_bench_%s.json'%n)); kb=r['kernel_breakdown_ms']
        print(n, r['value'], r['ms_per_step'], 'lat1', r.get('latency_bs1_ms_p50'), 'w32', kb.get('conv3x3_w32_kernel',{}).get('ms'), 'g32', kb.get('gemm_w32_kernel',{}).get('ms'), 'frac', r['roofline'].get('frac'), r.get('power'))
    except Exception as e: print(n, 'FAILED', e)
PY
timeout 400 python -m pytest tests/test_e2e_gpu.py -m gpu -x -q -k "cfg2 or tiny_pix2pix or odd or cyclegan" > $O/r5i_tests_e2e.log 2>&1
tail -3 $O/r5i_tests_e2e.log
