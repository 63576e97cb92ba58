#!/bin/bash
# Round-5 call J: the prologues of the two wide kernels (conv3x3_w32: no up_src() address arithmetic in the plain form, hidden halo loads
# with counted waits so that the transform takes chunks in arrival order, accumulator zeroing in the loads' shadow; gemm_w32: accumulator
# zeroing in the shadow of the first DMAs) against the library of the previous commit (libi2i_turbo_r5i.so), same box, interleaved.
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "w32 or conv3x3 or halo or subpix or second_contraction or gemm or geglu or gather or linear" > $O/r5j_tests_ops.log 2>&1
tail -3 $O/r5j_tests_ops.log
OLD=img2img-turbo_amd/csrc/libi2i_turbo_r5i.so; NEW=img2img-turbo_amd/csrc/libi2i_turbo.so
SH="vae 128->128@512 gn,vae 256->256@256 gn,vae 512->512@128 gn,vae 512->512@64 gn,vae 256->128@512 gn"
for rep in 1 2; do
  for lib in $OLD $NEW; do
    echo "== $lib rep $rep"
    python benchmarks/bench_ops.py --lib $lib --tiles 0 --iters 7 --only "$SH" --out $O/r5j_ops.json
    python benchmarks/bench_ops.py --lib $lib --tiles 0 --iters 7 --only "$SH" --res --out $O/r5j_ops.json | sed 's/$/  [+res]/'
    python benchmarks/bench_ops.py --lib $lib --tiles 0 --iters 7 --only "vae 128->128@512 gn,vae 512->512@128 gn" --fill zero --out $O/r5j_ops.json | sed 's/$/  [zero operands]/'
    python benchmarks/bench_ops.py --lib $lib --tiles 0 --iters 7 --subpix --only "vae up" --out $O/r5j_ops.json | sed 's/$/  [subpix]/'
  done
done 2>&1 | grep -v amdgpu.ids | tee $O/r5j_w32_prologue_ab.log
GS="unet lin 320->2560,unet lin 1280->320 T4096,unet lin 320->320 T4096,unet lin 640->5120,unet lin 2560->640,unet lin 1280->10240,unet lin 5120->1280,unet lin 640->640,unet lin 1280->1280 T256,vae down 128@512 s2,unet 320->320@64 gn,unet 960->320@64 gn"
for rep in 1 2; do
  for lib in $OLD $NEW; do
    echo "== $lib rep $rep"
    python benchmarks/bench_ops.py --lib $lib --nogn --tiles 50 --iters 7 --only "$GS" --out $O/r5j_ops.json
    python benchmarks/bench_ops.py --lib $lib --nogn --tiles 50 --iters 7 --only "unet lin 320->2560,unet lin 640->5120,unet lin 1280->10240" --geglu --out $O/r5j_ops.json
  done
done 2>&1 | grep -v "amdgpu.ids\|n/a" | tee $O/r5j_g32_prologue_ab.log
for rep in 1 2 3; do
I2I_LIB=$OLD python bench.py --no-cpu-baseline --no-f32 --no-power --no-latency > $O/r5j_bench_bs8_old_$rep.json 2>> $O/r5j_bench.err
python bench.py --no-cpu-baseline --no-f32 --no-power --no-latency > $O/r5j_bench_bs8_new_$rep.json 2>> $O/r5j_bench.err
done
I2I_LIB=$OLD python bench.py --batch 1 --no-cpu-baseline --no-f32 --no-power > $O/r5j_bench_bs1_old.json 2>> $O/r5j_bench.err
python bench.py --batch 1 --no-cpu-baseline --no-f32 --no-power > $O/r5j_bench_bs1_new.json 2>> $O/r5j_bench.err
python - <<'PY'
import json
for n in ("bs8_old_1","bs8_new_1","bs8_old_2","bs8_new_2","bs8_old_3","bs8_new_3","bs1_old","bs1_new"):
    try:
        r=json.load(open('gpurun_out/r5j_bench_%s.json'%n)); kb=r['kernel_breakdown_ms']
        print("%-10s %8.3f img/s  %7.3f ms/step  conv3x3_w32 %6.3f ms  subpix %6.3f  gemm_w32 %6.3f ms  roofline.frac %.4f  lat_bs1 %s" % (n, r['value'], r['ms_per_step'], kb['conv3x3_w32_kernel']['ms'], kb.get('conv3x3_w32_kernel<SUBPIX>',{}).get('ms',0), kb['gemm_w32_kernel']['ms'], r['roofline']['frac'], r.get('latency_bs1_ms_p50')))
    except Exception as e: print(n, 'FAILED', e)
PY
timeout 400 python -m pytest tests/test_e2e_gpu.py -m gpu -x -q -k "cfg2 or tiny_pix2pix or odd or cyclegan" > $O/r5j_tests_e2e.log 2>&1
tail -3 $O/r5j_tests_e2e.log
