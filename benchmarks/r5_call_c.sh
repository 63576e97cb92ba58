#!/bin/bash
# Round-5 call C: tile stream (persistent workgroups) of the wide-tile conv: parity, then same-box A/B of
#   r4 library | new library, one workgroup per tile (I2I_W32_STREAM=0) | new library, tile stream
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "w32 or conv3x3 or halo or subpix or second_contraction or gn_stats" > $O/r5c_tests_ops.log 2>&1
tail -3 $O/r5c_tests_ops.log
OLD=img2img-turbo_amd/csrc/libi2i_turbo_r4base.so; NEW=img2img-turbo_amd/csrc/libi2i_turbo.so
SH="vae 128->128@512 gn,vae 256->256@256 gn,vae 512->512@128 gn,vae 512->512@64 gn,vae 256->128@512 gn"
for rep in 1 2; do
  for arm in "$OLD 1" "$NEW 0" "$NEW 1"; do
    set -- $arm
    echo "== $1 stream=$2 rep $rep"
    I2I_W32_STREAM=$2 python benchmarks/bench_ops.py --lib $1 --tiles 0 --iters 5 --only "$SH" --out $O/r5c_ops.json
    I2I_W32_STREAM=$2 python benchmarks/bench_ops.py --lib $1 --tiles 0 --iters 5 --only "$SH" --res --out $O/r5c_ops.json | sed 's/$/  [+res]/'
  done
done > $O/r5c_w32_ab.log 2>&1
grep -v amdgpu.ids $O/r5c_w32_ab.log
python benchmarks/bench_ops.py --lib $NEW --subpix --tiles 0 --iters 5 --only "vae up" --out $O/r5c_ops.json 2>&1 | grep -v amdgpu
I2I_W32_STREAM=0 python benchmarks/bench_ops.py --lib $NEW --subpix --tiles 0 --iters 5 --only "vae up" --out $O/r5c_ops.json 2>&1 | grep -v amdgpu
python benchmarks/ab.py --arms "I2I_W32_STREAM=1" "I2I_W32_STREAM=0" --repeats 5 --steps 8 --out $O/r5c_ab_stream.json > $O/r5c_ab_stream.log 2>&1
tail -8 $O/r5c_ab_stream.log
timeout 300 python -m pytest tests/test_e2e_gpu.py -m gpu -x -q -k "cfg2 or tiny_pix2pix or odd" > $O/r5c_tests_e2e.log 2>&1
tail -3 $O/r5c_tests_e2e.log
