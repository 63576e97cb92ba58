#!/bin/bash
# A/B of the hidden LDS-DMA experiment build (csrc/build.py --tag glds_asm --defs="-DI2I_GLDS_ASM=1 -DI2I_GEMM_GNPART=1 -DI2I_PST_CONV=1"; see DESIGN.md
# "first experiments of the next round"): op parity on the real DMA paths, per-op rates, then the bench line, each
# against the product library.  Run on the GPU box from the repo root; writes gpurun_out/exp_glds_asm.txt.
EXP=img2img-turbo_amd/csrc/libi2i_turbo_glds_asm.so
O=gpurun_out/exp_glds_asm.txt; mkdir -p gpurun_out; : > $O
[ -f $EXP ] || python img2img-turbo_amd/csrc/build.py --tag glds_asm --defs="-DI2I_GLDS_ASM=1 -DI2I_GEMM_GNPART=1 -DI2I_PST_CONV=1" >> $O 2>&1
echo "== parity (experiment library)" >> $O
I2I_LIB=$EXP timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q >> $O 2>&1
echo "== end-to-end parity, tiny architecture + SD-Turbo 512 (experiment library)" >> $O
I2I_LIB=$EXP timeout 600 python -m pytest tests/test_e2e_gpu.py -m gpu -x -q -s -k "tiny_pix2pix or full_sd_turbo" >> $O 2>&1
# exp = everything; exp_nopst = experiment library with the persistent streams off (I2I_PERSIST_WGS=0: one tile per workgroup in
# the halo conv AND the igemm), which isolates the hidden-DMA / counted-wait effect
for lib in product exp exp_nopst; do
  unset I2I_LIB I2I_PERSIST_WGS
  if [ $lib != product ]; then export I2I_LIB=$EXP; fi
  if [ $lib = exp_nopst ]; then export I2I_PERSIST_WGS=0; fi
  echo "== bench_ops ($lib)" >> $O
  timeout 200 python benchmarks/bench_ops.py --only "vae 128->128@512 gn,vae 256->256@256 gn,vae 512->512@128 gn,unet 320->320@64,lin 320->2560,lin 1280->320,skip" --out gpurun_out/exp_ops_$lib.json >> $O 2>&1
  echo "== bench_attention ($lib)" >> $O
  timeout 100 python benchmarks/bench_attention.py >> $O 2>&1
  echo "== bench ($lib)" >> $O
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline >> $O 2>&1
done
unset I2I_LIB I2I_PERSIST_WGS
grep -v amdgpu.ids $O | grep -v "^{" ; grep -o "\"value\": [0-9.]*" $O
