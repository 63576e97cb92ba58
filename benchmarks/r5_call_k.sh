#!/bin/bash
# Round-5 call K: sub-pixel upsampler form of the wide conv on its 16 x 32 x 128 tile with a FOUR-deep weight ring (-DW32_SPX_RING4=1,
# libi2i_turbo_spx4.so) against the two-deep ring (every wait is vmcnt(0) on the youngest batch) and against the 8 x 32 x 256 tile.
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
NEW=img2img-turbo_amd/csrc/libi2i_turbo_spx4.so; OLD=img2img-turbo_amd/csrc/libi2i_turbo.so
I2I_LIB=$NEW I2I_ALLOW_LIB_OVERRIDE=1 timeout 300 python - <<'PY' 2>&1 | tail -3
import os, sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from img2img_turbo_amd import _capi
import opcheck as oc
lib = _capi.Library(os.environ["I2I_LIB"])
for kw in (dict(n=2, cin=128, cout=256, h=20, w=40), dict(n=1, cin=64, cout=136, h=9, w=33, seed=4), dict(n=2, cin=64, cout=256, h=20, w=40, k2c=128, seed=7),
           dict(n=1, cin=192, cout=128, h=16, w=32, seed=11), dict(n=2, cin=256, cout=256, h=64, w=64, seed=12), dict(n=1, cin=512, cout=512, h=32, w=64, k2c=256, seed=13)):
    oc.check_conv(lib, "cuda", torch.bfloat16, ups=1, subpix=True, tile=42, **kw)
print("spx4 parity ok")
PY
for rep in 1 2; do
  for lib in $OLD $NEW; do
    echo "== $lib rep $rep"
    python benchmarks/bench_ops.py --lib $lib --tiles 41,42 --iters 7 --subpix --only "vae up" --out $O/r5k_ops.json | sed 's/$/  [subpix]/'
  done
done 2>&1 | grep -v amdgpu.ids | tee $O/r5k_subpix_ring_ab.log
