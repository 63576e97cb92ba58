"""GPU parity of the wide-tile conv (conv3x3_w32.hip, tile ids 41, 42) against F.conv2d: quick standalone runner used
next to the A/B benchmarks (the same checks live in tests/test_ops_gpu.py)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import opcheck as oc  # noqa: E402
from img2img_turbo_amd import _capi as K  # noqa: E402

lib = K.default_library()
tiles = [int(t) for t in (sys.argv[1].split(",") if len(sys.argv) > 1 else "41,42".split(","))]
gn_ok = {41, 42}
for t in tiles:
    for rep in range(3):      # repeated: races show up as run-to-run differences
        e = []
        if t in gn_ok:
            e.append(oc.check_conv(lib, "cuda", torch.bfloat16, n=2, cin=128, cout=256, h=40, w=72, gn=True, act=1, groups=8, res=True, tile=t, seed=rep))
            e.append(oc.check_conv(lib, "cuda", torch.float16, n=1, cin=64, cin2=128, cout=136, h=33, w=65, gn=True, act=1, groups=8, tile=t, seed=rep))
        e.append(oc.check_conv(lib, "cuda", torch.bfloat16, n=3, cin=256, cout=384, h=24, w=64, tile=t, seed=rep))
        e.append(oc.check_conv(lib, "cuda", torch.float16, n=9, cin=64, cout=128, h=64, w=96, res=True, tile=t, seed=rep))
        print("tile", t, "rep", rep, " ".join("%.2e" % x for x in e), flush=True)
print("W32 GPU PARITY OK")
