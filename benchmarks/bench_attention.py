"""Time the fused attention kernels alone (hipGraph-free, HIP events inside the library): UNet / VAE shapes at bs 8."""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from img2img_turbo_amd import _capi as K   # noqa: E402
from img2img_turbo_amd import ops as O     # noqa: E402

SHAPES = [  # name, heads, d, tq, tk
    ("vae mid 1x512 T4096", 1, 512, 4096, 4096),
    ("unet self 5x64 T4096", 5, 64, 4096, 4096),
    ("unet self 10x64 T1024", 10, 64, 1024, 1024),
    ("unet self 20x64 T256", 20, 64, 256, 256),
    ("unet cross 5x64 T4096x77", 5, 64, 4096, 77),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--causal", type=int, default=0)
    a = ap.parse_args()
    lib = K.default_library()
    dt = torch.bfloat16
    for name, H, d, tq, tk in SHAPES:
        B, C = a.batch, H * d
        q = torch.randn(B, tq, C, device="cuda").to(dt)
        k = torch.randn(B, tk, C, device="cuda").to(dt)
        ldvt = (tk + 7) // 8 * 8
        vt = torch.randn(B, C, ldvt, device="cuda").to(dt)
        o = torch.empty(B, tq, C, device="cuda", dtype=dt)
        prog = K.Program()
        for _ in range(a.iters + 1):
            op = O.attention(q, k, vt, o, batch=B, heads=H, d=d, tq=tq, tk=tk, ldq=C, ldk=C, ldvt=ldvt, ldo=C,
                             q_bs=tq * C, k_bs=tk * C, vt_bs=C * ldvt, o_bs=tq * C, scale=1.0 / math.sqrt(d), causal=a.causal)
            prog.add(op[0], O.DT[dt], op[1], name)
        prog.freeze()
        ms = sorted(lib.run_timed(prog, torch.cuda.current_stream().cuda_stream)[1:])
        t = ms[len(ms) // 2]
        fl = 4.0 * B * H * tq * tk * d
        print("%-28s %8.3f ms  %8.1f TF" % (name, t, fl / (t * 1e-3) / 1e12), flush=True)


if __name__ == "__main__":
    main()
