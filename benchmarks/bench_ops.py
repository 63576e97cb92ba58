"""Micro-benchmark of the implicit-GEMM kernel on the conv / linear shapes of the 512x512 forward.

    python benchmarks/bench_ops.py [--dtype bf16] [--batch 8] [--out gpurun_out/bench_ops.json]

Uses HIP events around single launches (i2i_run_timed), random (not zero) operands.
"""
import argparse
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from img2img_turbo_amd import _capi as K, ops as O  # noqa: E402

SHAPES = [  # (name, cin, cout, H, W, ks, stride, ups, gn)
    ("vae 128->128@512 gn", 128, 128, 512, 512, 3, 1, 0, 1),
    ("vae 128->128@512", 128, 128, 512, 512, 3, 1, 0, 0),
    ("vae 256->256@256 gn", 256, 256, 256, 256, 3, 1, 0, 1),
    ("vae 512->512@128 gn", 512, 512, 128, 128, 3, 1, 0, 1),
    ("vae 512->512@64 gn", 512, 512, 64, 64, 3, 1, 0, 1),
    ("vae up 256->256 @256->512", 256, 256, 256, 256, 3, 1, 1, 0),
    ("vae up 512->512 @128->256", 512, 512, 128, 128, 3, 1, 1, 0),
    ("vae 256->128@512 gn", 256, 128, 512, 512, 3, 1, 0, 1),
    ("vae conv_in 3->128@512", 8, 128, 512, 512, 3, 1, 0, 0),
    ("vae conv_out 128->3@512 gn", 128, 3, 512, 512, 3, 1, 0, 1),
    ("vae down 128@512 s2", 128, 128, 512, 512, 3, 2, 0, 0),
    ("vae down 256@256 s2", 256, 256, 256, 256, 3, 2, 0, 0),
    ("vae down 512@128 s2", 512, 512, 128, 128, 3, 2, 0, 0),
    ("unet down 320@64 s2", 320, 320, 64, 64, 3, 2, 0, 0),
    ("unet down 640@32 s2", 640, 640, 32, 32, 3, 2, 0, 0),
    ("unet down 1280@16 s2", 1280, 1280, 16, 16, 3, 2, 0, 0),
    ("unet 320->320@64 gn", 320, 320, 64, 64, 3, 1, 0, 1),
    ("unet 640->640@32 gn", 640, 640, 32, 32, 3, 1, 0, 1),
    ("unet 960->320@64 gn", 960, 320, 64, 64, 3, 1, 0, 1),
    ("unet 1280->640@32 gn", 1280, 640, 32, 32, 3, 1, 0, 1),
    ("unet 1280->1280@16 gn", 1280, 1280, 16, 16, 3, 1, 0, 1),
    ("unet 1280->1280@8 gn", 1280, 1280, 8, 8, 3, 1, 0, 1),
    ("unet 2560->1280@8 gn", 2560, 1280, 8, 8, 3, 1, 0, 1),
    ("unet lin 320->2560 T4096", 320, 2560, 64, 64, 1, 1, 0, 0),
    ("unet lin 1280->1280 T256", 1280, 1280, 16, 16, 1, 1, 0, 0),
    ("vae skip 128->256@512 1x1", 128, 256, 512, 512, 1, 1, 0, 0),
    ("unet lin 320->320 T4096", 320, 320, 64, 64, 1, 1, 0, 0),
    ("unet lin 1280->320 T4096", 1280, 320, 64, 64, 1, 1, 0, 0),
    ("unet lin 640->5120 T1024", 640, 5120, 32, 32, 1, 1, 0, 0),
    ("unet lin 2560->640 T1024", 2560, 640, 32, 32, 1, 1, 0, 0),
    ("unet lin 1280->10240 T256", 1280, 10240, 16, 16, 1, 1, 0, 0),
    ("unet lin 5120->1280 T256", 5120, 1280, 16, 16, 1, 1, 0, 0),
    ("vae lin 512->1024 T4096", 512, 1024, 64, 64, 1, 1, 0, 0),
    ("vae lin 512->512 T4096", 512, 512, 64, 64, 1, 1, 0, 0),
    ("vae sc 256->128@512 1x1", 256, 128, 512, 512, 1, 1, 0, 0),
    ("vae sc 512->256@256 1x1", 512, 256, 256, 256, 1, 1, 0, 0),
    ("unet lin 640->640 T1024", 640, 640, 32, 32, 1, 1, 0, 0),
    ("unet lin 1280->1280 T256", 1280, 1280, 16, 16, 1, 1, 0, 0),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--tiles", default="0")
    ap.add_argument("--out", default="gpurun_out/bench_ops.json")
    ap.add_argument("--only", default="", help="substring filter on shape names")
    ap.add_argument("--lib", default=None, help="alternate build of the library (ablation experiments)")
    ap.add_argument("--splitk", default="0", help="comma list of split-K factors to try (LDS-DMA igemm)")
    ap.add_argument("--trace", action="store_true", help="library built with -DI2I_TRACE=1: print the per-segment cycle split of the halo conv")
    ap.add_argument("--subpix", action="store_true", help="upsampler shapes (ups = 1) in the sub-pixel form the product uses ([4][N][2][2][cin] weights)")
    ap.add_argument("--res", action="store_true", help="add a residual tensor in the epilogue (the resnets' conv2)")
    ap.add_argument("--nogn", action="store_true", help="drop the GroupNorm prologue (paths that need a materialised input)")
    ap.add_argument("--geglu", action="store_true", help="1x1 shapes with N % 32 == 0: GEGLU epilogue (ff.net.0 of the transformer blocks; output N/2 columns)")
    ap.add_argument("--fill", default="randn", choices=["randn", "zero", "const"], help="operand fill: the chip clocks to its power budget, and data toggling is part of it "
                    "(MI355X_MICROARCH.md 'DVFS give-back'): zero / constant operands show how much of a kernel's time is clock, not cycles")
    a = ap.parse_args()
    dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[a.dtype]
    peak = 157.3 if a.dtype == "f32" else 2500.0
    lib = K.Library(a.lib) if a.lib else K.default_library()
    dev = "cuda"
    res = []
    for name, cin, cout, H, W, ks, stride, ups, gn in SHAPES:
        if a.only and not any(t in name for t in a.only.split(',')):
            continue
        B = a.batch
        x = torch.randn(B, H, W, cin, device=dev).to(dt)
        w = (torch.randn(cout, ks * ks * cin, device=dev) / math.sqrt(ks * ks * cin)).to(dt)
        if a.fill != "randn":
            x = torch.full_like(x, 0.0 if a.fill == "zero" else 0.5)
            w = torch.full_like(w, 0.0 if a.fill == "zero" else 0.01)
        sp = 1 if (a.subpix and ups == 1 and ks == 3) else 0
        if sp:
            w = (torch.randn(4 * cout, 4 * cin, device=dev) / math.sqrt(4 * cin)).to(dt)
        ho, wo = (H << ups) // stride, (W << ups) // stride
        gg = 1 if (a.geglu and ks == 1 and cout % 32 == 0) else 0
        coutp = (cout // 2 if gg else (cout + 7) // 8 * 8)
        out = torch.empty(B, ho, wo, coutp, device=dev, dtype=dt)
        bias = torch.randn(cout, device=dev)
        resid = torch.randn(B, ho, wo, coutp, device=dev).to(dt) if a.res else None
        gn = 0 if a.nogn else gn
        ss = torch.randn(B, cin, 2, device=dev) if gn else None
        for tile, sk in [(int(t), int(k)) for t in a.tiles.split(",") for k in a.splitk.split(",")]:
            # trace builds reuse the split-K field of the halo conv as ablation bits: no split-K slab then
            ws = torch.empty(sk * B * ho * wo * cout, device=dev) if (sk > 1 and not (a.lib and "trace" in a.lib)) else None
            if a.trace:
                ws = torch.zeros(B * ho * wo * ((cout + 63) // 64) // 4, dtype=torch.int32, device=dev)   # >= workgroups * waves * 16
            prog = K.Program()
            for _ in range(a.iters + 1):
                prog.add(*_op(O.conv(x, w, out, nimg=B, hin=H, win=W, ho=ho, wo=wo, ks=ks, stride=stride, pad=ks // 2, ups=ups,
                                      N=cout, gn_ss=ss, act=1 if gn else 0, bias=bias, tile=tile, splitk=sk, ws=ws, subpix=sp,
                                      res=resid, ldr=coutp if a.res else None, geglu=gg, ldc=coutp), dt))
            prog.freeze()
            try:
                ms = lib.run_timed(prog, torch.cuda.current_stream().cuda_stream)[1:]
            except Exception as e:      # forced tile id not applicable to this shape
                print("%-32s tile %d  n/a (%s)" % (name, tile, str(e)[:60]), flush=True)
                continue
            t = sorted(ms)[len(ms) // 2]
            fl = 2.0 * B * ho * wo * cout * ks * ks * cin
            tf = fl / (t * 1e-3) / 1e12
            rec = dict(name=name + (" geglu" if gg else ""), tile=tile, splitk=sk, ms=t, tflops=tf, frac=tf / peak, dtype=a.dtype, batch=B)
            res.append(rec)
            print("%-32s tile %d sk %d  %8.3f ms  %8.1f TF  (%.1f%% of peak)" % (rec["name"], tile, sk, t, tf, 100 * tf / peak), flush=True)
            if a.trace:
                tr = ws.view(-1, 16).cpu().double()
                tr = tr[tr.sum(1) > 0]
                tot = tr.sum(1).mean().item()
                w32 = tile >= 40 or (tile == 0 and tr[:, 8:].sum() == 0)       # the wide-tile kernel writes 8 segments, the halo kernel 16
                names = ["prologue", "k16 steps 0-2", "vmcnt wait", "step barrier", "window+k16 step 3", "slab barrier+reads", "epilogue", "-"] + ["-"] * 8 if w32 else ["kgroup0", "vmcnt", "barrier", "issue", "kgroup1", "handover", "pro:setup+w0read", "epi:gnstats", "pro:setup+issue", "pro:vmcnt0", "pro:barrier", "pro:halo_store", "pro:barrier2",
                         "epi:tail_vmcnt", "epi:bias+store", "-"]
                print("   trace: %d waves, %.0f cycles/wave (%.2f GHz if the wave spans the launch): " % (len(tr), tot, tot / (t * 1e-3) / 1e9) +
                      "  ".join("%s %.1f%%" % (n, 100 * tr[:, i].mean().item() / tot) for i, n in enumerate(names)), flush=True)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)


def _op(op, dt):
    return op[0], O.DT[dt], op[1]


if __name__ == "__main__":
    main()
