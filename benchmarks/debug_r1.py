"""Throw-away diagnostics: (1) which entries are NaN in the fused-GN-partials case, (2) which op of the SD-Turbo
program faults (ops executed one by one with a sync in between)."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from img2img_turbo_amd import _capi as K, ops as O
import opcheck as oc

lib = K.default_library()
dev = "cuda"
which = sys.argv[1] if len(sys.argv) > 1 else "part"
if which == "part":
    for cfg in (12, 18, 13):
        for (n, cin, cout, h, w, groups, res) in ((2, 128, 128, 72, 40, 32, True), (1, 64, 512, 32, 32, 32, False)):
            dtype = torch.bfloat16
            g = torch.Generator().manual_seed(0)
            x = torch.randn(n, cin, h, w, generator=g)
            wt = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
            b = torch.randn(cout, generator=g) * 0.1
            r = torch.randn(n, cout, h, w, generator=g) if res else None
            x0 = oc.nhwc(x, dtype).to(dev); wp = oc.pack_conv_weight(wt, dtype).to(dev)
            out = torch.full((n, h, w, cout), float("nan"), dtype=dtype, device=dev)
            rd = oc.nhwc(r, dtype).to(dev) if res else None
            opcode, p = O.conv(x0, wp, out, nimg=n, hin=h, win=w, ho=h, wo=w, ks=3, stride=1, pad=1, N=cout, bias=b.to(dev), res=rd, tile=cfg)
            parts = lib.igemm_gn_parts(p, O.DT[dtype], groups)
            part = torch.full((n * parts * groups * 2,), float("nan"), device=dev)
            p.gn_part, p.gn_part_groups = part.data_ptr(), groups
            prog = K.Program(); prog.add(opcode, O.DT[dtype], p); prog.freeze()
            lib.run(prog, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
            o = out.float().cpu(); pt = part.cpu().view(n, parts, groups, 2)
            nan_o = torch.isnan(o).nonzero()
            nan_p = torch.isnan(pt).nonzero()
            print("cfg", cfg, "shape", (n, cin, cout, h, w), "parts", parts, "out NaN", len(nan_o), nan_o[:4].tolist(), "part NaN", len(nan_p), nan_p[:6].tolist(), flush=True)
else:
    sys.argv = [sys.argv[0]]
    from img2img_turbo_amd.arch import SD_TURBO_UNET, SD_TURBO_VAE
    from img2img_turbo_amd.pix2pix_turbo import Pix2Pix_Turbo
    from img2img_turbo_amd.synth import make_pix2pix_weights
    w = make_pix2pix_weights(SD_TURBO_UNET, SD_TURBO_VAE, seed=1236)
    model = Pix2Pix_Turbo(weights=w, device="cuda:0", dtype=torch.bfloat16)
    plan = model.get_plan(8, 512, 512)
    plan.x_in.copy_((torch.rand(8, 1, 512, 512) < 0.08).float().expand(8, 3, 512, 512).cuda())
    plan.ctx.copy_(torch.randn(1, 77, 1024).cuda().to(torch.bfloat16)); plan.eps.copy_(torch.randn(8, 4, 64, 64).cuda())
    ops = plan.prog.ops
    for i, (opc, dt, p, label) in enumerate(ops):
        print(i, opc, label, flush=True)
        pr = K.Program(); pr.add(opc, dt, p, label); pr.freeze()
        lib.run(pr, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
    print("all ops ran")
