"""Diagnostic: run-to-run reproducibility of the sliced GroupNorm statistics kernel (tests/opcheck.py check_gn_stats, sliced=True)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import opcheck as oc
from img2img_turbo_amd import _capi as K, ops as O
lib = K.default_library()
for dtype in (torch.float32, torch.bfloat16):
    for (n, c, h, w, nparts) in ((1, 320, 64, 64, 20), (8, 320, 64, 64, 20), (1, 512, 64, 64, 32)):
        g = torch.Generator().manual_seed(0)
        x = torch.randn(n, c, h, w, generator=g) * 1.5 + 0.3
        gamma = 1 + 0.1 * torch.randn(c, generator=g); beta = 0.1 * torch.randn(c, generator=g)
        ref = oc.gn_scale_shift(x.to(dtype).float(), 32, gamma, beta, 1e-5)
        x0 = oc.nhwc(x, dtype).cuda()
        partial = torch.zeros(n * nparts * 32 * 2, device="cuda")
        ss = torch.full((n, c, 2), float("nan"), device="cuda")
        counters = torch.zeros(n * 32, dtype=torch.int32, device="cuda")
        opcode, p = O.gn_stats(x0, gamma.cuda(), beta.cuda(), partial, ss, nimg=n, hw=h * w, groups=32, eps=1e-5, nparts=nparts, c0=c, counters=counters)
        outs = []
        for it in range(6):
            ss.fill_(float("nan"))
            oc.run_op(lib, opcode, p, dtype, "cuda")
            outs.append(ss.clone())
        for it, o in enumerate(outs):
            d = (o - outs[0]).abs()
            print(dtype, (n, c, h, w), "run", it, "nan", int(torch.isnan(o).sum()), "max diff vs run 0 %.3e" % float(torch.nan_to_num(d).max()),
                  "n differing", int((o != outs[0]).sum()), "rel err vs ref %.2e" % oc.rel_err(o.cpu(), ref), "counters", int(counters.abs().sum()))
