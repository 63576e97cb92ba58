#!/bin/bash
# The C host (examples/plan_host.c) on hardware: export a bf16 batch-8 512x512 pix2pix plan, build the host with gcc against libi2i_turbo.so,
# run it on raw input files and compare its images with the Python replay.
O=gpurun_out; mkdir -p $O /tmp/ph; export TMPDIR=/tmp
python - <<'PY'
import torch, sys, time
sys.path.insert(0, ".")
import bench
from img2img_turbo_amd.arch import SD_TURBO_UNET, SD_TURBO_VAE
from img2img_turbo_amd.pix2pix_turbo import Pix2Pix_Turbo
from img2img_turbo_amd.synth import make_pix2pix_weights
from img2img_turbo_amd.plan_file import export_plan
w = make_pix2pix_weights(SD_TURBO_UNET, SD_TURBO_VAE, seed=1236)
x, cap, eps, _ = bench.synth_inputs("canny", 8, 512, SD_TURBO_UNET.cross_attention_dim, SD_TURBO_VAE.latent_channels, 1236)
m = Pix2Pix_Turbo(weights=w, device="cuda:0", dtype=torch.bfloat16)
out = m(x.cuda(), caption_enc=cap.cuda(), eps=eps.cuda()).cpu()
plan = list(m._plans.values())[0]
t = time.time(); info = export_plan(plan, "/tmp/ph/p.i2iplan"); print("export: %.1f s" % (time.time() - t), {k: v for k, v in info.items() if k != "io"}, info["io"])
for n, tns in (("x", x.to(plan.x_in.dtype)), ("ctx", cap.to(plan.ctx.dtype).reshape(plan.ctx.shape)), ("eps", eps.to(plan.eps.dtype))):
    open("/tmp/ph/%s.bin" % n, "wb").write(tns.contiguous().view(torch.uint8).numpy().tobytes())
torch.save(plan.out.cpu(), "/tmp/ph/out_python.pt")      # (the plan's own output buffer: bf16 NCHW, what the host reads)
PY
gcc -O2 -Wall -I include examples/plan_host.c -o /tmp/ph/plan_host -L img2img-turbo_amd/csrc -li2i_turbo -Wl,-rpath,$PWD/img2img-turbo_amd/csrc
( time /tmp/ph/plan_host /tmp/ph/p.i2iplan /tmp/ph/x.bin /tmp/ph/ctx.bin /tmp/ph/eps.bin /tmp/ph/out.bin ) 2>&1 | grep -v amdgpu.ids
python - <<'PY'
import torch
out = torch.load("/tmp/ph/out_python.pt")
got = torch.frombuffer(bytearray(open("/tmp/ph/out.bin", "rb").read()), dtype=out.dtype).reshape(out.shape)
print("C host vs Python replay: equal =", bool(torch.equal(got.float(), out.float())), " max abs diff", float((got.float() - out.float()).abs().max()))
PY
ls -la /tmp/ph/p.i2iplan | awk '{print "plan file bytes:", $5}'
