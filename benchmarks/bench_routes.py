"""Latency of the planned forward for several planner routings (halo_min_tiles: fewer halo-conv tiles than this go to the
LDS-DMA igemm + split-K instead).  python benchmarks/bench_routes.py --batch 1 --values 0,64,96,128,160,256"""
import argparse
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from img2img_turbo_amd.arch import SD_TURBO_UNET, SD_TURBO_VAE  # noqa: E402
from img2img_turbo_amd.pix2pix_turbo import Pix2Pix_Turbo  # noqa: E402
from img2img_turbo_amd.synth import make_pix2pix_weights  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--values", default="0,64,96,128,160,256")
ap.add_argument("--size", type=int, default=512)
a = ap.parse_args()
w = make_pix2pix_weights(SD_TURBO_UNET, SD_TURBO_VAE, seed=1236)
for v in [int(x) for x in a.values.split(",")]:
    model = Pix2Pix_Turbo(weights=w, device="cuda", dtype=torch.bfloat16, plan_options=dict(halo_min_tiles=v))
    plan = model.get_plan(a.batch, a.size, a.size)
    plan.x_in.normal_(); plan.ctx.normal_(); plan.eps.normal_()
    for _ in range(3):
        plan.replay()
    torch.cuda.synchronize()
    lat = []
    for _ in range(30):
        t = time.perf_counter(); plan.replay(); torch.cuda.synchronize(); lat.append((time.perf_counter() - t) * 1e3)
    print("halo_min_tiles %4d  batch %d: p50 %.3f ms  (%d launches)" % (v, a.batch, statistics.median(lat), plan.prog.n), flush=True)
    model.release_plans()
    del model, plan
    torch.cuda.empty_cache()
