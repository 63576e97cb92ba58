"""Headline benchmark: 512x512 images/s of the Pix2Pix_Turbo forward (BASELINE.json configs[1]:
pix2pix-turbo edge_to_image, bf16, bs=8 per GPU) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path (VAE encode -> UNet @ t=999 -> DDPM step -> VAE decode with skips) over
one batch of synthetic inputs already resident in HBM, replayed as a hipGraph; for N > 1 each rank runs its
own batch shard (weak scaling, weights replicated, no data-path collective) and the finished images are
gathered to rank 0 over RCCL inside the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

F_ALG = {512: 4.477e12, 1024: 20.22e12}   # algorithmic FLOP / image
PEAK_TF = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}   # dense MFMA peaks, MI355X_MICROARCH.md
CPU_BASELINE_THREADS = 16
PER_OP_PATH = None
DTYPES = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}


def synth_inputs(B, size, cross_dim, lat, device, seed):
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(B, 1, size, size, generator=g) < 0.08).float().expand(B, 3, size, size).contiguous()
    cap = torch.randn(1, 77, cross_dim, generator=g)
    eps = torch.randn(B, lat, size // 8, size // 8, generator=g)
    return x.to(device), cap.to(device), eps.to(device)


def kernel_roofline(plan, dtype_name, reps=2):
    """HIP-event time of every launch of the program (i2i_run_timed: events recorded on the stream the kernels
    run on), grouped by the HIP kernel each op resolves to (plan.op_kernel).  The dominant kernel is the halo-tiled
    3x3 convolution (csrc/conv3x3.hip): `achieved` = its ALGORITHMIC FLOPs (2*9*Cin*Cout per output pixel, SURVEY
    Appendix B) / its measured time; the sub-pixel upsampler launches execute 4/9 of their algorithmic MACs."""
    ms = None
    for _ in range(reps):
        cur = plan.run_timed()
        ms = cur if ms is None else [min(a, b) for a, b in zip(ms, cur)]
    tot = sum(ms)
    fam = {}
    for name, t, fl in zip(plan.op_kernel, ms, plan.op_flops):
        f = fam.setdefault(name, [0.0, 0.0, 0])
        f[0] += t
        f[1] += fl
        f[2] += 1
    if PER_OP_PATH:
        rows = sorted(((t, label, kn, fl) for (opc, _dt, _p, label), kn, t, fl in zip(plan.prog.ops, plan.op_kernel, ms, plan.op_flops)), reverse=True)
        with open(PER_OP_PATH, "w") as f:
            for t, label, kn, fl in rows:
                f.write("%8.4f ms  %7.1f TF  %-28s %s\n" % (t, fl / (t * 1e-3) / 1e12 if t > 0 else 0.0, kn[:28], label))
    halo = [k for k in fam if k.startswith("conv3x3_halo_kernel")]
    t3 = sum(fam[k][0] for k in halo)
    f3 = sum(fam[k][1] for k in halo)
    n3 = sum(fam[k][2] for k in halo)
    achieved = f3 / (t3 * 1e-3) / 1e12
    peak = PEAK_TF[dtype_name]
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r1_traffic_conv3x3_halo.json")     # written by tools/pmc_traffic.py from a PMC run of THIS command
    if os.path.exists(tpath):
        with open(tpath) as f:
            tj = json.load(f)
        if tj.get("batch") == plan.B and tj.get("dtype") == dtype_name:
            traffic = tj["hbm_bytes_per_launch"]
    roof = {"bound": "mfma", "kernel": "conv3x3_halo_kernel (halo-tiled 3x3 implicit-GEMM conv, incl. sub-pixel upsampler form)",
            "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
            "traffic_unit": "bytes per launch (2*FETCH_SIZE + WRITE_SIZE, rocprofv3 PMC, MI355X_MICROARCH.md HBM section)",
            "launches": n3, "avg_launch_ms": round(t3 / n3, 4), "share_of_step_time": round(t3 / tot, 3),
            "algorithmic_flops_per_launch": f3 / n3, "executed_mfma_flops_per_step": getattr(plan, "halo_flops_real", None)}
    breakdown = {k: {"ms": round(v[0], 3), "launches": v[2], "tflops": round(v[1] / (v[0] * 1e-3) / 1e12, 1) if v[1] else None}
                 for k, v in sorted(fam.items(), key=lambda kv: -kv[1][0])}
    return roof, breakdown, tot


def cpu_baseline(weights, size, seed):
    """The CPU oracle (pure-PyTorch fp32 restatement of the reference path) timed on this host's cores on a
    bounded sample: one 512x512 image, one forward."""
    from oracle.pipeline import ModelWeights, pix2pix_forward
    mw = ModelWeights(weights.unet, weights.vae, weights.unet_arch, weights.vae_arch, weights.unet_scaling, weights.vae_scaling)
    x, cap, eps = synth_inputs(1, size, weights.unet_arch.cross_attention_dim, weights.vae_arch.latent_channels, "cpu", seed)
    # oneDNN/OpenMP oversubscribe badly past a few dozen threads on these shapes (256 threads: 488 s for one
    # forward on the 256-core GPU box, 8 threads: 32 s): cap the pool and report the cap as `cores`
    torch.set_num_threads(min(CPU_BASELINE_THREADS, os.cpu_count() or 1))
    t0 = time.time()
    out = pix2pix_forward(mw, x, cap, eps)
    dt = time.time() - t0
    return {"value": round(1.0 / dt, 4), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "1 image %dx%d, fp32, unmerged LoRA, one forward (%.1f s)" % (size, size, dt)}, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU per step")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--dtype", default="bf16", choices=list(DTYPES))
    ap.add_argument("--arch", default="sd-turbo", choices=["sd-turbo", "tiny"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--per-op", default=None, help="write the per-launch timing table (label, ms, TF) to this file")
    a = ap.parse_args()
    global PER_OP_PATH
    PER_OP_PATH = a.per_op

    from img2img_turbo_amd import dp
    from img2img_turbo_amd.arch import SD_TURBO_UNET, SD_TURBO_VAE, TINY_UNET, TINY_VAE
    from img2img_turbo_amd.pix2pix_turbo import Pix2Pix_Turbo
    from img2img_turbo_amd.synth import make_pix2pix_weights

    rank, world, local = dp.init_from_env()
    assert world == a.gpus, "launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (a.gpus, world)
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local)
    dev = "cuda:%d" % local
    dtype = DTYPES[a.dtype]
    ua, va = (SD_TURBO_UNET, SD_TURBO_VAE) if a.arch == "sd-turbo" else (TINY_UNET, TINY_VAE)

    weights = make_pix2pix_weights(ua, va, seed=1234 + 2)           # random init of the exact architecture
    model = Pix2Pix_Turbo(weights=weights, device=dev, dtype=dtype)
    B = a.batch
    x, cap, eps = synth_inputs(B, a.size, ua.cross_attention_dim, va.latent_channels, dev, 1234 + 2 + rank)
    plan = model.get_plan(B, a.size, a.size)
    plan.x_in.copy_(x)
    plan.ctx.copy_(cap.to(dtype))
    plan.eps.copy_(eps)
    total = B * world

    def step():
        plan.replay()
        if world > 1 and not a.no_gather:
            dp.gather_images(plan.out, total, dst=0)

    for _ in range(a.warmup):
        step()
    dp.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    dp.barrier()
    elapsed = dp.max_over_ranks(time.perf_counter() - t0, dev)
    ms_per_step = elapsed / a.steps * 1e3
    value = total * a.steps / elapsed

    if rank != 0:
        return
    rec = {"metric": "512x512 images/sec (whole node), pix2pix-turbo edge_to_image forward", "value": round(value, 3),
           "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": round(value / 9.09, 3) if (a.size == 512 and a.arch == "sd-turbo") else None,
           "dtype": a.dtype, "data": "synthetic (random-init weights of the SD-Turbo architecture + LoRA r8/r4, Bernoulli edge maps)",
           "config": {"workload": "pix2pix-turbo edge_to_image %s bs=%d/GPU %dx%d, deterministic path, hipGraph replay" % (a.dtype, B, a.size, a.size),
                      "global_batch": total, "parallelism": "dp%d (batch shards, replicated weights, RCCL gather of outputs)" % world,
                      "arch": a.arch, "kernel_library": os.path.basename(model.lib.path)},
           "baseline_note": "vs_baseline = value / 9.09 img/s (0.11 s per 512x512 image on A100, reference README.md:17)"}
    falg = F_ALG.get(a.size)
    if falg and a.arch == "sd-turbo":
        rec["e2e_mfma_frac"] = round(value / world * falg / 1e12 / PEAK_TF[a.dtype], 4)
    if world == 1:
        roof, breakdown, tot = kernel_roofline(plan, a.dtype)
        rec["roofline"] = roof
        rec["kernel_breakdown_ms"] = breakdown
        rec["sum_kernel_ms"] = round(tot, 3)
        if not a.no_latency:
            p1 = model.get_plan(1, a.size, a.size)
            p1.x_in.copy_(x[:1]); p1.ctx.copy_(cap.to(dtype)); p1.eps.copy_(eps[:1])
            for _ in range(3):
                p1.replay()
            torch.cuda.synchronize()
            lat = []
            for _ in range(20):
                t = time.perf_counter()
                p1.replay()
                torch.cuda.synchronize()
                lat.append((time.perf_counter() - t) * 1e3)
            rec["latency_bs1_ms_p50"] = round(statistics.median(lat), 3)
            rec["images_per_s_bs1"] = round(1e3 / statistics.median(lat), 2)
        if not a.no_cpu_baseline:
            cb, _ = cpu_baseline(weights, a.size, 1234 + 2)
            rec["cpu_baseline"] = cb
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
