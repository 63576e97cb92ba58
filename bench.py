"""Headline benchmark: 512x512 images/s of the Pix2Pix_Turbo / CycleGAN_Turbo generator forward on N MI355X GPUs.

    python bench.py                                    # BASELINE configs[1]: pix2pix-turbo edge_to_image bf16 bs=8 512^2, 1 GPU
    python bench.py --gpus 8                           # spawns 8 ranks itself (torch.distributed.run, 127.0.0.1), weak scaling
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W         # what the driver runs for N > 1
    python bench.py --model cyclegan --batch 4         # configs[2] per-GPU share (day_to_night bf16, 4 images / GPU)
    python bench.py --stochastic --gamma 0.4 --batch 16          # configs[3] (sketch_to_image_stochastic, TwinConv, r-scaled LoRA)
    python bench.py --size 1024 --dtype f16 --batch 8  # configs[4] per-GPU share

A step = one pass of the hot path (VAE encode -> UNet @ t=999 -> DDPM step -> VAE decode with skips) over one batch
of synthetic inputs already resident in HBM, replayed as a hipGraph; for N > 1 each rank runs its own batch shard
(weak scaling, weights replicated, no data-path collective) and the finished images are gathered to rank 0 over RCCL
inside the timed region.  Rank 0 prints ONE JSON line.  At N = 1 the line also carries the roofline of the dominant
kernel (HIP events on the launch stream), the bs=1 latency, the CPU oracle timed on image 0 of the SAME inputs, and
the parity of the GPU output against that oracle image (``parity_max_abs`` / ``parity_psnr_db``).
"""
import argparse
import hashlib
import json
import os
import socket
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
CPU_BASELINE_TIMED = 2        # + 1 warm-up: ~30 s of CPU work in the default run
PER_OP_PATH = None
DTYPES = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "traffic_conv3x3.json")   # tools/pmc_traffic.py, PMC passes of THIS command (every conv3x3_* launch)
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E spec peak (6.3 TB/s achievable)


def synth_inputs(kind, B, size, cross_dim, lat, seed):
    """Seeded synthetic inputs on the CPU: 'canny' Bernoulli(0.08) edge maps in {0,1}, 'sketch' Bernoulli(0.05),
    'photo' low-passed U(-1,1).  The caption embedding is drawn first so image i is the same for every batch size."""
    g = torch.Generator().manual_seed(seed)
    cap = torch.randn(1, 77, cross_dim, generator=g)
    if kind == "photo":
        x = torch.rand(B, 3, size, size, generator=g) * 2 - 1
        x = torch.nn.functional.avg_pool2d(x, 9, 1, 4)
        x = (x / x.abs().amax(dim=(1, 2, 3), keepdim=True)).contiguous()
    else:
        p = 0.08 if kind == "canny" else 0.05
        x = (torch.rand(B, 1, size, size, generator=g) < p).float().expand(B, 3, size, size).contiguous()
    eps = torch.randn(B, lat, size // 8, size // 8, generator=g)
    noise = torch.randn(B, lat, size // 8, size // 8, generator=g)
    return x, cap, eps, noise


def source_hash():
    """sha256 over the kernel sources: a PMC traffic record is only replayed into the line for the build it was taken on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "img2img-turbo_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h")):
            with open(os.path.join(d, fn), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


def kernel_roofline(plan, dtype_name, reps=2):
    """HIP-event time of every launch of the program (i2i_run_timed: events recorded on the stream the kernels
    run on), grouped by the HIP kernel each op resolves to (plan.op_kernel = i2i_igemm_route for the contractions).
    The dominant family is the 3x3 stride-1 convolution of the VAE: `conv3x3_w32_kernel` (csrc/conv3x3_w32.hip, wide tiles on
    32x32x16 MFMA, incl. its sub-pixel Upsample2D form) plus `conv3x3_halo_kernel` (csrc/conv3x3.hip) for what the wide tiles
    do not take (since round 4 only the three conv_out launches: the UNet's 3x3 convolutions run on `gemm_w32_kernel`, whose
    rate is in `kernel_breakdown_ms`).  `achieved` = the family's ALGORITHMIC FLOPs (2*9*Cin*Cout per output
    pixel) / its measured time; `executed` prices the sub-pixel launches at the 4/9 of the 3x3 MACs
    the matrix pipe really runs (plan.op_flops_exec).  Every other MFMA family gets the same two numbers in
    `kernel_breakdown_ms` (attention: 4*heads*Tq*Tk*d per image)."""
    ms = None
    for _ in range(reps):
        cur = plan.run_timed()
        ms = cur if ms is None else [min(a, b) for a, b in zip(ms, cur)]
    tot = sum(ms)
    fam = {}
    for name, t, fl, fx, nb in zip(plan.op_kernel, ms, plan.op_flops, plan.op_flops_exec, plan.op_bytes):
        f = fam.setdefault(name, [0.0, 0.0, 0, 0.0, 0.0])
        f[0] += t
        f[1] += fl
        f[2] += 1
        f[3] += fx
        f[4] += nb
    if PER_OP_PATH:
        rows = sorted(((t, label, kn, fl) for (opc, _dt, _p, label), kn, t, fl in zip(plan.prog.ops, plan.op_kernel, ms, plan.op_flops)), reverse=True)
        with open(PER_OP_PATH, "w") as f:
            for t, label, kn, fl in rows:
                f.write("%8.4f ms  %7.1f TF  %-28s %s\n" % (t, fl / (t * 1e-3) / 1e12 if t > 0 else 0.0, kn[:28], label))
    halo = [k for k in fam if k.startswith("conv3x3_")]       # conv3x3_w32_kernel + conv3x3_halo_kernel[<SUBPIX>]
    t3 = sum(fam[k][0] for k in halo)
    f3 = sum(fam[k][1] for k in halo)
    n3 = sum(fam[k][2] for k in halo)
    achieved = f3 / (t3 * 1e-3) / 1e12
    # the sub-pixel upsampler form executes 4/9 of the 3x3 MACs of upsample + conv (a folded 1x1 skip conv in full): the matrix-pipe rate
    f3x = sum(fam[k][3] for k in halo)
    executed = f3x / (t3 * 1e-3) / 1e12
    peak = PEAK_TF[dtype_name]
    traffic, traffic_src = None, None
    if os.path.exists(TRAFFIC_JSON):
        with open(TRAFFIC_JSON) as f:
            tj = json.load(f)
        # only for the build and workload the counters were collected on; anything else reports null rather than a stale number
        if tj.get("batch") == plan.B and tj.get("dtype") == dtype_name and tj.get("size") == plan.H and tj.get("source_hash") == source_hash():
            traffic, traffic_src = tj["hbm_bytes_per_launch"], tj.get("collected")
    roof = {"bound": "mfma", "kernel": "conv3x3_w32_kernel[<SUBPIX>] + conv3x3_halo_kernel[<SUBPIX>] (the conv3x3_* launches of the step: the VAE's 3x3 stride-1 convs on the "
                                       "wide-tile 32x32x16-MFMA kernel incl. the sub-pixel upsampler form, plus whatever the halo kernel still takes -- the conv_out "
                                       "launches, small planes at small batch, the exact-f32 mode; the UNet's 3x3 convs run on gemm_w32_kernel: kernel_breakdown_ms)",
            "per_kernel": {k: {"launches": fam[k][2], "avg_launch_ms": round(fam[k][0] / fam[k][2], 4),
                               "tflops": round(fam[k][1] / (fam[k][0] * 1e-3) / 1e12, 1)} for k in halo},
            "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
            "executed": round(executed, 2), "frac_executed": round(executed / peak, 4),
            "executed_note": "achieved counts the ALGORITHMIC FLOPs of the reference ops (9 taps for upsample + conv); executed counts what the "
                             "matrix pipe really does (<SUBPIX> launches: 4/9)",
            "traffic": traffic,
            "traffic_unit": "bytes per launch (2*FETCH_SIZE + WRITE_SIZE, rocprofv3 PMC, MI355X_MICROARCH.md HBM section)",
            "traffic_source": traffic_src,
            "launches": n3, "avg_launch_ms": round(t3 / n3, 4), "share_of_step_time": round(t3 / tot, 3),
            "algorithmic_flops_per_launch": f3 / n3, "executed_mfma_flops_per_step": getattr(plan, "halo_flops_real", None)}
    # MFMA families: algorithmic TFLOP/s against the dense peak; HBM-bound families (norms, layout / latent ops: plan.op_bytes =
    # one read [+ one write] per element): algorithmic GB/s against the 8 TB/s HBM3E peak
    breakdown = {k: {"ms": round(v[0], 3), "launches": v[2], "tflops": round(v[1] / (v[0] * 1e-3) / 1e12, 1) if v[1] else None,
                     "frac_of_mfma_peak": round(v[1] / (v[0] * 1e-3) / 1e12 / peak, 4) if v[1] else None,
                     "gbytes_per_s": round(v[4] / (v[0] * 1e-3) / 1e9, 1) if v[4] else None,
                     "frac_of_hbm_peak": round(v[4] / (v[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if v[4] else None}
                 for k, v in sorted(fam.items(), key=lambda kv: -kv[1][0])}
    return roof, breakdown, tot


def cpu_baseline(a, weights, x, cap, eps, noise):
    """The CPU oracle (pure-PyTorch fp32 restatement of the reference path, unmerged LoRA) timed on this host's cores on a
    bounded sample: image 0 of the benchmarked batch, one warm-up forward + the median of CPU_BASELINE_TIMED timed ones,
    CPU_BASELINE_THREADS threads.  Returns (record, oracle image)."""
    from oracle.pipeline import ModelWeights, cyclegan_forward, pix2pix_forward
    mw = ModelWeights(weights.unet, weights.vae, weights.unet_arch, weights.vae_arch, weights.unet_scaling, weights.vae_scaling,
                      weights.vae_b2a)
    # oneDNN/OpenMP oversubscribe badly past a few dozen threads on these shapes (256 threads: 488 s for one
    # forward on the 256-core GPU box, 8 threads: 32 s): cap the pool and report the cap as `cores`
    torch.set_num_threads(min(CPU_BASELINE_THREADS, os.cpu_count() or 1))

    def fwd():
        if a.model == "cyclegan":
            return cyclegan_forward(mw, x[:1], cap, eps[:1], direction=a.direction)
        if a.stochastic:
            return pix2pix_forward(mw, x[:1], cap, eps[:1], deterministic=False, r=a.gamma, noise_map=noise[:1])
        return pix2pix_forward(mw, x[:1], cap, eps[:1])

    out = fwd()                         # warm-up (oneDNN primitive creation, page faults)
    times = []
    for _ in range(CPU_BASELINE_TIMED):
        t0 = time.time()
        out = fwd()
        times.append(time.time() - t0)
    dt = statistics.median(times)
    return {"value": round(1.0 / dt, 4), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "image 0 of the batch (%dx%d), fp32, unmerged LoRA: 1 warm-up + %d timed forwards, median %.1f s (host has %d cores; "
                      "more threads are slower on these shapes)" % (a.size, a.size, CPU_BASELINE_TIMED, dt, os.cpu_count() or 0)}, out


def add_activation_offset(weights, off):
    """--activation-offset: every VAE resnet's conv2 bias += off (both VAEs of a CycleGAN model), in place, before packing."""
    if not off:
        return 0
    n = 0
    for sd in (weights.vae, getattr(weights, "vae_b2a", None)):
        if sd is None:
            continue
        for k in list(sd):
            if k.endswith(".conv2.bias") or k.endswith(".conv2.base_layer.bias"):
                sd[k] = sd[k] + off
                n += 1
    return n


def _smi_sample():
    """One `amd-smi metric` reading of GPU 0: (socket power W, mean gfx clock MHz over the XCDs, max gfx clock MHz) or None."""
    import subprocess
    try:
        r = subprocess.run(["amd-smi", "metric", "-g", "0", "--power", "--clock", "--json"], capture_output=True, text=True, timeout=20)
        d = json.loads(r.stdout)
        d = (d.get("gpu_data") if isinstance(d, dict) else d)[0]
        pw = d["power"]["socket_power"]["value"]
        ck = [v["clk"]["value"] for k, v in d["clock"].items() if k.startswith("gfx_") and isinstance(v.get("clk", {}).get("value"), (int, float))]
        mx = [v["max_clk"]["value"] for k, v in d["clock"].items() if k.startswith("gfx_") and isinstance(v.get("max_clk", {}).get("value"), (int, float))]
        return float(pw), sum(ck) / len(ck), float(max(mx))
    except Exception:        # tool missing / different schema: the field is simply absent from the line
        return None


def power_and_clock(plan, seconds=3.0):
    """Socket power and shader clock while the benchmarked graph replays back to back (outside the timed region).

    The big MFMA kernels of this path run at the chip's power budget on random data (profiles/r5h_power_probe_*.log: 1.29-1.38 kW
    of the 1.4 kW cap, shader clock 1.8-2.0 GHz of 2.4): the roofline's `peak` is the guide's 2.4 GHz figure, this says what clock
    the measured step actually had.
    """
    import threading
    rows, stop = [], []

    def loop():
        while not stop:
            s = _smi_sample()
            if s:
                rows.append(s)

    th = threading.Thread(target=loop, daemon=True)
    t0 = time.perf_counter()
    th.start()
    while time.perf_counter() - t0 < seconds:
        plan.replay()
        torch.cuda.synchronize()
    stop.append(1)
    th.join(timeout=30)
    rows = rows[1:] if len(rows) > 2 else rows          # the first reading may predate the ramp
    if not rows:
        return None
    n = len(rows)
    return {"socket_w": round(sum(r[0] for r in rows) / n, 1), "gfx_mhz": round(sum(r[1] for r in rows) / n, 1), "gfx_max_mhz": rows[0][2], "samples": n,
            "how": "amd-smi metric --power --clock polled while the same graph replays for %.0f s after the timed region" % seconds}


# Pool reference of the calibration loads: medians over the MI355X boxes this round's calls drew (profiles/r6_calib_boxes.json).  A bench
# line measured on box X is normalised to this reference box family by family (see `calibrate`); raw `value` stays the headline.
CALIB_REF = {"mfma_tflops": None, "hbm_tbytes_per_s": None, "graph_node_us": None}
CALIB_REF_PATH = os.path.join(ROOT, "profiles", "calib_reference.json")
if os.path.exists(CALIB_REF_PATH):
    with open(CALIB_REF_PATH) as _f:
        CALIB_REF.update({k: v for k, v in json.load(_f).items() if k in CALIB_REF})


def _gpu_local_cpus(index=0):
    """CPUs on the NUMA node the GPU hangs off (sysfs local_cpulist of its PCI function), or None."""
    try:
        pr = torch.cuda.get_device_properties(index)
        bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        with open("/sys/bus/pci/devices/%s/local_cpulist" % bdf) as f:
            txt = f.read().strip()
        cpus = set()
        for part in txt.split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        return (cpus or None), txt
    except Exception:
        return None, None


def _smi_json(*args):
    import subprocess
    try:
        r = subprocess.run(["amd-smi"] + list(args) + ["--json"], capture_output=True, text=True, timeout=20)
        d = json.loads(r.stdout)
        return (d.get("gpu_data") if isinstance(d, dict) else d)[0]
    except Exception:
        return None


def calibrate(lib, dev, breakdown=None, ms_per_step=None, batch=None):
    """What THIS box delivers on three elementary loads (csrc/calib.hip), outside the timed region and in about a second: a bare
    v_mfma_f32_32x32x16_bf16 loop on random operands (one wave per SIMD: matrix pipe + power management), a 2 x 256 MiB copy stream (HBM),
    the replay of a 1000-node hipGraph of empty kernels (launch path: microseconds per node); plus the clocks / partition / host facts
    amd-smi and sysfs report.  The boxes of a pool differ by +-8 % on the same binary: with a reference (profiles/calib_reference.json)
    the step time is normalised FAMILY BY FAMILY -- MFMA-bound kernel families by the MFMA ratio, HBM-bound ones by the stream ratio,
    weighted with their measured shares of the step -- and reported as `value_normalised` beside the raw value."""
    from img2img_turbo_amd import _capi as K
    import ctypes as C
    st = torch.cuda.current_stream(dev).cuda_stream
    ev = lambda: torch.cuda.Event(enable_timing=True)
    out = {}
    # 1. matrix pipe
    g = torch.Generator().manual_seed(7)
    operands = torch.randn(32768, generator=g).to(torch.bfloat16).to(dev)          # 64 KiB
    sink = torch.zeros(65536, dtype=torch.float32, device=dev)
    iters = 60000
    run = lambda: lib.check(lib.lib.i2i_calib_mfma(K.BF16, iters, C.c_void_p(operands.data_ptr()), C.c_void_p(sink.data_ptr()), C.c_void_p(st)))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(12):                       # ~0.2 s back to back: the power-managed steady state, not the first cold launch
        a, b = ev(), ev()
        a.record(); run(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    flops = 1024.0 * iters * 8 * 32768
    out["mfma_tflops"] = round(flops / (statistics.median(ts[4:]) * 1e-3) / 1e12, 1)
    # 2. HBM stream
    n = 256 << 20
    src = torch.empty(n, dtype=torch.uint8, device=dev).random_(0, 255)
    dst = torch.empty(n, dtype=torch.uint8, device=dev)
    cp = lambda: lib.check(lib.lib.i2i_calib_stream(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), C.c_size_t(n), C.c_void_p(st)))
    for _ in range(2):
        cp()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        a, b = ev(), ev()
        a.record(); cp(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    out["hbm_tbytes_per_s"] = round(2.0 * n / (statistics.median(ts) * 1e-3) / 1e12, 3)
    del src, dst
    # 3. launch path: 1000 empty kernels as one hipGraph
    prog = K.Program()
    for _ in range(1000):
        prog.add(K.OP_NOP, K.BF16, K.NopParams())
    prog.freeze()
    gr = lib.graph_create(prog)
    for _ in range(2):
        lib.graph_launch(gr, st)
    torch.cuda.synchronize()
    ts = []
    for _ in range(8):
        t0 = time.perf_counter()
        lib.graph_launch(gr, st)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6 / 1000)
    lib.graph_destroy(gr)
    out["graph_node_us"] = round(statistics.median(ts), 3)
    # 4. what the box says about itself
    m = _smi_json("metric", "-g", "0", "--clock", "--power")
    if m:
        ck = m.get("clock", {})
        val = lambda k: (ck.get(k, {}).get("clk", {}) or {}).get("value")
        out["clocks_mhz"] = {"mem": val("mem_0"), "fabric": val("fclk_0"), "soc": val("socclk_0"), "gfx_max": (ck.get("gfx_0", {}).get("max_clk", {}) or {}).get("value")}
    p_ = _smi_json("partition", "-g", "0", "--current")
    if p_:
        pp = p_.get("partition", p_)
        out["partition"] = {k: pp.get(k) for k in ("accelerator_type", "compute_partition", "memory", "memory_partition", "accelerator_profile_index") if k in pp} or str(pp)[:200]
    try:
        with open("/proc/cpuinfo") as f:
            out["host_cpu"] = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except Exception:
        pass
    out["host_cpus_allowed"] = len(os.sched_getaffinity(0))
    out["gpu_local_cpulist"] = _gpu_local_cpus(torch.device(dev).index or 0)[1]
    out["reference"] = dict(CALIB_REF)
    if all(CALIB_REF.get(k) for k in ("mfma_tflops", "hbm_tbytes_per_s")) and breakdown and ms_per_step:
        # time this step would take on the reference box: each family's share scaled by the ratio of the load that bounds it
        tot = sum(v["ms"] for v in breakdown.values())
        rm, rh = out["mfma_tflops"] / CALIB_REF["mfma_tflops"], out["hbm_tbytes_per_s"] / CALIB_REF["hbm_tbytes_per_s"]
        scaled = sum(v["ms"] * (rm if v.get("tflops") else rh) for v in breakdown.values())
        out["step_scale_to_reference"] = round(scaled / tot, 4)
        out["ms_per_step_normalised"] = round(ms_per_step * scaled / tot, 3)
        if batch:
            out["value_normalised"] = round(batch / (ms_per_step * scaled / tot) * 1e3, 2)
    out["how"] = ("csrc/calib.hip: 1024 waves x 60000 x 8 MFMA 32x32x16 bf16 on random operands (median of 8 back-to-back launches after 4), 2 x 256 MiB copy "
                  "(median of 10), 1000-node empty-kernel hipGraph replay (median of 8); amd-smi / sysfs for the rest")
    return out


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="images per GPU per step (default 8; 4 for --model cyclegan, 16 for --stochastic)")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--dtype", default="bf16", choices=list(DTYPES))
    ap.add_argument("--arch", default="sd-turbo", choices=["sd-turbo", "tiny"])
    ap.add_argument("--model", default="pix2pix", choices=["pix2pix", "cyclegan"])
    ap.add_argument("--direction", default="a2b", choices=["a2b", "b2a"], help="CycleGAN direction (day_to_night = a2b)")
    ap.add_argument("--stochastic", action="store_true", help="sketch_to_image_stochastic path: TwinConv, noise interpolation, r-scaled LoRA/skips")
    ap.add_argument("--gamma", type=float, default=0.4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--no-f32", action="store_true", help="skip the exact-f32-mode replays (images_per_s_f32 / parity_max_abs_f32) of the N = 1 line")
    ap.add_argument("--no-modes", action="store_true", help="skip the side-by-side precision modes (mixed fp16-UNet / bf16-VAE, all-fp16) of the N = 1 bf16 line")
    ap.add_argument("--no-power", action="store_true", help="skip the amd-smi socket-power / shader-clock reading of the N = 1 line")
    ap.add_argument("--no-calib", action="store_true", help="skip the ~1 s box calibration (MFMA loop, HBM stream, empty-graph replay) of the N = 1 line")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--serial-gather", action="store_true", help="N > 1: gather straight from the plan's output buffer on the launch stream (no overlap with the next replay)")
    ap.add_argument("--per-op", default=None, help="write the per-launch timing table (label, ms, TF) to this file")
    ap.add_argument("--emulate", default=None, metavar="LIBI2I_TURBO_EMU_SO",
                    help="TEST HOOK (tests/test_host_logic.py): run the whole bench flow -- rendezvous, shard, replay, double-buffered gather, "
                         "max-over-ranks timing, the JSON line -- on the CPU wave emulator (tiny architecture, 64x64, gloo). Says nothing about speed.")
    ap.add_argument("--activation-offset", type=float, default=0.0,
                    help="add this constant to every VAE ResnetBlock2D conv2 bias: the residual streams then sit on DC offsets of tens to hundreds "
                         "(what real SD-VAE activations do; the 1/sqrt(fan_in) synthetic weights keep everything zero-mean), GroupNorm groups trip the "
                         "cancellation test and take the second, shifted pass (csrc/norm.hip gn_refine_group) -- prices that path in the step")
    a = ap.parse_args()
    global PER_OP_PATH
    PER_OP_PATH = a.per_op
    if a.batch is None:
        a.batch = 4 if a.model == "cyclegan" else (16 if a.stochastic else 8)
    assert not (a.stochastic and a.model == "cyclegan")

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher (one rank per GPU over RCCL; rank 0 prints the line)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)

    from img2img_turbo_amd import dp
    from img2img_turbo_amd.arch import SD_TURBO_UNET, SD_TURBO_VAE, TINY_UNET, TINY_VAE
    from img2img_turbo_amd.cyclegan_turbo import CycleGAN_Turbo
    from img2img_turbo_amd.pix2pix_turbo import Pix2Pix_Turbo
    from img2img_turbo_amd.synth import make_cyclegan_weights, make_pix2pix_weights

    emu = a.emulate is not None
    rank, world, local = dp.init_from_env("gloo" if emu else None)
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (run `python bench.py --gpus N`, which spawns the ranks, or "
                         "torch.distributed.run --nproc-per-node N)" % (a.gpus, world))
    lib_kw = {}
    if emu:
        from img2img_turbo_amd import _capi
        dev, a.arch, a.size, a.dtype = "cpu", "tiny", 64, "f32"
        a.no_cpu_baseline = a.no_latency = a.no_f32 = True
        lib_kw = {"lib": _capi.Library(a.emulate)}
        torch.cuda.synchronize = lambda *args, **kw: None       # (no streams on the CPU: every call below is synchronous)
    else:
        assert torch.cuda.is_available(), "bench.py needs a GPU"
        torch.cuda.set_device(local)
        dev = "cuda:%d" % local
        # host threads next to the GPU: the launch path (hipGraphLaunch enqueues every node from the host) should not cross sockets
        cpus, _ = _gpu_local_cpus(local)
        if cpus and os.environ.get("I2I_BENCH_PIN", "1") != "0":
            try:
                os.sched_setaffinity(0, cpus & os.sched_getaffinity(0) or os.sched_getaffinity(0))
            except OSError:
                pass
    dtype = DTYPES[a.dtype]
    ua, va = (SD_TURBO_UNET, SD_TURBO_VAE) if a.arch == "sd-turbo" else (TINY_UNET, TINY_VAE)
    B = a.batch
    total = B * world

    # random init of the exact architecture (no checkpoints offline); every rank builds the same weights, its own images.
    # N ranks build and pack ~950 M synthetic parameters on the host at the same time: each gets its share of the cores
    # (unconstrained they oversubscribe each other), and the set-up time is reported per rank next to the timed region.
    t_setup = time.perf_counter()
    phase, t_phase = {}, [time.perf_counter()]      # wall seconds of every phase of this process, reported as `phase_s` (N = 1)

    def mark(name):
        now = time.perf_counter()
        phase[name] = round(phase.get(name, 0.0) + now - t_phase[0], 1)
        t_phase[0] = now
    if world > 1:
        torch.set_num_threads(max(1, (os.cpu_count() or world) // world))
    elif not emu:
        # torch sizes its intra-op pool to the HOST's cores (256) before this process confines itself to the GPU's NUMA node (128): the
        # over-subscribed pool made the host side of packing 3-5x slower (round 6: 68 s per model; the whole default line 330 s).  The
        # packer's ops are short memory-bound copies: a small fixed pool is the fast setting.
        torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
    # (N > 1: rank 0 synthesises, the other ranks map the same /dev/shm safetensors file -- dp.shared_weights -- instead of 8 ranks
    # building ~950 M parameters each at the same time)
    def build_weights():
        if a.model == "cyclegan":
            w_ = make_cyclegan_weights(ua, va, seed=1234 + 3)            # r_unet = 128, r_vae = 4 (training_utils.py:140-141)
        else:
            w_ = make_pix2pix_weights(ua, va, seed=1234 + (4 if a.stochastic else 2), sketch=a.stochastic)
        add_activation_offset(w_, a.activation_offset)
        return w_
    weights = dp.shared_weights(build_weights, rank, world, tag="bench")
    t_weights = time.perf_counter() - t_setup
    mark("weights_synthesis")
    if a.model == "cyclegan":
        model = CycleGAN_Turbo(weights=weights, device=dev, dtype=dtype, **lib_kw)
        kind, cfg = "photo", 3
    else:
        model = Pix2Pix_Turbo(weights=weights, device=dev, dtype=dtype, **lib_kw)
        kind, cfg = ("sketch", 4) if a.stochastic else ("canny", 2)
    x, cap, eps, noise = synth_inputs(kind, B, a.size, ua.cross_attention_dim, va.latent_channels, 1234 + cfg + 1000 * rank)
    if a.model == "cyclegan":
        plan = model.get_plan(B, a.size, a.size, direction=a.direction)
    else:
        plan = model.get_plan(B, a.size, a.size, stochastic=a.stochastic, r=a.gamma)
    model.stage(plan, x.to(dev), cap.to(dev), eps.to(dev), noise.to(dev) if a.stochastic else None)
    # double-buffered: the RCCL gather of step i runs on a side stream beside the replay of step i+1 (dp.OutputGather)
    gather = dp.OutputGather(plan.out, total, overlap=not a.serial_gather) if (world > 1 and not a.no_gather) else None
    torch.cuda.synchronize()
    setup_s = dp.all_ranks(time.perf_counter() - t_setup, dev)
    mark("pack_upload_plan")

    def step():
        plan.replay()
        if gather is not None:
            gather()     # stages plan.out and gathers it behind an event; --serial-gather: straight from plan.out, the stream waits

    for _ in range(a.warmup):
        step()
    dp.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    if gather is not None:
        gather.wait()        # the last step's images have arrived on rank 0 inside the timed region
    torch.cuda.synchronize()
    dp.barrier()
    elapsed = dp.max_over_ranks(time.perf_counter() - t0, dev)
    ms_per_step = elapsed / a.steps * 1e3
    value = total * a.steps / elapsed
    ms_compute = None
    if world > 1:
        # attribution of a scaling shortfall (outside the timed region): the same steps without the gather, per rank
        dp.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            plan.replay()
        torch.cuda.synchronize()
        ms_compute = [round(v / a.steps * 1e3, 3) for v in dp.all_ranks(time.perf_counter() - t1, dev)]

    mark("warmup_and_timed_region")
    if rank != 0:
        return
    name = {"pix2pix": "pix2pix-turbo sketch_to_image_stochastic gamma=%g" % a.gamma if a.stochastic else "pix2pix-turbo edge_to_image",
            "cyclegan": "CycleGAN-Turbo day_to_night (%s)" % a.direction}[a.model]
    headline = a.size == 512 and a.arch == "sd-turbo"
    rec = {"metric": "512x512 images/sec (whole node), %s forward" % name, "value": round(value, 3),
           "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": round(value / 9.09, 3) if headline else None,
           "dtype": a.dtype, "data": "synthetic (random-init weights of the SD-Turbo architecture + LoRA, seeded %s inputs)" % kind,
           "config": {"workload": "%s %s bs=%d/GPU %dx%d, hipGraph replay" % (name, a.dtype, B, a.size, a.size),
                      "global_batch": total, "parallelism": "dp%d (batch shards, replicated weights, RCCL gather of outputs)" % world,
                      "arch": a.arch, "kernel_library": os.path.basename(model.lib.path)},
           "baseline_note": "vs_baseline = value / 9.09 img/s (0.11 s per 512x512 image on A100, reference README.md:17)"}
    if a.activation_offset:
        rec["config"]["activation_offset"] = a.activation_offset
    if world > 1:
        rec["ms_compute_per_rank"] = ms_compute                     # replay only (no gather), measured after the timed region
        rec["ms_gather"] = round(ms_per_step - max(ms_compute), 3)   # what the RCCL gather adds to the slowest rank's step
        rec["gather"] = "serial (launch stream waits)" if a.serial_gather else "double-buffered, overlapped with the next replay (side stream)"
        rec["setup_s_per_rank"] = [round(v, 1) for v in setup_s]      # weights (built once on rank 0, mapped by the others) + packed + uploaded + plan
        rec["weights_s_rank0"] = round(t_weights, 1)
    falg = F_ALG.get(a.size)
    if falg and a.arch == "sd-turbo":
        rec["e2e_mfma_frac"] = round(value / world * falg / 1e12 / PEAK_TF[a.dtype], 4)
    if emu:
        rec["data"] = "EMULATED on the CPU (test hook): " + rec["data"]
    if world == 1 and not emu:
        roof, breakdown, tot = kernel_roofline(plan, a.dtype)
        rec["roofline"] = roof
        rec["kernel_breakdown_ms"] = breakdown
        rec["sum_kernel_ms"] = round(tot, 3)
        mark("per_op_event_pass")
        if not a.no_calib:
            rec["calib"] = calibrate(model.lib, dev, breakdown, ms_per_step, B)
            if rec["calib"].get("value_normalised"):
                rec["value_normalised"] = rec["calib"]["value_normalised"]
        mark("calib")
        pc = None if a.no_power else power_and_clock(plan)
        mark("power_poll")
        if pc:
            rec["power"] = pc
            if roof and roof.get("frac") and pc["gfx_mhz"] > 0:
                roof["frac_at_measured_clock"] = round(roof["frac"] * pc["gfx_max_mhz"] / pc["gfx_mhz"], 4)
        # throughput through the public forward() (boundary copies + output clone included), same batch
        fw = {"caption_enc" if a.model == "pix2pix" else "caption_emb": cap.to(dev), "eps": eps.to(dev)}
        if a.model == "cyclegan":
            fw["direction"] = a.direction
        elif a.stochastic:
            fw.update(deterministic=False, r=a.gamma, noise_map=noise.to(dev))
        xd = x.to(dev)
        out = model(xd, **fw)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(5):
            out = model(xd, **fw)
        torch.cuda.synchronize()
        rec["images_per_s_via_forward"] = round(5 * B / (time.perf_counter() - t), 2)
        if not a.no_latency:
            p1 = (model.get_plan(1, a.size, a.size, direction=a.direction) if a.model == "cyclegan" else
                  model.get_plan(1, a.size, a.size, stochastic=a.stochastic, r=a.gamma))
            model.stage(p1, xd[:1], cap.to(dev), eps[:1].to(dev), noise[:1].to(dev) if a.stochastic else None)
            for _ in range(3):
                p1.replay()
            torch.cuda.synchronize()
            lat = []
            for _ in range(30):
                t = time.perf_counter()
                p1.replay()
                torch.cuda.synchronize()
                lat.append((time.perf_counter() - t) * 1e3)
            rec["latency_bs1_ms_p50"] = round(statistics.median(lat), 3)
            rec["images_per_s_bs1"] = round(1e3 / statistics.median(lat), 2)
            # BASELINE's metric names bs = 1 / 8 / 32: p50 of one synchronous batch (submit -> all images of the batch done)
            # and the per-image share of it; deterministic pix2pix / cyclegan plans only differ in B
            for lb in (8, 32):
                if a.size != 512 or a.arch != "sd-turbo":
                    break
                if lb == B:
                    pb = plan
                else:
                    pb = (model.get_plan(lb, a.size, a.size, direction=a.direction) if a.model == "cyclegan" else
                          model.get_plan(lb, a.size, a.size, stochastic=a.stochastic, r=a.gamma))
                    xb, _, eb, nb = synth_inputs(kind, lb, a.size, ua.cross_attention_dim, va.latent_channels, 1234 + cfg)
                    model.stage(pb, xb.to(dev), cap.to(dev), eb.to(dev), nb.to(dev) if a.stochastic else None)
                for _ in range(2):
                    pb.replay()
                torch.cuda.synchronize()
                lat = []
                for _ in range(15):
                    t = time.perf_counter()
                    pb.replay()
                    torch.cuda.synchronize()
                    lat.append((time.perf_counter() - t) * 1e3)
                rec["latency_bs%d_ms_p50" % lb] = round(statistics.median(lat), 3)
                rec["latency_bs%d_ms_per_image_p50" % lb] = round(statistics.median(lat) / lb, 3)
        mark("forward_api_and_latency_plans")
        out32 = None
        mode_out = {}          # image 0 of the batch in the other precision modes (parity against the CPU oracle below)
        if a.dtype != "f32" and not a.no_f32 and a.arch == "sd-turbo":
            # north_star's 1e-3 bound is met by the exact-f32 MFMA mode (v_mfma_f32_16x16x4_f32 = an fmaf chain): its throughput and
            # its parity ride in the same line, a few replays of the same batch (roofline against the 157.3 TFLOP/s f32 peak)
            m32 = type(model)(weights=weights, device=dev, dtype=torch.float32)
            p32 = (m32.get_plan(B, a.size, a.size, direction=a.direction) if a.model == "cyclegan" else
                   m32.get_plan(B, a.size, a.size, stochastic=a.stochastic, r=a.gamma))
            m32.stage(p32, xd, cap.to(dev), eps.to(dev), noise.to(dev) if a.stochastic else None)
            for _ in range(2):
                p32.replay()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(3):
                p32.replay()
            torch.cuda.synchronize()
            dt32 = (time.perf_counter() - t) / 3
            out32 = p32.out[:1].float().cpu()
            rec["images_per_s_f32"] = round(B / dt32, 2)
            rec["ms_per_step_f32"] = round(dt32 * 1e3, 2)
            if falg:
                rec["e2e_mfma_frac_f32"] = round(B / dt32 * falg / 1e12 / PEAK_TF["f32"], 4)
            rec["f32_note"] = ("the same batch through the exact-f32 MFMA mode (fp32 activations and weights, 157.3 TFLOP/s peak): 3 hipGraph "
                               "replays after 2 warm-ups; parity_max_abs_f32 = its image 0 vs the CPU fp32 oracle")
            if not a.no_latency and B != 1:
                # the within-1e-3 answer to north_star's bs = 1 target (0.11 s per image on A100): p50 of one image in the exact-f32 mode
                p32_1 = (m32.get_plan(1, a.size, a.size, direction=a.direction) if a.model == "cyclegan" else
                         m32.get_plan(1, a.size, a.size, stochastic=a.stochastic, r=a.gamma))
                m32.stage(p32_1, xd[:1], cap.to(dev), eps[:1].to(dev), noise[:1].to(dev) if a.stochastic else None)
                for _ in range(2):
                    p32_1.replay()
                torch.cuda.synchronize()
                lat = []
                for _ in range(7):
                    t = time.perf_counter()
                    p32_1.replay()
                    torch.cuda.synchronize()
                    lat.append((time.perf_counter() - t) * 1e3)
                rec["latency_bs1_ms_p50_f32"] = round(statistics.median(lat), 3)
            m32.release_plans()
            del m32, p32
            torch.cuda.empty_cache()
        mark("f32_mode")
        if a.dtype == "bf16" and not a.no_modes and a.arch == "sd-turbo":
            # the 16-bit precision modes side by side on the SAME batch: throughput + parity of image 0 (filled in below)
            rec["precision_modes"] = {"bf16": {"images_per_s": round(value, 2)}}
            for mname, kw in (("mixed_unet_f16_vae_bf16", dict(dtype=torch.bfloat16, unet_dtype=torch.float16)), ("f16", dict(dtype=torch.float16))):
                mm = type(model)(weights=weights, device=dev, **kw)
                pm = (mm.get_plan(B, a.size, a.size, direction=a.direction) if a.model == "cyclegan" else
                      mm.get_plan(B, a.size, a.size, stochastic=a.stochastic, r=a.gamma))
                mm.stage(pm, xd, cap.to(dev), eps.to(dev), noise.to(dev) if a.stochastic else None)
                for _ in range(3):
                    pm.replay()
                torch.cuda.synchronize()
                t = time.perf_counter()
                for _ in range(8):
                    pm.replay()
                torch.cuda.synchronize()
                rec["precision_modes"][mname] = {"images_per_s": round(8 * B / (time.perf_counter() - t), 2)}
                mode_out[mname] = pm.out[:1].float().cpu()
                mm.release_plans()
                del mm, pm
                torch.cuda.empty_cache()
            if "images_per_s_f32" in rec:
                rec["precision_modes"]["f32_exact"] = {"images_per_s": rec["images_per_s_f32"]}
        mark("precision_modes")
        if not a.no_cpu_baseline:
            cb, ref = cpu_baseline(a, weights, x, cap, eps, noise)
            mark("cpu_baseline")
            rec["cpu_baseline"] = cb
            d = (out[:1].float().cpu() - ref).abs()
            mse = float((d ** 2).mean())
            rec["parity_max_abs"] = round(float(d.max()), 5)
            rec["parity_mean_abs"] = round(float(d.mean()), 6)
            rec["parity_psnr_db"] = round(10 * torch.log10(torch.tensor(4.0 / max(mse, 1e-20))).item(), 2)
            rec["parity_note"] = "GPU %s image 0 of the benchmarked batch vs the CPU fp32 oracle on the same inputs, outputs in [-1,1]" % a.dtype
            if out32 is not None:
                rec["parity_max_abs_f32"] = round(float((out32 - ref).abs().max()), 7)
            if "precision_modes" in rec:
                pm_ = rec["precision_modes"]
                pm_["bf16"].update(parity_max_abs=rec["parity_max_abs"], parity_psnr_db=rec["parity_psnr_db"])
                for mname, o in mode_out.items():
                    dd = (o - ref).abs()
                    pm_[mname].update(parity_max_abs=round(float(dd.max()), 5),
                                      parity_psnr_db=round(10 * torch.log10(torch.tensor(4.0 / max(float((dd ** 2).mean()), 1e-20))).item(), 2))
                if out32 is not None and "f32_exact" in pm_:
                    pm_["f32_exact"].update(parity_max_abs=rec["parity_max_abs_f32"])
                pm_["note"] = ("image 0 of the benchmarked batch vs the CPU fp32 oracle; mixed = UNet in fp16 (the network whose error the 1-step scheduler "
                               "multiplies by 14.6) with the VAE in bf16 (whose real activations overflow fp16)")
    if world == 1:
        rec["phase_s"] = dict(phase, total=round(sum(phase.values()), 1))
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
