"""Same-box, same-process A/B of the whole step with error bars: arms differ by environment variables (planner / launcher
hooks such as I2I_FUSE_SKIP, I2I_CROSS_KV_MERGED, I2I_GEMM_W32) and are timed INTERLEAVED.

    python benchmarks/ab.py --arms "I2I_GEMM_W32=1" "I2I_GEMM_W32=0" [--repeats 7 --steps 10 --batch 8]

Every arm gets its own model + plan + captured hipGraph (the hooks are read at plan / capture time), all arms share the
synthetic weights and inputs.  One repeat = `steps` graph replays of each arm in turn (order rotated per repeat so that
clock / thermal drift does not favour an arm); the report is mean +- sd of ms/step over the repeats and the paired
difference against the first arm.  A difference inside ~2 sd of the paired differences is noise: single-shot A/Bs across
gpurun boxes (which differ by 5-15 %) cannot resolve 1-2 % effects, this can.
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arms", nargs="+", required=True, help='each arm: "VAR=val[,VAR2=val2]" ("" / "-" = no override)')
    ap.add_argument("--repeats", type=int, default=7)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from img2img_turbo_amd.arch import SD_TURBO_UNET, SD_TURBO_VAE
    from img2img_turbo_amd.pix2pix_turbo import Pix2Pix_Turbo
    from img2img_turbo_amd.synth import make_pix2pix_weights

    dev, dtype = "cuda:0", bench.DTYPES[a.dtype]
    weights = make_pix2pix_weights(SD_TURBO_UNET, SD_TURBO_VAE, seed=1234 + 2)
    x, cap, eps, _ = bench.synth_inputs("canny", a.batch, a.size, SD_TURBO_UNET.cross_attention_dim, SD_TURBO_VAE.latent_channels, 1236)
    plans, keep = [], []
    touched = set()
    for arm in a.arms:
        env = dict(kv.split("=", 1) for kv in arm.split(",") if "=" in kv)
        for k in touched:
            os.environ.pop(k, None)
        os.environ.update(env)
        touched |= set(env)
        # (an arm may also name another build of the library: I2I_LIB=<path> -- csrc/build.py --tag ...)
        from img2img_turbo_amd import _capi
        lib = _capi.Library(env["I2I_LIB"]) if env.get("I2I_LIB") else None
        model = Pix2Pix_Turbo(weights=weights, device=dev, dtype=dtype, lib=lib)
        plan = model.get_plan(a.batch, a.size, a.size)
        model.stage(plan, x.to(dev), cap.to(dev), eps.to(dev), None)
        for _ in range(3):
            plan.replay()          # captures the graph under this arm's environment
        torch.cuda.synchronize()
        plans.append(plan)
        keep.append(model)
    for k in touched:
        os.environ.pop(k, None)
    ms = [[] for _ in plans]
    for r in range(a.repeats):
        order = list(range(len(plans)))
        order = order[r % len(order):] + order[:r % len(order)]
        for i in order:
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(a.steps):
                plans[i].replay()
            torch.cuda.synchronize()
            ms[i].append((time.perf_counter() - t) / a.steps * 1e3)
    rep = {"batch": a.batch, "size": a.size, "dtype": a.dtype, "repeats": a.repeats, "steps_per_repeat": a.steps, "arms": []}
    for i, arm in enumerate(a.arms):
        m, sd = statistics.mean(ms[i]), (statistics.stdev(ms[i]) if len(ms[i]) > 1 else 0.0)
        rec = {"arm": arm or "-", "ms_per_step_mean": round(m, 4), "ms_per_step_sd": round(sd, 4), "images_per_s": round(a.batch / m * 1e3, 2),
               "launches": len(plans[i].prog.ops)}
        if i:
            dif = [b - c for b, c in zip(ms[i], ms[0])]
            rec["paired_diff_ms_vs_first"] = round(statistics.mean(dif), 4)
            rec["paired_diff_sd"] = round(statistics.stdev(dif), 4) if len(dif) > 1 else 0.0
            rec["paired_diff_pct"] = round(100 * statistics.mean(dif) / statistics.mean(ms[0]), 3)
        rep["arms"].append(rec)
        print("%-40s %8.3f +- %.3f ms/step  %7.2f img/s  %s" % (rec["arm"], m, sd, rec["images_per_s"],
              ("diff vs first %+.3f +- %.3f ms (%+.2f %%)" % (rec["paired_diff_ms_vs_first"], rec["paired_diff_sd"], rec["paired_diff_pct"])) if i else ""), flush=True)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(rep, f, indent=1)


if __name__ == "__main__":
    main()
