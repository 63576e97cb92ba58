"""Precision floor of bf16 / fp16 at every BASELINE.json configuration (test infrastructure; CPU only).

    python tests/golden/make_dtype_floors.py [--only cfg3_a2b,...]      # writes tests/golden/dtype_floors.json

For each configuration the GPU suite checks (tests/test_e2e_gpu.py) this runs the CPU oracle twice on the SAME seeded
weights and inputs: plain fp32, and the emulated-precision mode (oracle/nn.py `quantized`: every weight and every layer
output rounded to the dtype, fp32 accumulation and statistics, fp32 latent path).  The distance between the two -- RMS
and max-abs of the output image, outputs in [-1, 1] -- is what ANY kernel set computing this network in that dtype pays
on that input.  The GPU tests gate the HIP path at 1.25 x the recorded RMS and 1.5 x the recorded max-abs (the max of
~1.5 M samples of a different error realisation scatters more than its RMS) instead of one global tolerance.

The weights are synthetic and seeded (oracle/synth.py), so the floors are constants of (configuration, seed): they are
computed once here, on the CPU, and committed; the generating command is this file.  Nothing in the product reads it.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import SD_TURBO_UNET, SD_TURBO_VAE  # noqa: E402
from oracle.nn import quantized  # noqa: E402
from oracle.pipeline import cyclegan_forward, pix2pix_forward  # noqa: E402
from oracle.synth import make_cyclegan_weights, make_inputs, make_pix2pix_weights  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dtype_floors.json")
CD = SD_TURBO_UNET.cross_attention_dim


# name -> (dtype, builder returning a closure fwd() -> image [n,3,H,W]); seeds / image subsets exactly as in the GPU tests
def _cfg2():
    mw = make_pix2pix_weights(SD_TURBO_UNET, SD_TURBO_VAE, seed=1234 + 2)
    x, cap, eps, _ = make_inputs("canny", 8, 512, 512, CD, seed=2)
    x[5], eps[5] = x[0], eps[0]
    return lambda: pix2pix_forward(mw, x[[0, 7]], cap, eps[[0, 7]], return_intermediates=True)


def _cfg2_all():
    mw = make_pix2pix_weights(SD_TURBO_UNET, SD_TURBO_VAE, seed=1234 + 2)
    x, cap, eps, _ = make_inputs("canny", 8, 512, 512, CD, seed=2)
    x[5], eps[5] = x[0], eps[0]
    return lambda: pix2pix_forward(mw, x, cap, eps, return_intermediates=True)


def _cfg3(direction):
    def build():
        mw = make_cyclegan_weights(SD_TURBO_UNET, SD_TURBO_VAE)
        x, cap, eps, _ = make_inputs("photo", 4, 512, 512, CD, seed=3)
        return lambda: cyclegan_forward(mw, x[[0, 3]], cap, eps[[0, 3]], direction=direction, return_intermediates=True)
    return build


def _cfg4(r, imgs):
    def build():
        mw = make_pix2pix_weights(SD_TURBO_UNET, SD_TURBO_VAE, seed=1234 + 4, sketch=True)
        x, cap, eps, nm = make_inputs("sketch", 16, 512, 512, CD, seed=4)
        return lambda: pix2pix_forward(mw, x[imgs], cap, eps[imgs], deterministic=False, r=r, noise_map=nm[imgs], return_intermediates=True)
    return build


def _cfg5():
    mw = make_pix2pix_weights(SD_TURBO_UNET, SD_TURBO_VAE, seed=1234 + 5)
    x, cap, eps, _ = make_inputs("canny", 2, 1024, 1024, CD, seed=5)
    return lambda: pix2pix_forward(mw, x[1:2], cap, eps[1:2], return_intermediates=True)


def _skipfold():
    mw = make_pix2pix_weights(SD_TURBO_UNET, SD_TURBO_VAE, seed=1234 + 9, sketch=True)
    x, cap, eps, nm = make_inputs("sketch", 8, 512, 512, CD, seed=9)
    return lambda: pix2pix_forward(mw, x[:1], cap, eps[:1], deterministic=False, r=0.6, noise_map=nm[:1], return_intermediates=True)


def _full512():
    mw = make_pix2pix_weights(SD_TURBO_UNET, SD_TURBO_VAE, seed=1234 + 1)
    x, cap, eps, _ = make_inputs("canny", 1, 512, 512, CD, seed=1)
    return lambda: pix2pix_forward(mw, x, cap, eps, return_intermediates=True)


def _floor512():
    mw = make_pix2pix_weights(SD_TURBO_UNET, SD_TURBO_VAE, seed=3)
    x, cap, eps, _ = make_inputs("canny", 1, 512, 512, CD, seed=5)
    return lambda: pix2pix_forward(mw, x, cap, eps, return_intermediates=True)


CONFIGS = {
    "cfg2_pix2pix_bf16_bs8_512": (torch.bfloat16, _cfg2),
    # all eight images of the benchmarked batch (round 6), and the same batch in the per-network precision mode (fp16 UNet, bf16 VAE)
    "cfg2_pix2pix_bf16_bs8_512_all8": (torch.bfloat16, _cfg2_all),
    "cfg2_pix2pix_mixed_unet_f16_vae_bf16_bs8_512": ((torch.bfloat16, torch.float16), _cfg2),
    "cfg3_cyclegan_a2b_bf16_bs4_512": (torch.bfloat16, _cfg3("a2b")),
    "cfg3_cyclegan_b2a_bf16_bs4_512": (torch.bfloat16, _cfg3("b2a")),
    "cfg4_stochastic_r0.4_bf16_bs16_512": (torch.bfloat16, _cfg4(0.4, [0, 15])),
    "cfg4_stochastic_r0.8_bf16_image0": (torch.bfloat16, _cfg4(0.8, [0])),
    "cfg5_pix2pix_f16_1024": (torch.float16, _cfg5),
    "skipfold_r0.6_bf16_image0": (torch.bfloat16, _skipfold),
    "full_sd_turbo_512_bf16": (torch.bfloat16, _full512),
    "floor_seed3_512_bf16": (torch.bfloat16, _floor512),
    "floor_seed3_512_f16": (torch.float16, _floor512),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    rec = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for name, (dtype, build) in CONFIGS.items():
        if a.only and name not in a.only.split(","):
            continue
        t0 = time.time()
        fwd = build()
        ref, ri = fwd()
        with (quantized(*dtype) if isinstance(dtype, tuple) else quantized(dtype)):
            emu, ei = fwd()
        d = (emu - ref)
        rec[name] = {"dtype": str(dtype).replace("torch.", ""), "images": int(ref.shape[0]), "size": int(ref.shape[-1]),
                     "rms": float(d.pow(2).mean().sqrt()), "max_abs": float(d.abs().max()), "mean_abs": float(d.abs().mean()),
                     # the stages the planned program exposes (tests: test_error_is_at_the_floor_of_the_dtype)
                     "stages": {k: {"rms": float((ei[k] - ri[k]).pow(2).mean().sqrt()), "max_abs": float((ei[k] - ri[k]).abs().max())}
                                for k in ("moments", "eps")}}
        print(name, rec[name], "%.0f s" % (time.time() - t0), flush=True)
        with open(OUT, "w") as f:
            json.dump(rec, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
