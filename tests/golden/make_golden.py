"""Regenerate tests/golden/tiny_pix2pix.pt: the oracle's output (sub-sampled) for one seeded tiny-architecture
forward.  The reference itself cannot run here (no diffusers/peft/weights), so this fixture pins the ORACLE
against silent drift; it is not a reference-generated vector (parity unpinned, see oracle/__init__.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import TINY_UNET, TINY_VAE  # noqa: E402
from oracle.pipeline import pix2pix_forward  # noqa: E402
from oracle.synth import make_inputs, make_pix2pix_weights  # noqa: E402

seed, input_seed = 11, 5
mw = make_pix2pix_weights(TINY_UNET, TINY_VAE, seed=seed)
x, cap, eps, _ = make_inputs("canny", 1, 64, 64, TINY_UNET.cross_attention_dim, seed=input_seed)
out = pix2pix_forward(mw, x, cap, eps)
torch.save({"seed": seed, "input_seed": input_seed, "out_sub": out[0, :, ::4, ::4].clone(), "sum": out.double().sum().item()},
           os.path.join(os.path.dirname(os.path.abspath(__file__)), "tiny_pix2pix.pt"))
print("wrote golden", out.shape)
