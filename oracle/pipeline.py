"""End-to-end reference pipelines (test infrastructure; see oracle/__init__.py).

pix2pix_forward  <- src/pix2pix_turbo.py:186-219
cyclegan_forward <- src/cyclegan_turbo.py:199-207 (+ VAE_encode/VAE_decode :15-45)

Text conditioning (``caption_enc`` [B or 1, 77, cross_dim]) and both RNG draws
are INPUTS: the reference draws eps on the device generator
(``latent_dist.sample()`` then ``sched.step``), which no CPU oracle can
reproduce, so parity harnesses inject them (SURVEY.md section 7 hard part 7).
"""
from dataclasses import dataclass, field
from typing import Dict, Optional

import torch

from .arch import UNetArch, VAEArch
from .nn import Weights
from .sched import ddpm_step
from .unet import unet_forward
from .vae import decoder_forward, encoder_forward, posterior_sample


@dataclass
class ModelWeights:
    """Canonical weight bundle: diffusers/peft-keyed state dicts + LoRA scalings."""
    unet: Dict[str, torch.Tensor]
    vae: Dict[str, torch.Tensor]
    unet_arch: UNetArch
    vae_arch: VAEArch
    unet_scaling: Dict[str, float] = field(default_factory=dict)   # adapter -> lora_alpha / r
    vae_scaling: Dict[str, float] = field(default_factory=dict)
    vae_b2a: Optional[Dict[str, torch.Tensor]] = None              # CycleGAN second VAE

    def W_unet(self, r=1.0):
        return Weights(self.unet, {k: v * r for k, v in self.unet_scaling.items()})

    def W_vae(self, r=1.0, direction="a2b"):
        sd = self.vae if direction == "a2b" or self.vae_b2a is None else self.vae_b2a
        return Weights(sd, {k: v * r for k, v in self.vae_scaling.items()})


@torch.no_grad()
def pix2pix_forward(mw: ModelWeights, c_t, caption_enc, eps_enc, deterministic=True, r=1.0,
                    noise_map=None, eps_sched=None, return_intermediates=False):
    """Pix2Pix_Turbo.forward (src/pix2pix_turbo.py:186-219), fp32.

    deterministic: lines 197-203.  stochastic (r, noise_map): lines 204-218 --
    every LoRA scale x r (206-207), unet_input = enc*r + noise*(1-r) (210),
    TwinConv(r) (211), scheduler steps from unet_input (214), decoder gamma = r (217).
    """
    sf = mw.vae_arch.scaling_factor
    if deterministic:
        Wv, Wu, gamma, twin_r = mw.W_vae(), mw.W_unet(), 1.0, None
    else:
        Wv, Wu, gamma, twin_r = mw.W_vae(r), mw.W_unet(r), r, r
    moments, skips = encoder_forward(Wv, mw.vae_arch, c_t)
    z = posterior_sample(moments, eps_enc) * sf
    u = z if deterministic else z * r + noise_map * (1 - r)
    eps = unet_forward(Wu, mw.unet_arch, u, caption_enc, twin_r=twin_r)
    x0 = ddpm_step(eps, u, mw.unet_arch.timestep, eps_sched)
    out = decoder_forward(Wv, mw.vae_arch, x0 / sf, skips, gamma).clamp(-1, 1)
    if return_intermediates:
        return out, dict(moments=moments, skips=skips, z=z, unet_in=u, eps=eps, x0=x0)
    return out


@torch.no_grad()
def cyclegan_forward(mw: ModelWeights, x_t, caption_enc, eps_enc, direction="a2b", eps_sched=None,
                     return_intermediates=False):
    """CycleGAN_Turbo.forward_with_networks (src/cyclegan_turbo.py:199-207).

    The per-sample scheduler loop (:205) is the same closed form for every
    sample; a length-1 ``timesteps`` is broadcast (A.9 quirk 3: reference only
    works at B=1).
    """
    assert direction in ("a2b", "b2a")
    sf = mw.vae_arch.scaling_factor
    Wv, Wu = mw.W_vae(direction=direction), mw.W_unet()
    moments, skips = encoder_forward(Wv, mw.vae_arch, x_t)
    z = posterior_sample(moments, eps_enc) * sf
    eps = unet_forward(Wu, mw.unet_arch, z, caption_enc)
    x0 = ddpm_step(eps, z, mw.unet_arch.timestep, eps_sched)
    out = decoder_forward(Wv, mw.vae_arch, x0 / sf, skips, 1.0).clamp(-1, 1)
    if return_intermediates:
        return out, dict(moments=moments, skips=skips, z=z, eps=eps, x0=x0)
    return out
