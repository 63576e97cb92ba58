"""Generator weights: the canonical bundle and readers for the reference's on-disk formats.

Canonical form = diffusers/peft-keyed state dicts exactly as they sit in the reference's modules after
adapter injection (``X.base_layer.weight``, ``X.lora_A.<adapter>.weight``, ``X.lora_B.<adapter>.weight``)
plus the per-adapter LoRA scaling ``lora_alpha / r`` that peft keeps on the module.

Readers (all local paths; the reference downloads over HTTP, impossible and out of scope here):
  * ``from_pix2pix_checkpoint``  -- the ``.pkl`` dict written by Pix2Pix_Turbo.save_model
                                    (src/pix2pix_turbo.py:221-229, consumed at :66-78 / :100-113)
  * ``from_cyclegan_checkpoint`` -- the dict written at src/train_cyclegan_turbo.py:293-307
                                    (consumed by load_ckpt_from_state_dict, src/cyclegan_turbo.py:162-190)
  * ``load_sd_turbo_base``       -- the SD-Turbo ``unet/`` and ``vae/`` safetensors those overlay
"""
import os
from dataclasses import dataclass, field
from typing import Dict, Optional

import torch

from .arch import SD_TURBO_UNET, SD_TURBO_VAE, UNetArch, VAEArch


@dataclass
class GeneratorWeights:
    unet: Dict[str, torch.Tensor]
    vae: Dict[str, torch.Tensor]
    unet_arch: UNetArch = SD_TURBO_UNET
    vae_arch: VAEArch = SD_TURBO_VAE
    unet_scaling: Dict[str, float] = field(default_factory=dict)
    vae_scaling: Dict[str, float] = field(default_factory=dict)
    vae_b2a: Optional[Dict[str, torch.Tensor]] = None
    meta: Dict[str, object] = field(default_factory=dict)     # checkpoint fields carried through for save_model (ranks, target module lists)

    @property
    def is_twin_conv(self):
        return "conv_in.conv_in_pretrained.weight" in self.unet


def _wrap_adapted(sd):
    """Move ``X.weight/bias`` to ``X.base_layer.*`` for every module X that carries a LoRA adapter."""
    mods = set()
    for k in sd:
        i = k.find(".lora_A.")
        if i >= 0:
            mods.add(k[:i])
    for m in mods:
        for leaf in ("weight", "bias"):
            if f"{m}.{leaf}" in sd and f"{m}.base_layer.{leaf}" not in sd:
                sd[f"{m}.base_layer.{leaf}"] = sd.pop(f"{m}.{leaf}")
    return sd


def from_pix2pix_checkpoint(base_unet, base_vae, ckpt, unet_arch=SD_TURBO_UNET, vae_arch=SD_TURBO_VAE) -> GeneratorWeights:
    """Overlay a Pix2Pix_Turbo checkpoint on the pretrained state dicts (src/pix2pix_turbo.py:66-78).

    LoraConfig is built there without lora_alpha => peft default 8 => scaling 8 / rank."""
    unet, vae = dict(base_unet), dict(base_vae)
    for k, v in ckpt["state_dict_unet"].items():
        unet[k] = v
    for k, v in ckpt["state_dict_vae"].items():
        vae[k] = v
    if "conv_in.conv_in_pretrained.weight" in unet:      # sketch model: conv_in is a TwinConv (:100-101)
        unet.pop("conv_in.weight", None)
        unet.pop("conv_in.bias", None)
    _wrap_adapted(unet)
    _wrap_adapted(vae)
    meta = {k: ckpt.get(k) for k in ("unet_lora_target_modules", "vae_lora_target_modules", "rank_unet", "rank_vae")}
    return GeneratorWeights(unet, vae, unet_arch, vae_arch,
                            unet_scaling={"default": 8.0 / ckpt["rank_unet"]},
                            vae_scaling={"vae_skip": 8.0 / ckpt["rank_vae"]}, meta=meta)


def from_cyclegan_checkpoint(base_unet, ckpt, unet_arch=SD_TURBO_UNET, vae_arch=SD_TURBO_VAE) -> GeneratorWeights:
    """CycleGAN_Turbo.load_ckpt_from_state_dict (src/cyclegan_turbo.py:162-190).

    UNet: three adapters whose tensors are stored with the adapter name stripped (:169-180),
    lora_alpha = rank => scaling 1.  VAE: ``sd_vae_enc`` and ``sd_vae_dec`` are both complete
    wrapper state dicts over the SAME two VAE objects and are loaded in that order (:186-190), so
    the values of ``sd_vae_dec`` are the ones in effect; both VAEs come from it."""
    unet = dict(base_unet)
    for key, ad in (("sd_encoder", "default_encoder"), ("sd_decoder", "default_decoder"), ("sd_other", "default_others")):
        for k, v in ckpt[key].items():
            assert k.endswith(".weight") and ".lora_" in k, k
            unet[k[: -len(".weight")] + f".{ad}.weight"] = v
    _wrap_adapted(unet)
    vae, vae_b2a = {}, {}
    for k, v in ckpt["sd_vae_dec"].items():
        if k.startswith("vae_b2a."):
            vae_b2a[k[len("vae_b2a."):]] = v
        elif k.startswith("vae."):
            vae[k[len("vae."):]] = v
    return GeneratorWeights(unet, vae, unet_arch, vae_arch,
                            unet_scaling={"default_encoder": 1.0, "default_decoder": 1.0, "default_others": 1.0},
                            vae_scaling={"vae_skip": 8.0 / ckpt["rank_vae"]}, vae_b2a=vae_b2a)


def load_sd_turbo_base(root):
    """Read ``<root>/unet/diffusion_pytorch_model.safetensors`` and ``<root>/vae/...`` (a local snapshot of
    stabilityai/sd-turbo as ``from_pretrained(..., subfolder="unet"|"vae")`` lays it out, src/pix2pix_turbo.py:36,45;
    the ``.fp16.safetensors`` variant is accepted).  Everything is widened to fp32: the packer rounds once, after the merge."""
    from safetensors.torch import load_file
    out = []
    for sub in ("unet", "vae"):
        for fn in ("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.fp16.safetensors"):
            p = os.path.join(root, sub, fn)
            if os.path.exists(p):
                out.append({k: v.float() for k, v in load_file(p).items()})
                break
        else:
            raise FileNotFoundError(f"no safetensors under {os.path.join(root, sub)}")
    unet, vae = out
    # diffusers renames the legacy VAE attention keys on load
    ren = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}
    for k in list(vae):
        for old, new in ren.items():
            if f".attentions.0.{old}." in k:
                vae[k.replace(f".{old}.", f".{new}.")] = vae.pop(k)
    return unet, vae


def load_checkpoint_file(path):
    """torch.load of a reference ``.pkl`` onto the CPU (the reference omits map_location, A.8)."""
    return torch.load(path, map_location="cpu", weights_only=False)
