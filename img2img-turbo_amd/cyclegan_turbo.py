"""CycleGAN_Turbo with the reference's API surface (src/cyclegan_turbo.py:109-254), MI355X-native inside.

``forward(x_t, direction=None, caption=None, caption_emb=None)`` follows :241-254; the generator itself is
``forward_with_networks`` (:199-207): VAE (``vae`` for a2b, ``vae_b2a`` for b2a, :21-27/:36-45) -> one shared
UNet -> per-sample scheduler step (same closed form for every sample; the reference's ``timesteps[i]``
indexing only works at B=1, here a length-1 ``timesteps`` broadcasts) -> VAE decode with skips -> clamp.
"""
from typing import Optional

import torch

from .pix2pix_turbo import TurboGeneratorBase
from .weights import GeneratorWeights, from_cyclegan_checkpoint, load_checkpoint_file, load_sd_turbo_base

PRETRAINED = {  # src/cyclegan_turbo.py:126-149: name -> (checkpoint file, caption, direction)
    "day_to_night": ("day2night.pkl", "driving in the night", "a2b"),
    "night_to_day": ("night2day.pkl", "driving in the day", "b2a"),
    "clear_to_rainy": ("clear2rainy.pkl", "driving in heavy rain", "a2b"),
    "rainy_to_clear": ("rainy2clear.pkl", "driving in the day", "b2a"),
}


class _XformersShim:
    """``model.unet.enable_xformers_memory_efficient_attention()`` (src/inference_unpaired.py:36) is accepted:
    the fused flash-style attention kernel is always on."""

    def enable_xformers_memory_efficient_attention(self):
        return None


class CycleGAN_Turbo(TurboGeneratorBase):
    def __init__(self, pretrained_name=None, pretrained_path=None, ckpt_folder="checkpoints", lora_rank_unet=8, lora_rank_vae=4,
                 *, weights: Optional[GeneratorWeights] = None, base_dir=None, caption=None, direction=None, **kw):
        self.caption, self.direction = caption, direction
        if weights is None:
            import os
            if base_dir is None:
                raise ValueError("give weights=GeneratorWeights(...) or base_dir=<local sd-turbo snapshot> (no network here)")
            if pretrained_name in PRETRAINED:
                fn, cap, dr = PRETRAINED[pretrained_name]
                pretrained_path = os.path.join(ckpt_folder, fn)
                caption, direction = cap, dr
            if pretrained_path is None:
                raise ValueError("pretrained_name or pretrained_path required")
            unet, _ = load_sd_turbo_base(base_dir)
            weights = from_cyclegan_checkpoint(unet, load_checkpoint_file(pretrained_path))
        super().__init__(weights, **kw)
        self.caption, self.direction = caption, direction
        self.unet = _XformersShim()

    @torch.no_grad()
    def forward_u8(self, images_u8, *args, **kw):
        """uint8 HWC in/out on the device: ``Normalize([0.5],[0.5])(to_tensor(img))`` (src/inference_unpaired.py:47) and
        ``ToPILImage()(out*0.5+0.5)`` (:53) run inside the boundary kernels."""
        assert images_u8.dtype == torch.uint8 and images_u8.dim() == 4 and images_u8.shape[-1] == 3
        return self.forward(images_u8, *args, _u8_io=(2.0, -1.0), **kw)

    @torch.no_grad()
    def forward(self, x_t, direction=None, caption=None, caption_emb=None, *, eps=None, _u8_io=None):
        if direction is None:
            assert self.direction is not None
            direction = self.direction
        assert direction in ("a2b", "b2a")
        if caption is None and caption_emb is None:
            assert self.caption is not None
            caption = self.caption
        caption_enc = caption_emb if caption_emb is not None else self.encode_prompt(caption)
        if _u8_io is not None:
            B, H, W, _ = x_t.shape
        else:
            B, _, H, W = x_t.shape
        lat = self.weights.vae_arch.latent_channels
        if eps is None:
            eps = torch.randn(B, lat, H // 8, W // 8, device=self.device_, dtype=torch.float32)
            torch.randn(B, lat, H // 8, W // 8, device=self.device_, dtype=torch.float32)
        ctx_batch = caption_enc.shape[0] if caption_enc.dim() == 3 else 1
        plan = self.get_plan(B, H, W, direction=direction, ctx_batch=ctx_batch, u8_io=_u8_io)
        out = self._execute(plan, x_t, caption_enc, eps)
        return out if _u8_io is not None else out.to(x_t.dtype)
