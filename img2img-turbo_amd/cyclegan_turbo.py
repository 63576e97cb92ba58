"""CycleGAN_Turbo with the reference's API surface (src/cyclegan_turbo.py:109-254), MI355X-native inside.

``forward(x_t, direction=None, caption=None, caption_emb=None)`` follows :241-254; the generator itself is
``forward_with_networks`` (:199-207): VAE (``vae`` for a2b, ``vae_b2a`` for b2a, :21-27/:36-45) -> one shared
UNet -> per-sample scheduler step (same closed form for every sample; the reference's ``timesteps[i]``
indexing only works at B=1, here a length-1 ``timesteps`` broadcasts) -> VAE decode with skips -> clamp.
"""
from typing import Optional

import torch

from .pix2pix_turbo import TurboGeneratorBase, _NetHandle
from .weights import GeneratorWeights, from_cyclegan_checkpoint, load_checkpoint_file, load_sd_turbo_base

PRETRAINED = {  # src/cyclegan_turbo.py:126-149: name -> (checkpoint file, caption, direction)
    "day_to_night": ("day2night.pkl", "driving in the night", "a2b"),
    "night_to_day": ("night2day.pkl", "driving in the day", "b2a"),
    "clear_to_rainy": ("clear2rainy.pkl", "driving in heavy rain", "a2b"),
    "rainy_to_clear": ("rainy2clear.pkl", "driving in the day", "b2a"),
}


class CycleGAN_Turbo(TurboGeneratorBase):
    def __init__(self, pretrained_name=None, pretrained_path=None, ckpt_folder="checkpoints", lora_rank_unet=8, lora_rank_vae=4,
                 *, weights: Optional[GeneratorWeights] = None, base_dir=None, caption=None, direction=None, **kw):
        self.caption, self.direction = caption, direction
        if weights is None:
            import os
            base_dir = base_dir or os.environ.get("I2I_SD_TURBO_DIR")
            if base_dir is None:
                raise ValueError("give weights=GeneratorWeights(...), base_dir=<local sd-turbo snapshot> or set I2I_SD_TURBO_DIR "
                                 "(no network here)")
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
        # the wrappers the reference builds at src/cyclegan_turbo.py:115-116 (each holds both VAEs, picks one per direction)
        self.vae_enc = _NetHandle("vae_enc", weights.vae, weights.vae_b2a if weights.vae_b2a is not None else weights.vae)
        self.vae_dec = _NetHandle("vae_dec", weights.vae, weights.vae_b2a if weights.vae_b2a is not None else weights.vae)
        self.vae_enc._model = self.vae_dec._model = self.unet._model = self

    @torch.no_grad()
    def forward_u8(self, images_u8, *args, resize=None, resize_back=False, image_prep=None, **kw):
        """uint8 HWC in/out on the device: ``Normalize([0.5],[0.5])(to_tensor(img))`` (src/inference_unpaired.py:47) and
        ``ToPILImage()(out*0.5+0.5)`` (:53) run inside the boundary kernels.  ``resize=(width, height)`` applies the script's
        ``transforms.Resize(..., LANCZOS)`` (:40-47, e.g. (512, 512) for "resize_512x512") before the generator and
        ``resize_back=True`` its ``output_pil.resize((input width, input height), Image.LANCZOS)`` (:53) after it, both on the
        device and bit-identical to Pillow (image_ops.lanczos_resize_u8)."""
        assert images_u8.dtype == torch.uint8 and images_u8.dim() == 4 and images_u8.shape[-1] == 3
        in_hw = images_u8.shape[1:3]
        if image_prep is not None:          # the script's build_transform(args.image_prep) (src/inference_unpaired.py:40, default "resize_512x512")
            from .image_ops import apply_image_prep
            with self._on_device():
                images_u8 = apply_image_prep(images_u8, image_prep, self.lib)
        if resize is not None:
            from .image_ops import lanczos_resize_u8
            with self._on_device():
                images_u8 = lanczos_resize_u8(images_u8, resize, self.lib)
        out = self.forward(images_u8, *args, _u8_io=(2.0, -1.0), **kw)
        if resize_back and tuple(out.shape[1:3]) != tuple(in_hw):
            from .image_ops import lanczos_resize_u8
            with self._on_device():
                out = lanczos_resize_u8(out, (in_hw[1], in_hw[0]), self.lib)
        return out

    @staticmethod
    @torch.no_grad()
    def forward_with_networks(x, direction, vae_enc, unet, vae_dec, sched, timesteps, text_emb, *, eps=None):
        """Static generator entry of the reference (src/cyclegan_turbo.py:199-207), called as
        ``CycleGAN_Turbo.forward_with_networks(x, "a2b", model.vae_enc, model.unet, model.vae_dec, model.sched,
        model.timesteps, text_emb)`` (src/train_cyclegan_turbo.py:181).  The three network arguments must be the handles of
        ONE model of this package (they name its planned program: vae_enc(x, direction) -> unet -> per-sample sched.step ->
        vae_dec(x, direction)); ``sched`` / ``timesteps`` are the fixed one-step schedule and are checked, not used."""
        model = getattr(unet, "_model", None)
        if model is None or getattr(vae_enc, "_model", None) is not model or getattr(vae_dec, "_model", None) is not model:
            raise ValueError("forward_with_networks: pass model.vae_enc, model.unet, model.vae_dec of one CycleGAN_Turbo")
        assert direction in ("a2b", "b2a")
        ts = [int(t) for t in (timesteps.tolist() if hasattr(timesteps, "tolist") else list(timesteps))]
        if any(t != model.weights.unet_arch.timestep for t in ts):
            raise ValueError("the planned forward is specialised for the reference's fixed timestep %d" % model.weights.unet_arch.timestep)
        return model.forward(x, direction=direction, caption_emb=text_emb, eps=eps)

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
