"""Pix2Pix_Turbo with the reference's API surface (src/pix2pix_turbo.py:29-229), MI355X-native inside.

``forward(c_t, prompt=None, prompt_tokens=None, deterministic=True, r=1.0, noise_map=None)`` keeps the
reference signature and semantics (:186-219).  Differences, all explicit:
  * weights come from local files / in-memory state dicts (no HTTP download, :48-64);
  * the LoRA scale / skip gamma / TwinConv blend ``r`` are per-call arguments of the planned program, so a
    stochastic call does not leave state behind (reference quirk: :206-207,217 persist);
  * a single prompt is broadcast over the batch (reference: shape error for B>1);
  * the two RNG draws (posterior eps :198, scheduler noise :200) are drawn with torch on the device in the
    same order; ``eps`` may also be injected for parity tests;
  * text conditioning: pass ``caption_enc`` ([1|B,77,1024]) or give the model a ``text_encoder`` /
    ``tokenizer`` pair (CLIP weights are not available offline).
All arithmetic runs in the HIP kernels behind include/i2i_turbo.h; there is no torch/diffusers fallback.
"""
import collections
from typing import Optional

import torch

from . import _capi
from .model import make_1step_sched
from .plan import ForwardPlan
from .packer import Packer
from .weights import GeneratorWeights, from_pix2pix_checkpoint, load_checkpoint_file, load_sd_turbo_base


class _NetHandle:
    """What callers reach through ``model.unet`` / ``model.vae`` / ``model.vae_enc`` / ``model.vae_dec``
    (src/pix2pix_turbo.py:161, src/cyclegan_turbo.py:186-190): the canonical state dict of that network plus the no-op
    switches the reference scripts call on it (``enable_xformers_memory_efficient_attention`` src/inference_unpaired.py:36,
    ``eval`` / ``requires_grad_`` / ``to``).  The arithmetic itself lives in the planned program, not in these objects."""

    def __init__(self, name, sd, sd_b2a=None):
        self.name, self._sd, self._sd_b2a = name, sd, sd_b2a

    def state_dict(self):
        if self._sd_b2a is None:
            return dict(self._sd)
        out = {"vae." + k: v for k, v in self._sd.items()}         # VAE_encode / VAE_decode wrap both VAEs (:15-45)
        out.update({"vae_b2a." + k: v for k, v in self._sd_b2a.items()})
        return out

    def enable_xformers_memory_efficient_attention(self):
        return None            # the fused flash-style attention kernel is always on

    def eval(self):
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("training (autograd through the generator) is out of scope of the MI355X forward path")
        return self

    def requires_grad_(self, flag=False):
        return self

    def to(self, *a, **k):
        return self


class TurboGeneratorBase(torch.nn.Module):
    """Shared plumbing: plan cache (LRU, graphs released on eviction), packers (one per network and dtype; the LoRA scale r is
    re-merged on the device), boundary staging, text conditioning."""

    MAX_PLANS = 8          # distinct (batch, size, mode) programs kept alive; each owns an activation pool + a hipGraph

    def __init__(self, weights: GeneratorWeights, device="cuda", dtype=torch.float32, lib=None,
                 tokenizer=None, text_encoder=None, use_graph=True, fuse_gn=True, flash=True, plan_options=None, unet_dtype=None):
        """``dtype``: element type of activations and packed weights (fp32 = the exact-MFMA parity mode).  ``unet_dtype``: another 16-bit
        type for the UNet alone -- ``dtype=torch.bfloat16, unet_dtype=torch.float16`` keeps the VAE (whose real activations overflow fp16)
        in bf16 and gives the UNet, whose error the 1-step scheduler multiplies by 14.6, fp16's three extra mantissa bits."""
        super().__init__()
        self.unet_dtype_ = unet_dtype
        self.weights = weights
        self.device_ = torch.device(device)
        if self.device_.type == "cuda" and self.device_.index is None:
            self.device_ = torch.device("cuda", torch.cuda.current_device())
        self.dtype_ = dtype
        self.lib = lib or _capi.default_library()     # raises if the HIP build is missing: no fallback
        if self.lib.backend == "gfx950" and self.device_.type != "cuda":
            raise _capi.I2IError("the gfx950 kernel library needs a CUDA/HIP device, got %s" % device)
        self.tokenizer, self.text_encoder = tokenizer, text_encoder
        self.sched = make_1step_sched()
        self.timesteps = self.sched.timesteps
        self.use_graph = use_graph and self.lib.backend == "gfx950"
        self.fuse_gn, self.flash = fuse_gn, flash
        self.plan_options = dict(plan_options or {})   # extra ForwardPlan switches (tests / ablations): halo_min_tiles, subpix, ...
        self._plans = collections.OrderedDict()
        self._packers = {}
        self._caption_cache = {}
        self._r = 1.0
        self.unet = _NetHandle("unet", weights.unet)

    # ---- reference-compatible no-ops / switches ----
    def set_eval(self):
        return self.eval()

    def set_train(self):
        raise NotImplementedError("training (autograd through the generator) is out of scope of the MI355X forward path")

    def half(self):
        return self.to_dtype(torch.float16)

    def bfloat16(self):
        return self.to_dtype(torch.bfloat16)

    def float(self):
        return self.to_dtype(torch.float32)

    def to_dtype(self, dtype):
        if dtype != self.dtype_ or self.unet_dtype_ is not None:
            self.dtype_, self.unet_dtype_ = dtype, None
            self.release_plans()
            self._packers.clear()
        return self

    def release_plans(self):
        for plan in self._plans.values():
            plan.release()
        self._plans.clear()

    # ---- plumbing ----
    def _on_device(self):
        import contextlib
        return torch.cuda.device(self.device_) if self.device_.type == "cuda" else contextlib.nullcontext()

    def _packer(self, which):
        """One packer per network ('unet', 'vae', 'vae_b2a') and dtype: base weights are packed and uploaded once, the LoRA
        merge at the current r runs on the device (Packer.set_scale)."""
        dt = (self.unet_dtype_ or self.dtype_) if which == "unet" else self.dtype_
        key = (dt, which)
        if key not in self._packers:
            w = self.weights
            sd, sc = {"unet": (w.unet, w.unet_scaling), "vae": (w.vae, w.vae_scaling), "vae_b2a": (w.vae_b2a, w.vae_scaling)}[which]
            with self._on_device():
                self._packers[key] = Packer(sd, sc, dt, self.device_, self.lib, self._r, self._r)
        return self._packers[key]

    def _get_packers(self, direction):
        vae = "vae" if (direction == "a2b" or self.weights.vae_b2a is None) else "vae_b2a"
        return self._packer("unet"), self._packer(vae)

    def set_lora_scale(self, r: float):
        """The reference's per-call ``unet.set_adapters(["default"], weights=[r])`` +
        ``set_weights_and_activate_adapters(vae, ["vae_skip"], [r])`` + ``decoder.gamma = r`` + ``conv_in.r = r``
        (src/pix2pix_turbo.py:206-217) as ONE device-side re-merge of the packed weights (csrc/lora_merge.hip): no re-pack,
        no re-upload, plans and captured graphs stay valid.  Asynchronous on the current stream."""
        r = float(r)
        if r != self._r:
            self._r = r
            with self._on_device():
                for pk in self._packers.values():
                    pk.set_scale(r, r)

    def get_plan(self, B, H, W, stochastic=False, r=1.0, direction="a2b", ctx_batch=1, u8_io=None) -> ForwardPlan:
        """The cached plan for this shape.  `r` is not part of the key (device addresses do not depend on it): ONE plan object
        per key, which records the r of the most recent get_plan() for that key and re-applies it whenever it runs
        (ForwardPlan.before_run).  Plans of DIFFERENT keys (a stochastic and a deterministic one, two sizes) can be held and
        interleaved safely; two r values for the SAME key are the same object -- the last get_plan() wins, so ask again
        (it is a dictionary lookup) before running with another r."""
        r_plan = float(r) if stochastic else 1.0
        self.set_lora_scale(r_plan)
        key = (B, H, W, self.dtype_, self.unet_dtype_, stochastic, direction, ctx_batch, u8_io)
        if key in self._plans:
            self._plans.move_to_end(key)
        else:
            opts = dict(self.plan_options)
            if u8_io is not None:
                opts["u8_io"] = u8_io
            with self._on_device():
                self._plans[key] = ForwardPlan(self.lib, self.weights, B, H, W, self.dtype_, self.device_, stochastic=stochastic,
                                               r=self._r, direction=direction, ctx_batch=ctx_batch, fuse_gn=self.fuse_gn,
                                               flash=self.flash, packers=self._get_packers(direction), unet_dtype=self.unet_dtype_, **opts)
            while len(self._plans) > self.MAX_PLANS:      # LRU: the evicted plan's graph and activation pool are released
                _, old = self._plans.popitem(last=False)
                old.release()
        plan = self._plans[key]
        plan.r = r_plan
        plan.before_run = self._apply_plan_r
        return plan

    def _apply_plan_r(self, plan):
        self.set_lora_scale(plan.r)

    def encode_prompt(self, prompt=None, prompt_tokens=None):
        """tokenizer(prompt, max_length=77, padding="max_length", truncation=True) -> text_encoder(ids)[0]
        (src/pix2pix_turbo.py:190-196).  Cached per prompt string."""
        if self.text_encoder is None:
            raise _capi.I2IError("no text encoder attached: pass caption_enc=[1|B,77,%d] or construct the model with "
                                 "tokenizer= and text_encoder=" % self.weights.unet_arch.cross_attention_dim)
        if prompt is not None:
            key = prompt if isinstance(prompt, str) else tuple(prompt)
            if key not in self._caption_cache:
                ids = self.tokenizer(prompt, max_length=self.tokenizer.model_max_length, padding="max_length",
                                     truncation=True, return_tensors="pt").input_ids
                self._caption_cache[key] = self.text_encoder(ids.to(self.device_))[0].detach()
            return self._caption_cache[key]
        ids = prompt_tokens.reshape(-1, prompt_tokens.shape[-1])   # [B,1,77] from the training dataset (A.5)
        return self.text_encoder(ids.to(self.device_))[0].detach()

    def stage(self, plan: ForwardPlan, x, caption_enc, eps, noise_map=None):
        """Copy one batch into the plan's static boundary buffers (device-to-device when the inputs are already resident)."""
        plan.x_in.copy_(x)
        plan.ctx.copy_(caption_enc.reshape(plan.ctx.shape))
        plan.eps.copy_(eps)
        if noise_map is not None:
            plan.noise.copy_(noise_map.expand_as(plan.noise))

    def _execute(self, plan: ForwardPlan, x, caption_enc, eps, noise_map=None):
        with self._on_device():
            self.stage(plan, x, caption_enc, eps, noise_map)
            if self.use_graph:
                plan.replay()
            else:
                plan.run()
            return plan.out.clone()


class Pix2Pix_Turbo(TurboGeneratorBase):
    def __init__(self, pretrained_name=None, pretrained_path=None, ckpt_folder="checkpoints", lora_rank_unet=8, lora_rank_vae=4,
                 *, weights: Optional[GeneratorWeights] = None, base_dir=None, **kw):
        if weights is None:
            # local-file counterpart of src/pix2pix_turbo.py:47-130: base SD-Turbo snapshot + LoRA .pkl
            import os
            base_dir = base_dir or os.environ.get("I2I_SD_TURBO_DIR")      # so Pix2Pix_Turbo(pretrained_name=...) works as written
            if base_dir is None:                                            # in src/inference_paired.py:31
                raise ValueError("give weights=GeneratorWeights(...), base_dir=<local sd-turbo snapshot> or set I2I_SD_TURBO_DIR "
                                 "(the reference downloads stabilityai/sd-turbo from the hub; there is no network here)")
            names = {"edge_to_image": "edge_to_image_loras.pkl", "sketch_to_image_stochastic": "sketch_to_image_stochastic_lora.pkl"}
            if pretrained_name in names:
                pretrained_path = os.path.join(ckpt_folder, names[pretrained_name])
            if pretrained_path is None:
                raise ValueError("pretrained_name or pretrained_path required (random-init training models are out of scope)")
            unet, vae = load_sd_turbo_base(base_dir)
            weights = from_pix2pix_checkpoint(unet, vae, load_checkpoint_file(pretrained_path))
        super().__init__(weights, **kw)
        self.lora_rank_unet, self.lora_rank_vae = lora_rank_unet, lora_rank_vae
        self.vae = _NetHandle("vae", weights.vae)
        self.target_modules_unet = weights.meta.get("unet_lora_target_modules")
        self.target_modules_vae = weights.meta.get("vae_lora_target_modules")
        if weights.meta.get("rank_unet"):
            self.lora_rank_unet, self.lora_rank_vae = weights.meta["rank_unet"], weights.meta["rank_vae"]

    @torch.no_grad()
    def forward_u8(self, images_u8, *args, resize=None, sketch=False, **kw):
        """uint8 HWC in, uint8 HWC out ([B, H, W, 3] on the device): the callers' ``F.to_tensor`` (src/inference_paired.py:50)
        and ``ToPILImage()(out*0.5+0.5)`` (:72) run inside the boundary kernels; everything else as ``forward``.
        ``resize="multiple_of_8"`` first applies the script's ``input_image.resize((w - w % 8, h - h % 8), Image.LANCZOS)``
        (:38-41) on the device, bit-identical to Pillow (image_ops.lanczos_resize_u8); ``resize=(width, height)`` any size.
        ``sketch=True``: the sketch branch's binarisation ``F.to_tensor(input_image) < 0.5`` (:57-58) instead of ``to_tensor``
        (bytes below 128 become 1.0, the others 0.0), as the stochastic sketch model expects."""
        assert images_u8.dtype == torch.uint8 and images_u8.dim() == 4 and images_u8.shape[-1] == 3
        if resize is not None:
            from .image_ops import lanczos_resize_u8, resize_to_multiple_of_8
            with self._on_device():
                images_u8 = resize_to_multiple_of_8(images_u8, self.lib) if resize == "multiple_of_8" else lanczos_resize_u8(images_u8, resize, self.lib)
        return self.forward(images_u8, *args, _u8_io=(1.0, 0.0, 128) if sketch else (1.0, 0.0), **kw)

    @torch.no_grad()
    def forward(self, c_t, prompt=None, prompt_tokens=None, deterministic=True, r=1.0, noise_map=None,
                *, caption_enc=None, eps=None, _u8_io=None):
        if caption_enc is None:
            # either the prompt or the prompt_tokens should be provided (src/pix2pix_turbo.py:188)
            assert (prompt is None) != (prompt_tokens is None), "Either prompt or prompt_tokens should be provided"
            caption_enc = self.encode_prompt(prompt, prompt_tokens)
        if _u8_io is not None:
            B, H, W, _ = c_t.shape
        else:
            B, _, H, W = c_t.shape
        if not deterministic:
            if noise_map is None:
                raise ValueError("stochastic forward needs noise_map (src/pix2pix_turbo.py:210)")
        elif self.weights.is_twin_conv:
            raise ValueError("this checkpoint wraps conv_in in a TwinConv, which the reference can only run with "
                             "deterministic=False (TwinConv.r is None otherwise, src/pix2pix_turbo.py:21-26)")
        lat = self.weights.vae_arch.latent_channels
        if eps is None:   # latent_dist.sample() draw, then the (numerically dead) scheduler draw: same order as the reference
            eps = torch.randn(B, lat, H // 8, W // 8, device=self.device_, dtype=torch.float32)
            torch.randn(B, lat, H // 8, W // 8, device=self.device_, dtype=torch.float32)
        ctx_batch = caption_enc.shape[0] if caption_enc.dim() == 3 else 1
        assert ctx_batch in (1, B)
        plan = self.get_plan(B, H, W, stochastic=not deterministic, r=r, ctx_batch=ctx_batch, u8_io=_u8_io)
        out = self._execute(plan, c_t, caption_enc, eps, None if deterministic else noise_map)
        if _u8_io is not None:
            return out
        return out.to(c_t.dtype) if c_t.dtype in (torch.float16, torch.bfloat16, torch.float32) else out

    def save_model(self, outf):
        """Same dict layout as the reference (src/pix2pix_turbo.py:221-229): keys as ``unet.state_dict()`` /
        ``vae.state_dict()`` name them after adapter injection (``X.base_layer.weight``, ``X.lora_A.<adapter>.weight``),
        filtered with the reference's substring tests, so the file loads back through from_pix2pix_checkpoint (and into the
        reference itself)."""
        sd = {"unet_lora_target_modules": self.target_modules_unet, "vae_lora_target_modules": self.target_modules_vae,
              "rank_unet": self.lora_rank_unet, "rank_vae": self.lora_rank_vae,
              "state_dict_unet": {k: v for k, v in self.weights.unet.items() if "lora" in k or "conv_in" in k},
              "state_dict_vae": {k: v for k, v in self.weights.vae.items() if "lora" in k or "skip" in k}}
        torch.save(sd, outf)
