"""Seeded synthetic weights with the exact diffusers/peft key layout (test infrastructure).

There are no SD-Turbo weights or reference ``.pkl`` checkpoints offline, so
parity runs on synthetic tensors of the real shapes under the real state-dict
keys.  Also emits the two reference checkpoint dict
layouts (src/pix2pix_turbo.py:221-229, src/train_cyclegan_turbo.py:293-307) so
the product loader can be round-trip tested.
"""
import math
from typing import Dict, List, Tuple

import torch

from .arch import UNetArch, VAEArch
from .pipeline import ModelWeights

# --------------------------------------------------------------------------------------
# layer enumeration: (name, kind, spec)  kind in {conv, linear, norm}
# conv spec = (cin, cout, k, bias) ; linear spec = (cin, cout, bias) ; norm spec = (c,)
# --------------------------------------------------------------------------------------


def _resnet(L, p, cin, cout, temb_dim=None):
    L.append((p + ".norm1", "norm", (cin,)))
    L.append((p + ".conv1", "conv", (cin, cout, 3, True)))
    if temb_dim:
        L.append((p + ".time_emb_proj", "linear", (temb_dim, cout, True)))
    L.append((p + ".norm2", "norm", (cout,)))
    L.append((p + ".conv2", "conv", (cout, cout, 3, True)))
    if cin != cout:
        L.append((p + ".conv_shortcut", "conv", (cin, cout, 1, True)))


def _vae_attn(L, p, c):
    L.append((p + ".group_norm", "norm", (c,)))
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        L.append((f"{p}.{n}", "linear", (c, c, True)))


def vae_layers(a: VAEArch) -> List[Tuple[str, str, tuple]]:
    L = []
    boc = a.block_out_channels
    nb = len(boc)
    L.append(("encoder.conv_in", "conv", (a.in_channels, boc[0], 3, True)))
    cin = boc[0]
    for i, c in enumerate(boc):
        for j in range(a.layers_per_block):
            _resnet(L, f"encoder.down_blocks.{i}.resnets.{j}", cin, c)
            cin = c
        if i < nb - 1:
            L.append((f"encoder.down_blocks.{i}.downsamplers.0.conv", "conv", (c, c, 3, True)))
    _resnet(L, "encoder.mid_block.resnets.0", cin, cin)
    _vae_attn(L, "encoder.mid_block.attentions.0", cin)
    _resnet(L, "encoder.mid_block.resnets.1", cin, cin)
    L.append(("encoder.conv_norm_out", "norm", (cin,)))
    L.append(("encoder.conv_out", "conv", (cin, 2 * a.latent_channels, 3, True)))
    L.append(("quant_conv", "conv", (2 * a.latent_channels, 2 * a.latent_channels, 1, True)))
    L.append(("post_quant_conv", "conv", (a.latent_channels, a.latent_channels, 1, True)))
    rev = list(reversed(boc))
    L.append(("decoder.conv_in", "conv", (a.latent_channels, rev[0], 3, True)))
    _resnet(L, "decoder.mid_block.resnets.0", rev[0], rev[0])
    _vae_attn(L, "decoder.mid_block.attentions.0", rev[0])
    _resnet(L, "decoder.mid_block.resnets.1", rev[0], rev[0])
    cin = rev[0]
    for i, c in enumerate(rev):
        for j in range(a.layers_per_block + 1):
            _resnet(L, f"decoder.up_blocks.{i}.resnets.{j}", cin, c)
            cin = c
        if i < nb - 1:
            L.append((f"decoder.up_blocks.{i}.upsamplers.0.conv", "conv", (c, c, 3, True)))
    L.append(("decoder.conv_norm_out", "norm", (cin,)))
    L.append(("decoder.conv_out", "conv", (cin, a.out_channels, 3, True)))
    for i, (ci, co) in enumerate(a.skip_conv_shapes):
        L.append((f"decoder.skip_conv_{i + 1}", "conv", (ci, co, 1, False)))
    return L


def _xformer(L, p, c, cross):
    L.append((p + ".norm", "norm", (c,)))
    L.append((p + ".proj_in", "linear", (c, c, True)))
    t = p + ".transformer_blocks.0"
    L.append((t + ".norm1", "norm", (c,)))
    for n in ("to_q", "to_k", "to_v"):
        L.append((f"{t}.attn1.{n}", "linear", (c, c, False)))
    L.append((t + ".attn1.to_out.0", "linear", (c, c, True)))
    L.append((t + ".norm2", "norm", (c,)))
    L.append((t + ".attn2.to_q", "linear", (c, c, False)))
    L.append((t + ".attn2.to_k", "linear", (cross, c, False)))
    L.append((t + ".attn2.to_v", "linear", (cross, c, False)))
    L.append((t + ".attn2.to_out.0", "linear", (c, c, True)))
    L.append((t + ".norm3", "norm", (c,)))
    L.append((t + ".ff.net.0.proj", "linear", (c, 8 * c, True)))
    L.append((t + ".ff.net.2", "linear", (4 * c, c, True)))
    L.append((p + ".proj_out", "linear", (c, c, True)))


def unet_layers(a: UNetArch) -> List[Tuple[str, str, tuple]]:
    L = []
    boc = a.block_out_channels
    nb = len(boc)
    td = a.time_embed_dim
    L.append(("time_embedding.linear_1", "linear", (boc[0], td, True)))
    L.append(("time_embedding.linear_2", "linear", (td, td, True)))
    L.append(("conv_in", "conv", (a.in_channels, boc[0], 3, True)))
    cin = boc[0]
    skips = [boc[0]]
    for i, c in enumerate(boc):
        for j in range(a.layers_per_block):
            _resnet(L, f"down_blocks.{i}.resnets.{j}", cin, c, td)
            cin = c
            if i < nb - 1:
                _xformer(L, f"down_blocks.{i}.attentions.{j}", c, a.cross_attention_dim)
            skips.append(c)
        if i < nb - 1:
            L.append((f"down_blocks.{i}.downsamplers.0.conv", "conv", (c, c, 3, True)))
            skips.append(c)
    _resnet(L, "mid_block.resnets.0", cin, cin, td)
    _xformer(L, "mid_block.attentions.0", cin, a.cross_attention_dim)
    _resnet(L, "mid_block.resnets.1", cin, cin, td)
    rev = list(reversed(boc))
    prev = cin
    for i, c in enumerate(rev):
        for j in range(a.layers_per_block + 1):
            sk = skips.pop()
            _resnet(L, f"up_blocks.{i}.resnets.{j}", prev + sk, c, td)
            prev = c
            if i > 0:
                _xformer(L, f"up_blocks.{i}.attentions.{j}", c, a.cross_attention_dim)
        if i < nb - 1:
            L.append((f"up_blocks.{i}.upsamplers.0.conv", "conv", (c, c, 3, True)))
    assert not skips
    L.append(("conv_norm_out", "norm", (prev,)))
    L.append(("conv_out", "conv", (prev, a.out_channels, 3, True)))
    return L


def count_params(layers) -> int:
    n = 0
    for _, kind, s in layers:
        if kind == "conv":
            n += s[0] * s[1] * s[2] * s[2] + (s[1] if s[3] else 0)
        elif kind == "linear":
            n += s[0] * s[1] + (s[1] if s[2] else 0)
        else:
            n += 2 * s[0]
    return n


# --------------------------------------------------------------------------------------
# LoRA target matching (peft: module name == t or name.endswith("." + t)); SURVEY A.7
# --------------------------------------------------------------------------------------
PIX2PIX_VAE_TARGETS = ["conv1", "conv2", "conv_in", "conv_shortcut", "conv", "conv_out",
                       "skip_conv_1", "skip_conv_2", "skip_conv_3", "skip_conv_4",
                       "to_k", "to_q", "to_v", "to_out.0"]       # src/pix2pix_turbo.py:137-140
PIX2PIX_UNET_TARGETS = ["to_k", "to_q", "to_v", "to_out.0", "conv", "conv1", "conv2", "conv_shortcut",
                        "conv_out", "proj_in", "proj_out", "ff.net.2", "ff.net.0.proj"]  # :144-147
CYCLEGAN_GREP = ["to_k", "to_q", "to_v", "to_out.0", "conv", "conv1", "conv2", "conv_in", "conv_shortcut",
                 "conv_out", "proj_out", "proj_in", "ff.net.2", "ff.net.0.proj"]  # src/cyclegan_turbo.py:53


def peft_match(name: str, targets) -> bool:
    return any(name == t or name.endswith("." + t) for t in targets)


def cyclegan_unet_target_split(a: UNetArch):
    """src/cyclegan_turbo.py:52-65: substring grep over *parameter* names."""
    enc, dec, oth = [], [], []
    for name, kind, s in unet_layers(a):
        if kind == "norm":
            continue
        n = name + ".weight"
        if "norm" in n:
            continue
        for pat in CYCLEGAN_GREP:
            if pat in n and ("down_blocks" in n or "conv_in" in n):
                enc.append(name)
                break
            elif pat in n and "up_blocks" in n:
                dec.append(name)
                break
            elif pat in n:
                oth.append(name)
                break
    return enc, dec, oth


# --------------------------------------------------------------------------------------
# tensor synthesis
# --------------------------------------------------------------------------------------
def _fill(sd, layers, gen, lora: Dict[str, Tuple[str, int]]):
    """lora: layer name -> (adapter name, rank)."""
    for name, kind, s in layers:
        if kind == "norm":
            sd[name + ".weight"] = 1.0 + 0.1 * torch.randn(s[0], generator=gen)
            sd[name + ".bias"] = 0.1 * torch.randn(s[0], generator=gen)
            continue
        if kind == "conv":
            cin, cout, k, bias = s
            shape, fan_in = (cout, cin, k, k), cin * k * k
        else:
            cin, cout, bias = s
            shape, fan_in = (cout, cin), cin
        std = 0.02 if "skip_conv" in name else 1.0 / math.sqrt(fan_in)
        w = std * torch.randn(shape, generator=gen)
        b = 0.02 * torch.randn(cout, generator=gen) if bias else None
        if name in lora:
            ad, r = lora[name]
            sd[name + ".base_layer.weight"] = w
            if b is not None:
                sd[name + ".base_layer.bias"] = b
            a_shape = (r, cin, k, k) if kind == "conv" else (r, cin)
            b_shape = (cout, r, 1, 1) if kind == "conv" else (cout, r)
            # peft inits B = 0 (would hide LoRA bugs): use small non-zero B instead
            sd[f"{name}.lora_A.{ad}.weight"] = torch.randn(a_shape, generator=gen) / math.sqrt(fan_in)
            sd[f"{name}.lora_B.{ad}.weight"] = torch.randn(b_shape, generator=gen) * (0.3 / math.sqrt(r))
        else:
            sd[name + ".weight"] = w
            if b is not None:
                sd[name + ".bias"] = b


def make_pix2pix_weights(unet_arch: UNetArch, vae_arch: VAEArch, seed=1234, rank_unet=8, rank_vae=4,
                         sketch=False) -> ModelWeights:
    """Pix2Pix_Turbo weights as after src/pix2pix_turbo.py:66-78 (LoRA injected).

    lora_alpha defaults to 8 (LoraConfig default) => scaling 8/r (A.7).
    ``sketch``: conv_in becomes a TwinConv (src/pix2pix_turbo.py:100-101).
    """
    gen = torch.Generator().manual_seed(seed)
    ul, vl = unet_layers(unet_arch), vae_layers(vae_arch)
    unet_lora = {n: ("default", rank_unet) for n, k, _ in ul if k != "norm" and peft_match(n, PIX2PIX_UNET_TARGETS)}
    vae_lora = {n: ("vae_skip", rank_vae) for n, k, _ in vl if k != "norm" and peft_match(n, PIX2PIX_VAE_TARGETS)}
    unet, vae = {}, {}
    _fill(unet, ul, gen, unet_lora)
    _fill(vae, vl, gen, vae_lora)
    if sketch:
        w, b = unet.pop("conv_in.weight"), unet.pop("conv_in.bias")
        unet["conv_in.conv_in_pretrained.weight"] = w
        unet["conv_in.conv_in_pretrained.bias"] = b
        unet["conv_in.conv_in_curr.weight"] = w + 0.05 * torch.randn(w.shape, generator=gen)
        unet["conv_in.conv_in_curr.bias"] = b + 0.01 * torch.randn(b.shape, generator=gen)
    return ModelWeights(unet, vae, unet_arch, vae_arch,
                        unet_scaling={"default": 8.0 / rank_unet}, vae_scaling={"vae_skip": 8.0 / rank_vae})


def make_cyclegan_weights(unet_arch: UNetArch, vae_arch: VAEArch, seed=4321, rank_unet=128, rank_vae=4) -> ModelWeights:
    """CycleGAN_Turbo weights as after src/cyclegan_turbo.py:162-190: three UNet adapters
    (lora_alpha = rank => scaling 1), two complete VAEs (vae, vae_b2a)."""
    gen = torch.Generator().manual_seed(seed)
    ul, vl = unet_layers(unet_arch), vae_layers(vae_arch)
    enc, dec, oth = cyclegan_unet_target_split(unet_arch)
    unet_lora = {}
    for names, ad in ((enc, "default_encoder"), (dec, "default_decoder"), (oth, "default_others")):
        for n in names:
            unet_lora[n] = (ad, rank_unet)
    vae_lora = {n: ("vae_skip", rank_vae) for n, k, _ in vl if k != "norm" and peft_match(n, PIX2PIX_VAE_TARGETS)}
    unet, vae, vae_b2a = {}, {}, {}
    _fill(unet, ul, gen, unet_lora)
    _fill(vae, vl, gen, vae_lora)
    _fill(vae_b2a, vl, gen, vae_lora)
    return ModelWeights(unet, vae, unet_arch, vae_arch,
                        unet_scaling={"default_encoder": 1.0, "default_decoder": 1.0, "default_others": 1.0},
                        vae_scaling={"vae_skip": 8.0 / rank_vae}, vae_b2a=vae_b2a)


# --------------------------------------------------------------------------------------
# reference checkpoint layouts + the "pretrained base" they overlay
# --------------------------------------------------------------------------------------
def _strip_lora(sd):
    """The base (pre-injection) state dict: base_layer.* -> plain keys, LoRA tensors dropped."""
    out = {}
    for k, v in sd.items():
        if ".lora_A." in k or ".lora_B." in k:
            continue
        out[k.replace(".base_layer.", ".")] = v
    return out


def split_pix2pix_checkpoint(mw: ModelWeights, rank_unet=8, rank_vae=4):
    """-> (base_unet_sd, base_vae_sd, ckpt) where ckpt has the save_model layout
    (src/pix2pix_turbo.py:221-229): UNet keys containing "lora"/"conv_in", VAE keys
    containing "lora"/"skip".  base_* are what ``from_pretrained`` would give."""
    base_unet = _strip_lora(mw.unet)
    base_vae = {k: v for k, v in _strip_lora(mw.vae).items() if "skip_conv" not in k}
    if "conv_in.conv_in_pretrained.weight" in base_unet:  # sketch model: hub conv_in = the pretrained twin
        base_unet["conv_in.weight"] = base_unet.pop("conv_in.conv_in_pretrained.weight")
        base_unet["conv_in.bias"] = base_unet.pop("conv_in.conv_in_pretrained.bias")
        base_unet.pop("conv_in.conv_in_curr.weight")
        base_unet.pop("conv_in.conv_in_curr.bias")
    ckpt = {
        "unet_lora_target_modules": list(PIX2PIX_UNET_TARGETS),
        "vae_lora_target_modules": list(PIX2PIX_VAE_TARGETS),
        "rank_unet": rank_unet,
        "rank_vae": rank_vae,
        "state_dict_unet": {k: v for k, v in mw.unet.items() if "lora" in k or "conv_in" in k},
        "state_dict_vae": {k: v for k, v in mw.vae.items() if "lora" in k or "skip" in k},
    }
    return base_unet, base_vae, ckpt


def split_cyclegan_checkpoint(mw: ModelWeights, rank_unet=128, rank_vae=4):
    """-> (base_unet_sd, base_vae_sd, ckpt) in the src/train_cyclegan_turbo.py:293-307 layout:
    sd_encoder/sd_decoder/sd_other with the adapter name stripped from keys, and
    sd_vae_enc / sd_vae_dec = full VAE_encode/VAE_decode wrapper state dicts."""
    enc, dec, oth = cyclegan_unet_target_split(mw.unet_arch)
    base_unet = _strip_lora(mw.unet)
    base_vae = {k: v for k, v in _strip_lora(mw.vae).items() if "skip_conv" not in k}

    def grab(ad):
        out = {}
        for k, v in mw.unet.items():
            if f".{ad}.weight" in k and "lora" in k:
                out[k.replace(f".{ad}.weight", ".weight")] = v
        return out

    wrapper = {}
    for k, v in mw.vae.items():
        wrapper["vae." + k] = v
    for k, v in mw.vae_b2a.items():
        wrapper["vae_b2a." + k] = v
    ckpt = {
        "l_target_modules_encoder": enc, "l_target_modules_decoder": dec, "l_modules_others": oth,
        "rank_unet": rank_unet,
        "sd_encoder": grab("default_encoder"), "sd_decoder": grab("default_decoder"),
        "sd_other": grab("default_others"),
        "rank_vae": rank_vae, "vae_lora_target_modules": list(PIX2PIX_VAE_TARGETS),
        "sd_vae_enc": dict(wrapper), "sd_vae_dec": dict(wrapper),
    }
    return base_unet, base_vae, ckpt


# --------------------------------------------------------------------------------------
# synthetic inputs
# --------------------------------------------------------------------------------------
def make_inputs(kind, B, H, W, cross_dim, seed=0, latent_channels=4):
    """kind: 'canny' (Bernoulli(0.08) in {0,1}, 3 equal channels), 'sketch' (Bernoulli(0.05)),
    'photo' (low-passed U(-1,1))."""
    g = torch.Generator().manual_seed(1234 + seed)
    if kind in ("canny", "sketch"):
        p = 0.08 if kind == "canny" else 0.05
        x = (torch.rand(B, 1, H, W, generator=g) < p).float().expand(B, 3, H, W).contiguous()
    else:
        x = torch.rand(B, 3, H, W, generator=g) * 2 - 1
        x = torch.nn.functional.avg_pool2d(x, 9, 1, 4)
        x = (x / x.abs().amax(dim=(1, 2, 3), keepdim=True)).contiguous()
    caption = torch.randn(1, 77, cross_dim, generator=g)
    eps = torch.randn(B, latent_channels, H // 8, W // 8, generator=g)
    noise_map = torch.randn(B, latent_channels, H // 8, W // 8, generator=g)
    return x, caption, eps, noise_map
