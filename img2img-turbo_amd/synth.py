"""Random-init generator weights with the real shapes and state-dict keys (benchmarks / smoke runs).

There is no network for SD-Turbo or the reference's LoRA checkpoints, so throughput is measured on
random weights of the exact architecture: diffusers/peft key layout after adapter injection
(src/pix2pix_turbo.py:66-78, src/cyclegan_turbo.py:162-190).  (oracle/synth.py is the test-side twin.)
"""
import math
from typing import Dict, List, Tuple

import torch

from .arch import UNetArch, VAEArch
from .weights import GeneratorWeights

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
                         sketch=False) -> GeneratorWeights:
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
    return GeneratorWeights(unet, vae, unet_arch, vae_arch,
                        unet_scaling={"default": 8.0 / rank_unet}, vae_scaling={"vae_skip": 8.0 / rank_vae})


def make_cyclegan_weights(unet_arch: UNetArch, vae_arch: VAEArch, seed=4321, rank_unet=128, rank_vae=4) -> GeneratorWeights:
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
    return GeneratorWeights(unet, vae, unet_arch, vae_arch,
                        unet_scaling={"default_encoder": 1.0, "default_decoder": 1.0, "default_others": 1.0},
                        vae_scaling={"vae_skip": 8.0 / rank_vae}, vae_b2a=vae_b2a)


