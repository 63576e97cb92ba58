"""AutoencoderKL with the reference's patched forwards (test infrastructure).

Follows src/model.py:14-27 (encoder, records the 4 pre-down-block activations)
and src/model.py:30-54 (decoder, adds skip_conv_i(skip * gamma) before each up
block) over diffusers 0.25.1 Encoder/Decoder blocks.
"""
import torch
import torch.nn.functional as F

from .arch import VAEArch
from .nn import Weights, conv2d, group_norm, q, resnet_block, vae_attention


def encoder_forward(W: Weights, arch: VAEArch, x):
    """-> (moments [B, 2*latent, h, w], skips [s0..s3]).  src/model.py:14-27 + quant_conv."""
    g, eps = arch.norm_num_groups, arch.eps
    h = conv2d(W, "encoder.conv_in", x, padding=1)
    skips = []
    nb = len(arch.block_out_channels)
    for i in range(nb):
        skips.append(h)  # src/model.py:18-20: recorded BEFORE the block
        for j in range(arch.layers_per_block):
            h = resnet_block(W, f"encoder.down_blocks.{i}.resnets.{j}", h, g, eps)
        if i < nb - 1:
            # Downsample2D(padding=0): F.pad (0,1,0,1) then conv s2 p0
            h = conv2d(W, f"encoder.down_blocks.{i}.downsamplers.0.conv", F.pad(h, (0, 1, 0, 1)), stride=2)
    h = resnet_block(W, "encoder.mid_block.resnets.0", h, g, eps)
    h = vae_attention(W, "encoder.mid_block.attentions.0", h, g, eps)
    h = resnet_block(W, "encoder.mid_block.resnets.1", h, g, eps)
    h = q(F.silu(group_norm(W, "encoder.conv_norm_out", h, g, eps)))
    h = conv2d(W, "encoder.conv_out", h, padding=1)
    return conv2d(W, "quant_conv", h), skips


def posterior_sample(moments, eps_noise):
    """DiagonalGaussianDistribution.sample(): mean + exp(.5*clamp(logvar,-30,20)) * eps."""
    mean, logvar = moments.chunk(2, dim=1)
    std = torch.exp(0.5 * logvar.clamp(-30.0, 20.0))
    return mean + std * eps_noise


def decoder_forward(W: Weights, arch: VAEArch, z, skips, gamma=1.0):
    """post_quant_conv + src/model.py:30-54.  ``skips`` = encoder's [s0..s3]."""
    g, eps = arch.norm_num_groups, arch.eps
    h = conv2d(W, "post_quant_conv", z)
    h = conv2d(W, "decoder.conv_in", h, padding=1)
    h = resnet_block(W, "decoder.mid_block.resnets.0", h, g, eps)
    h = vae_attention(W, "decoder.mid_block.attentions.0", h, g, eps)
    h = resnet_block(W, "decoder.mid_block.resnets.1", h, g, eps)
    nb = len(arch.block_out_channels)
    for i in range(nb):
        if skips is not None:  # ignore_skip == False
            h = q(h + conv2d(W, f"decoder.skip_conv_{i + 1}", q(skips[::-1][i] * gamma)))
        for j in range(arch.layers_per_block + 1):
            h = resnet_block(W, f"decoder.up_blocks.{i}.resnets.{j}", h, g, eps)
        if i < nb - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = conv2d(W, f"decoder.up_blocks.{i}.upsamplers.0.conv", h, padding=1)
    h = q(F.silu(group_norm(W, "decoder.conv_norm_out", h, g, eps)))
    return conv2d(W, "decoder.conv_out", h, padding=1)
