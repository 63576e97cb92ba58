"""UNet2DConditionModel (SD-2.1 architecture) forward (test infrastructure).

Restates diffusers 0.25.1 ``unet_2d_condition.py`` / ``unet_2d_blocks.py`` for
the sd-turbo config.  Called by the reference at
src/pix2pix_turbo.py:199,212 and src/cyclegan_turbo.py:204.
"""
import torch
import torch.nn.functional as F

from .arch import UNetArch
from .nn import (Weights, conv2d, group_norm, q, resnet_block, timestep_embedding,
                 transformer_2d)


def time_embedding(W: Weights, arch: UNetArch, t=None):
    t = arch.timestep if t is None else t
    e = timestep_embedding(t, arch.block_out_channels[0])
    w1, b1 = W.base("time_embedding.linear_1")
    w2, b2 = W.base("time_embedding.linear_2")
    return F.linear(F.silu(F.linear(e, w1, b1)), w2, b2)  # [1, 4*C0]


def conv_in(W: Weights, x, twin_r=None):
    """Plain conv_in, or TwinConv (src/pix2pix_turbo.py:16-26): pre*(1-r) + cur*r."""
    if W.has("conv_in.conv_in_pretrained.weight"):
        if twin_r is None:
            raise ValueError("TwinConv conv_in needs r (reference crashes with r=None, A.9 quirk 5)")
        x1 = conv2d(W, "conv_in.conv_in_pretrained", x, padding=1)
        x2 = conv2d(W, "conv_in.conv_in_curr", x, padding=1)
        return q(x1 * (1 - twin_r) + x2 * twin_r)
    return conv2d(W, "conv_in", x, padding=1)


def unet_forward(W: Weights, arch: UNetArch, x, ctx, twin_r=None, t=None):
    from .nn import unet_scope
    with unet_scope():
        return _unet_forward(W, arch, x, ctx, twin_r, t)


def _unet_forward(W: Weights, arch: UNetArch, x, ctx, twin_r=None, t=None):
    """x [B,4,h,w], ctx [B,77,cross_dim] -> eps prediction [B,4,h,w]."""
    g, eps = arch.norm_num_groups, arch.norm_eps
    boc, heads = arch.block_out_channels, arch.num_heads
    nb = len(boc)
    B = x.shape[0]
    temb = time_embedding(W, arch, t).expand(B, -1)
    if ctx.shape[0] == 1 and B > 1:
        ctx = ctx.expand(B, -1, -1)  # A.9 quirk 6: the new API broadcasts a single prompt

    h = conv_in(W, x, twin_r)
    res = [h]
    for i in range(nb):
        for j in range(arch.layers_per_block):
            h = resnet_block(W, f"down_blocks.{i}.resnets.{j}", h, g, eps, temb)
            if i < nb - 1:
                h = transformer_2d(W, f"down_blocks.{i}.attentions.{j}", h, ctx, heads[i], g)
            res.append(h)
        if i < nb - 1:
            h = conv2d(W, f"down_blocks.{i}.downsamplers.0.conv", h, stride=2, padding=1)
            res.append(h)

    h = resnet_block(W, "mid_block.resnets.0", h, g, eps, temb)
    h = transformer_2d(W, "mid_block.attentions.0", h, ctx, heads[-1], g)
    h = resnet_block(W, "mid_block.resnets.1", h, g, eps, temb)

    rheads = list(reversed(heads))
    for i in range(nb):
        for j in range(arch.layers_per_block + 1):
            skip = res.pop()
            h = torch.cat([h, skip], dim=1)
            h = resnet_block(W, f"up_blocks.{i}.resnets.{j}", h, g, eps, temb)
            if i > 0:
                h = transformer_2d(W, f"up_blocks.{i}.attentions.{j}", h, ctx, rheads[i], g)
        if i < nb - 1:
            # explicit output size = next skip's size (forward_upsample_size path, row f3)
            size = res[-1].shape[-2:]
            h = F.interpolate(h, size=size, mode="nearest") if tuple(size) != (h.shape[-2] * 2, h.shape[-1] * 2) \
                else F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = conv2d(W, f"up_blocks.{i}.upsamplers.0.conv", h, padding=1)
    assert not res
    h = q(F.silu(group_norm(W, "conv_norm_out", h, g, eps)))
    return conv2d(W, "conv_out", h, padding=1)
