"""Network shapes of the generator: the sd-turbo AutoencoderKL and UNet2DConditionModel configs
(diffusers 0.25.1; built by the reference at src/pix2pix_turbo.py:36,45) plus a tiny same-topology
variant for fast checks.  (oracle/arch.py is the test-side twin; the product never imports oracle/.)
"""
from dataclasses import dataclass, field
from typing import Tuple


@dataclass(frozen=True)
class VAEArch:
    in_channels: int = 3
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    latent_channels: int = 4
    norm_num_groups: int = 32
    scaling_factor: float = 0.18215
    eps: float = 1e-6

    @property
    def skip_conv_shapes(self):
        """(cin, cout) of decoder.skip_conv_1..4 (src/pix2pix_turbo.py:40-43)."""
        b = self.block_out_channels
        # skip_conv_1: s3 -> up0 input ; _2: s2 -> up1 ; _3: s1 -> up2 ; _4: s0 -> up3
        rev = list(reversed(b))  # [512, 512, 256, 128]
        skips = [b[0], b[0], b[1], b[2]]  # channels of s0..s3
        up_in = [rev[0], rev[0], rev[1], rev[2]]  # input channels of up0..up3
        return [(skips[3 - i], up_in[i]) for i in range(4)]


@dataclass(frozen=True)
class UNetArch:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    num_heads: Tuple[int, ...] = (5, 10, 20, 20)  # "attention_head_dim" legacy naming
    cross_attention_dim: int = 1024
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    # down block i has cross-attention iff i < len-1 ; up block i iff i > 0
    timestep: int = 999

    @property
    def time_embed_dim(self):
        return self.block_out_channels[0] * 4


SD_TURBO_VAE = VAEArch()
SD_TURBO_UNET = UNetArch()

# Same topology, small widths: CPU tests, emulator tests, quick GPU checks.
TINY_VAE = VAEArch(block_out_channels=(32, 64, 128, 128), norm_num_groups=8)
TINY_UNET = UNetArch(block_out_channels=(64, 128, 128, 128), num_heads=(1, 2, 2, 2),
                     cross_attention_dim=96, norm_num_groups=16)
