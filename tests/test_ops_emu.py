"""Kernel-logic parity on the CPU: the HIP kernel sources compiled for the wave-level emulator
(tests/emu/) vs plain PyTorch fp32 statements of each op.  Small shapes; every tile config, gather mode
and epilogue of the implicit-GEMM kernel is exercised."""
import pytest
import torch

import opcheck as oc
from img2img_turbo_amd import _capi as K
from img2img_turbo_amd import ops as O

DTYPES = [torch.float32, torch.bfloat16, torch.float16]


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv3x3_basic(emu_lib, dtype):
    oc.check_conv(emu_lib, "cpu", dtype)


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5])
def test_conv3x3_tiles(emu_lib, tile):
    oc.check_conv(emu_lib, "cpu", torch.bfloat16, cout=40, tile=tile, n=1, h=9, w=11)


def test_conv_gn_silu_prologue(emu_lib):
    oc.check_conv(emu_lib, "cpu", torch.float32, gn=True, act=1, n=3, h=6, w=6, cin=16)   # tiles span images
    oc.check_conv(emu_lib, "cpu", torch.bfloat16, gn=True, act=1)
    oc.check_conv(emu_lib, "cpu", torch.float32, gn=True, act=0, ks=1, pad=0)              # transformer entry GN


def test_conv_stride2_variants(emu_lib):
    oc.check_conv(emu_lib, "cpu", torch.float32, stride=2, pad=1, h=12, w=8)                  # UNet downsample
    oc.check_conv(emu_lib, "cpu", torch.float32, stride=2, asym_pad=True, h=12, w=8)          # VAE F.pad(0,1,0,1)


def test_conv_upsample_gather(emu_lib):
    oc.check_conv(emu_lib, "cpu", torch.float32, ups=1, h=5, w=6)
    oc.check_conv(emu_lib, "cpu", torch.bfloat16, ups=1, h=5, w=6)


def test_conv_concat_two_sources(emu_lib):
    oc.check_conv(emu_lib, "cpu", torch.float32, cin=16, cin2=24, gn=True, act=1, groups=5)   # group straddles the seam
    oc.check_conv(emu_lib, "cpu", torch.bfloat16, cin=16, cin2=8)


def test_conv_small_channels(emu_lib):
    oc.check_conv(emu_lib, "cpu", torch.float32, cin=3, cout=32)      # VAE conv_in (padded to 8)
    oc.check_conv(emu_lib, "cpu", torch.float32, cin=32, cout=3)      # VAE conv_out
    oc.check_conv(emu_lib, "cpu", torch.bfloat16, cin=4, cout=4, ks=1, pad=0)


def test_conv1x1_residual_alpha(emu_lib):
    oc.check_conv(emu_lib, "cpu", torch.float32, ks=1, pad=0, bias=False, res=True, alpha=0.4)  # skip conv + add
    oc.check_conv(emu_lib, "cpu", torch.float32, res=True)                                        # resnet conv2


@pytest.mark.parametrize("dtype", DTYPES)
def test_geglu(emu_lib, dtype):
    oc.check_geglu(emu_lib, "cpu", dtype)


def test_geglu_tiles(emu_lib):
    for tile in (1, 2, 3, 5):
        oc.check_geglu(emu_lib, "cpu", torch.float32, tile=tile, cff=128)


@pytest.mark.parametrize("dtype", DTYPES)
def test_bgemm(emu_lib, dtype):
    oc.check_bgemm(emu_lib, "cpu", dtype)
    oc.check_bgemm(emu_lib, "cpu", dtype, out_f32=0)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gn_stats(emu_lib, dtype):
    oc.check_gn_stats(emu_lib, "cpu", dtype)
    oc.check_gn_stats(emu_lib, "cpu", dtype, c0=40, c1=24, groups=4, h=5, w=5, nparts=2)      # two sources, cc=8
    oc.check_gn_stats(emu_lib, "cpu", dtype, c0=320, groups=32, h=3, w=3, nparts=1)           # cc=40: lpp=32, jn=2


def test_gn_stats_wide(emu_lib):
    oc.check_gn_stats(emu_lib, "cpu", torch.float32, c0=1280, c1=640, groups=32, h=2, w=2, nparts=1, n=1)  # 1920 ch, cpg 60


@pytest.mark.parametrize("dtype", DTYPES)
def test_layernorm(emu_lib, dtype):
    oc.check_layernorm(emu_lib, "cpu", dtype)
    oc.check_layernorm(emu_lib, "cpu", dtype, c=1280, rows=5)
    oc.check_layernorm(emu_lib, "cpu", dtype, c=64, rows=4)
    oc.check_layernorm(emu_lib, "cpu", dtype, c=640, rows=7)          # two chunks per lane: a wave takes two rows (one ragged)
    oc.check_layernorm(emu_lib, "cpu", dtype, c=1024, rows=3, seed=2)
    oc.check_layernorm(emu_lib, "cpu", dtype, c=320, rows=33, seed=3)  # four rows per wave, 16 per workgroup: a ragged last workgroup


@pytest.mark.parametrize("dtype", DTYPES)
def test_softmax(emu_lib, dtype):
    oc.check_softmax(emu_lib, "cpu", dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention(emu_lib, dtype):
    oc.check_attention(emu_lib, "cpu", dtype)
    oc.check_attention(emu_lib, "cpu", dtype, tq=64, tk=128, batch=1, heads=1, spike=True)


@pytest.mark.parametrize("dtype", DTYPES)
def test_boundary_and_latent_ops(emu_lib, dtype):
    oc.check_boundary(emu_lib, "cpu", dtype)
    oc.check_latent_ops(emu_lib, "cpu", dtype)


# ---- halo-tiled 3x3 kernel (conv3x3.hip): tile 10 forces it, so a silent fall-back to the generic kernel fails ----
@pytest.mark.parametrize("dtype", DTYPES)
def test_halo_conv_basic(emu_lib, dtype):
    oc.check_conv(emu_lib, "cpu", dtype, n=2, cin=64, cout=48, h=16, w=32, tile=10)


def test_halo_conv_gn_residual_partial_tiles(emu_lib):
    # plane 12x20: partial tiles in both directions; two slabs (f32: 4 slabs) exercise the halo prefetch ring
    oc.check_conv(emu_lib, "cpu", torch.bfloat16, n=2, cin=128, cout=128, h=12, w=20, gn=True, act=1, groups=8, res=True, tile=10)
    oc.check_conv(emu_lib, "cpu", torch.float32, n=1, cin=128, cout=32, h=12, w=20, gn=True, act=1, groups=8, res=True, tile=10)


def test_halo_conv_upsample_concat_smallN(emu_lib):
    oc.check_conv(emu_lib, "cpu", torch.float32, n=1, cin=32, cout=64, h=8, w=8, ups=1, tile=10)              # Upsample2D gather
    oc.check_conv(emu_lib, "cpu", torch.bfloat16, n=1, cin=64, cin2=128, cout=64, h=8, w=16, gn=True, act=1, groups=8, tile=10)  # concat
    oc.check_conv(emu_lib, "cpu", torch.float32, n=1, cin=64, cout=3, h=8, w=16, gn=True, act=1, groups=8, tile=10)   # conv_out (BN=16)


@pytest.mark.parametrize("cfg", [11, 12, 13, 14, 15, 16, 17, 18, 19, 31, 32, 33, 34])
def test_halo_conv_every_tile_config(emu_lib, cfg):
    """Each conv3x3.hip tile configuration forced by id: ragged plane (partial tiles both ways), two slabs
    (register-parked halo hand-over), GN+SiLU prologue, residual, N not a multiple of the channel tile."""
    cout = 3 if cfg == 16 else 72
    oc.check_conv(emu_lib, "cpu", torch.bfloat16, n=2, cin=128, cout=cout, h=20, w=24, gn=True, act=1, groups=8,
                  res=(cfg != 16), tile=cfg)


@pytest.mark.parametrize("cfg", [11, 13, 31, 32])
def test_halo_conv_f32_three_slabs_concat_upsample(emu_lib, cfg):
    oc.check_conv(emu_lib, "cpu", torch.float32, n=1, cin=32, cin2=64, cout=40, h=18, w=16, gn=True, act=1, groups=8, tile=cfg)
    oc.check_conv(emu_lib, "cpu", torch.float32, n=1, cin=64, cout=136, h=9, w=8, ups=1, tile=cfg)


# ---- LDS-DMA implicit GEMM (gemm_dma.hip): tile 20 = auto, 21..25 force a configuration ----
@pytest.mark.parametrize("cfg", [21, 22, 23, 24, 25])
def test_dma_igemm_every_tile_config(emu_lib, cfg):
    """Linear with a K tail (K % 64 != 0), ragged M and N, bias + residual; then a 3x3 stride-2 gather with padding."""
    oc.check_conv(emu_lib, "cpu", torch.bfloat16, n=1, cin=88, cout=72, h=9, w=23, ks=1, pad=0, res=True, alpha=0.7, tile=cfg)
    oc.check_conv(emu_lib, "cpu", torch.float32, n=2, cin=64, cout=40, h=12, w=10, stride=2, pad=1, tile=cfg)


def test_dma_igemm_gathers(emu_lib):
    oc.check_conv(emu_lib, "cpu", torch.float32, n=2, cin=32, cout=48, h=6, w=6, tile=20)                       # 3x3 s1 on a small plane
    oc.check_conv(emu_lib, "cpu", torch.bfloat16, n=1, cin=64, cout=32, h=12, w=8, stride=2, asym_pad=True, tile=20)  # VAE F.pad(0,1,0,1)
    oc.check_conv(emu_lib, "cpu", torch.float32, n=1, cin=32, cout=32, h=5, w=6, ups=1, tile=20)                # Upsample2D gather
    oc.check_conv(emu_lib, "cpu", torch.bfloat16, n=1, cin=64, cin2=128, cout=64, h=8, w=8, tile=20)              # 3x3 concat
    oc.check_conv(emu_lib, "cpu", torch.float32, n=1, cin=32, cin2=64, cout=48, h=4, w=8, ks=1, pad=0, tile=20)  # 1x1 concat (shortcut)


@pytest.mark.parametrize("dtype", DTYPES)
def test_dma_igemm_geglu_bgemm(emu_lib, dtype):
    oc.check_geglu(emu_lib, "cpu", dtype, tile=20)
    oc.check_geglu(emu_lib, "cpu", dtype, tile=21, cff=128, rows=200)
    oc.check_bgemm(emu_lib, "cpu", dtype, tile=20)
    oc.check_bgemm(emu_lib, "cpu", dtype, out_f32=0, tile=22)


def test_dma_igemm_small_tile(emu_lib):
    """Tile 26 (64 x 32, four waves along the rows): the batch-1 linears without K slices -- ragged rows / columns, residual, the
    two-source 1x1 (conv_shortcut of a concat), a K tail, fp32."""
    oc.check_conv(emu_lib, "cpu", torch.bfloat16, n=1, cin=320, cout=328, h=16, w=17, ks=1, pad=0, res=True, tile=26)
    oc.check_conv(emu_lib, "cpu", torch.float16, n=2, cin=64, cin2=128, cout=72, h=7, w=9, ks=1, pad=0, tile=26)
    oc.check_conv(emu_lib, "cpu", torch.bfloat16, n=1, cin=88, cout=40, h=8, w=9, ks=1, pad=0, res=True, alpha=0.7, tile=26)
    oc.check_conv(emu_lib, "cpu", torch.float32, n=1, cin=96, cout=56, h=8, w=8, ks=1, pad=0, tile=26)
    oc.check_conv_gn_part(emu_lib, "cpu", torch.bfloat16, n=2, cin=64, cout=128, h=16, w=16, groups=32, tile=26, ks=1)


@pytest.mark.parametrize("wgs", [1, 2, 3, 5])
def test_dma_igemm_persistent_stream(emu_lib, wgs, monkeypatch):
    """Persistent launch (fewer workgroups than tiles; I2I_PERSIST_WGS is the test hook): the K-step stream crosses
    tile borders with every ring phase (nk = 1, 2, 3, 5 steps per tile), ragged M / N, gathers, GEGLU, residuals."""
    monkeypatch.setenv("I2I_PERSIST_WGS", str(wgs))
    oc.check_conv(emu_lib, "cpu", torch.bfloat16, n=2, cin=64, cout=200, h=9, w=23, ks=1, pad=0, res=True, alpha=0.7, tile=24)   # nk = 1
    oc.check_conv(emu_lib, "cpu", torch.bfloat16, n=1, cin=88, cout=72, h=19, w=23, ks=1, pad=0, res=True, tile=22)              # nk = 2, K tail
    oc.check_conv(emu_lib, "cpu", torch.float32, n=2, cin=96, cout=136, h=12, w=10, ks=1, pad=0, tile=23)                        # nk = 3
    oc.check_conv(emu_lib, "cpu", torch.float16, n=2, cin=320, cout=72, h=16, w=20, ks=1, pad=0, tile=25)                        # nk = 5
    oc.check_conv(emu_lib, "cpu", torch.float32, n=2, cin=32, cout=40, h=12, w=10, stride=2, pad=1, tile=24)                     # 3x3 gather
    oc.check_conv(emu_lib, "cpu", torch.bfloat16, n=2, cin=3, cout=40, h=9, w=12, tile=24)                                       # narrow input
    oc.check_geglu(emu_lib, "cpu", torch.bfloat16, tile=24, cff=128, rows=200)


@pytest.mark.parametrize("splitk", [2, 3, 5])
def test_dma_igemm_splitk(emu_lib, splitk):
    """Weight-streaming shape in miniature: few rows, long K (3x3 over 4 slabs = 36 steps), split over grid z."""
    oc.check_conv(emu_lib, "cpu", torch.bfloat16, n=2, cin=128, cout=72, h=4, w=4, res=True, tile=23, splitk=splitk)
    oc.check_conv(emu_lib, "cpu", torch.float32, n=1, cin=96, cout=40, h=4, w=4, tile=24, splitk=splitk)


@pytest.mark.parametrize("cfg,dtype", [(13, torch.bfloat16), (17, torch.float32), (12, torch.bfloat16), (14, torch.float16), (15, torch.bfloat16)])
def test_conv_epilogue_groupnorm_partials(emu_lib, cfg, dtype):
    """gn_part: per-tile partial sums from the conv epilogue + finalize_only == statistics of the stored tensor."""
    oc.check_conv_gn_part(emu_lib, "cpu", dtype, n=2, cin=64, cout=64, h=20, w=24, groups=8, tile=cfg)      # ragged tiles
    oc.check_conv_gn_part(emu_lib, "cpu", dtype, n=1, cin=64, cout=192, h=16, w=16, groups=12, tile=cfg, res=False)  # cpg 16, 2 n-tiles


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_subpixel_upsampler_groupnorm_partials(emu_lib, dtype):
    """Upsample2D in sub-pixel form emits the next GroupNorm's partial sums: four parity workgroups per source tile,
    ragged source plane, two channel tiles."""
    oc.check_conv_gn_part(emu_lib, "cpu", dtype, n=2, cin=64, cout=64, h=12, w=20, groups=8, subpix=True)
    oc.check_conv_gn_part(emu_lib, "cpu", dtype, n=1, cin=64, cout=192, h=16, w=16, groups=12, subpix=True, res=False)


@pytest.mark.parametrize("dtype", DTYPES)
def test_subpixel_upsample_conv(emu_lib, dtype):
    """Upsample2D in sub-pixel form (4 parity 2x2 convs over the source plane, tap weights pre-summed) against
    F.conv2d on the nearest-upsampled input: ragged source planes, 2-3 slabs, bias + residual, N not a tile multiple."""
    oc.check_conv(emu_lib, "cpu", dtype, n=2, cin=128, cout=72, h=9, w=20, ups=1, res=True, subpix=True)
    oc.check_conv(emu_lib, "cpu", dtype, n=1, cin=64, cout=136, h=16, w=16, ups=1, subpix=True)     # 16-row tiles, 2 n-tiles


def test_subpixel_weights_identity():
    """packer.subpixel_weights: conv3x3(upsample2x(x)) == interleave of the four parity 2x2 convs (pure torch)."""
    import torch.nn.functional as F
    from img2img_turbo_amd.packer import subpixel_weights
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 5, 6, 7, generator=g, dtype=torch.float64)
    w = torch.randn(4, 5, 3, 3, generator=g, dtype=torch.float64)
    ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, padding=1)
    ws = subpixel_weights(w.float()).double().reshape(2, 2, 4, 2, 2, 5)          # [a][b][o][r][c][i]
    out = torch.zeros_like(ref)
    xp = F.pad(x, (1, 1, 1, 1))
    for a in range(2):
        for b in range(2):
            k = ws[a, b].permute(0, 3, 1, 2)                                        # [o][i][r][c]
            y = F.conv2d(xp[:, :, a:a + 7, b:b + 8], k)                             # source window starts at (y+a-1, x+b-1)
            out[:, :, a::2, b::2] = y
    assert (out - ref).abs().max() < 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_gn_apply_one_and_two_sources(emu_lib, dtype):
    oc.check_gn_apply(emu_lib, "cpu", dtype, n=2, c=64, h=6, w=7)
    oc.check_gn_apply(emu_lib, "cpu", dtype, n=2, c=64, c1=24, h=5, w=9, seed=1)        # torch.cat([x, skip], 1) in front of the norm
    oc.check_gn_apply(emu_lib, "cpu", dtype, n=1, c=8, c1=128, h=3, w=3, act=0, seed=2)


def test_gn_stats_single_launch_shapes(emu_lib):
    """gn_stats_small_kernel: channel ranges of a block that straddle the concat seam, cpg not a multiple of 8."""
    oc.check_gn_stats(emu_lib, "cpu", torch.bfloat16, c0=640, c1=320, groups=32, h=4, w=4, nparts=1, n=2)    # cpg 30, seam inside a block
    oc.check_gn_stats(emu_lib, "cpu", torch.float32, c0=1280, c1=1280, groups=32, h=2, w=2, nparts=1, n=1)   # cpg 80
    oc.check_gn_stats(emu_lib, "cpu", torch.float32, c0=128, groups=32, h=16, w=8, nparts=2, n=2)            # cpg 4 (VAE)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("qf", ["1", "2"])
def test_attention_dma_ring_and_tails(emu_lib, dtype, qf, monkeypatch):
    """attention_dma_kernel: >3 key tiles (ring wrap-around), query tails across workgroups, tk a multiple of 64,
    tk with a partially valid last V^T chunk (padding poisoned with NaN by the checker)."""
    monkeypatch.setenv("I2I_ATT_QF", qf)                   # 64-query (small grids) and 128-query workgroups
    oc.check_attention(emu_lib, "cpu", dtype, batch=1, heads=2, tq=130, tk=264)
    oc.check_attention(emu_lib, "cpu", dtype, batch=2, heads=1, tq=33, tk=64)
    oc.check_attention(emu_lib, "cpu", dtype, batch=1, heads=1, tq=128, tk=325, spike=True)
    # keys split over workgroups + the merge launch (the T = 4096 self-attention of a batch-1 forward): 5 key tiles over 2 splits (3 + 2),
    # 6 tiles over 4 (2 + 2 + 2 + one EMPTY split), a partially valid last V^T chunk in the last split, the late spike
    oc.check_attention(emu_lib, "cpu", dtype, batch=1, heads=2, tq=130, tk=264, ksplit=2)
    oc.check_attention(emu_lib, "cpu", dtype, batch=2, heads=1, tq=33, tk=325, ksplit=4, spike=True)
    oc.check_attention(emu_lib, "cpu", dtype, batch=1, heads=1, tq=70, tk=64, ksplit=3)


@pytest.mark.parametrize("dtype", DTYPES)
def test_softmax_row_kernel(emu_lib, dtype):
    oc.check_softmax(emu_lib, "cpu", dtype, rows=9, cols=256, ldp=256)        # register-resident path
    oc.check_softmax(emu_lib, "cpu", dtype, rows=5, cols=1024, ldp=1032)      # zero padding past cols


@pytest.mark.parametrize("dtype", DTYPES)
def test_u8_boundary(emu_lib, dtype):
    oc.check_u8_boundary(emu_lib, "cpu", dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_attention_wide_head(emu_lib, dtype):
    """attention_wide_kernel (one head of 512: the VAE mid-block attention): several key tiles of 32 (double buffer
    wrap-around), query tails, a key tail that is not a chunk multiple (poisoned V^T padding), late max jump."""
    oc.check_attention(emu_lib, "cpu", dtype, batch=1, heads=1, d=512, tq=70, tk=77, spike=True)
    oc.check_attention(emu_lib, "cpu", dtype, batch=2, heads=1, d=512, tq=130, tk=64)
    # keys split over workgroups + merge launch (small-batch VAE attention): 3 key tiles over 2 splits (2 + 1), over 4 (one empty
    # split: weight 0 in the merge), a late running-max jump inside the last split, two images
    oc.check_attention(emu_lib, "cpu", dtype, batch=1, heads=1, d=512, tq=70, tk=77, spike=True, ksplit=2)
    oc.check_attention(emu_lib, "cpu", dtype, batch=2, heads=1, d=512, tq=33, tk=96, ksplit=4)
    oc.check_attention(emu_lib, "cpu", dtype, batch=1, heads=1, d=512, tq=20, tk=77, ksplit=4)


@pytest.mark.parametrize("wgs", [0, 2])
def test_dma_igemm_epilogue_groupnorm_partials(emu_lib, wgs, monkeypatch):
    """Next-round feature (compiled into the emulator build only): the LDS-DMA igemm epilogue emits the
    GroupNorm partial sums of its output -- 1x1 skip conv with residual, stride-2 downsampler, two channel tiles, one tile
    per workgroup and the persistent stream."""
    monkeypatch.setenv("I2I_PERSIST_WGS", str(wgs))
    if emu_lib.igemm_gn_parts is None:
        pytest.skip("library without i2i_igemm_gn_parts")
    oc.check_conv_gn_part(emu_lib, "cpu", torch.bfloat16, n=2, cin=64, cout=128, h=16, w=16, groups=32, tile=20, ks=1)            # cpg 4, BM 64
    oc.check_conv_gn_part(emu_lib, "cpu", torch.float16, n=1, cin=64, cout=256, h=48, w=48, groups=32, tile=20, ks=1, res=False)  # BM 256, 2 n-tiles
    oc.check_conv_gn_part(emu_lib, "cpu", torch.bfloat16, n=2, cin=64, cout=64, h=32, w=32, groups=8, tile=20, stride=2)          # stride-2 gather


def test_gn_stats_large_offset_second_pass(emu_lib):
    """|mean| = 1000 sigma: the one-pass variance is noise; the flagged groups are re-read against the first-pass mean
    (single-launch kernel, partial + finalize pair, and finalize over conv-epilogue style one-pass partials)."""
    oc.check_gn_stats_offset(emu_lib, "cpu", torch.float32, h=24, w=20)                             # single launch (small tensor)
    oc.check_gn_stats_offset(emu_lib, "cpu", torch.float32, h=24, w=20, finalize_only=True, nparts=7)
    oc.check_gn_stats_offset(emu_lib, "cpu", torch.float32, c=24, groups=3, h=10, w=12, mean=-40.0, std=0.05)   # odd group count, negative offset
    oc.check_gn_stats_offset(emu_lib, "cpu", torch.float16, h=24, w=20, mean=100.0, std=0.2)          # 16-bit inputs (ulp 0.0625 at 100): single launch
    oc.check_gn_stats_offset(emu_lib, "cpu", torch.bfloat16, h=24, w=20, mean=30.0, std=0.3, finalize_only=True, nparts=7)   # bf16 (ulp 0.125 at 30)
    oc.check_gn_stats_offset(emu_lib, "cpu", torch.bfloat16, c=64, groups=16, h=24, w=20, mean=100.0, std=0.3, finalize_only=True, nparts=7)   # bf16 above its flag ratio: 8-byte pieces (cpg 4)
    oc.check_gn_stats_offset(emu_lib, "cpu", torch.float16, c=40, groups=4, h=12, w=10, mean=200.0, std=0.2)                  # cpg 10 (UNet): 4-byte pieces
    oc.check_gn_stats_offset(emu_lib, "cpu", torch.float16, c=128, groups=8, h=12, w=10, mean=200.0, std=0.2)                 # cpg 16: 16-byte pieces


def test_gn_stats_sliced_single_launch(emu_lib):
    """Round 5: the single-launch statistics kernel with its pixels cut into slices (ticket counters, ABI v7): UNet shapes with
    cpg 10 / 20 / 40 incl. a two-source concat whose seam lies inside a group set, the VAE's cpg 4, a ragged last slice, the
    large-offset second pass from the last-arriving workgroup."""
    oc.check_gn_stats(emu_lib, "cpu", torch.bfloat16, n=2, c0=320, groups=32, h=16, w=16, nparts=4, sliced=True)        # cpg 10, 4 slices of 64 px
    oc.check_gn_stats(emu_lib, "cpu", torch.float32, n=1, c0=640, c1=320, groups=32, h=12, w=11, nparts=2, sliced=True)  # cpg 30, concat, ragged slice
    oc.check_gn_stats(emu_lib, "cpu", torch.float16, n=2, c0=128, groups=32, h=16, w=24, nparts=6, sliced=True)         # cpg 4 (VAE), 6 slices
    oc.check_gn_stats(emu_lib, "cpu", torch.float32, n=1, c0=64, groups=8, h=8, w=8, nparts=8, sliced=True)             # 64 px: one slice (falls back)
    oc.check_gn_stats_offset(emu_lib, "cpu", torch.float32, c=64, groups=8, h=24, w=20, nparts=4, sliced=True)         # the flagged groups' second pass, from the last slice


def test_gn_finalize_many_parts(emu_lib):
    """Finalize with hundreds / thousands of parts per image (what 512x512 conv epilogues hand over): the launcher
    switches to 2 and then 1 group per block."""
    oc.check_gn_stats(emu_lib, "cpu", torch.float32, c0=64, groups=8, h=24, w=20, nparts=300, n=2)     # gpb = 2
    oc.check_gn_stats(emu_lib, "cpu", torch.bfloat16, c0=128, groups=32, h=40, w=32, nparts=1100, n=1)  # gpb = 1
    oc.check_gn_stats(emu_lib, "cpu", torch.float32, c0=24, groups=3, h=24, w=20, nparts=300, n=1)     # odd group count: stays at 3 per block


def test_dma_igemm_narrow_input_conv(emu_lib):
    """VAE conv_in shape class: 3 -> 8 padded input channels, several taps per K step (per-chunk tap decode)."""
    oc.check_conv(emu_lib, "cpu", torch.bfloat16, n=2, cin=3, cout=40, h=9, w=12, tile=20)
    oc.check_conv(emu_lib, "cpu", torch.float32, n=1, cin=4, cout=32, h=6, w=7, tile=20)        # latent conv_in (4 -> 8 padded)
    oc.check_conv(emu_lib, "cpu", torch.float16, n=1, cin=16, cout=24, h=5, w=5, stride=2, pad=1, tile=22)


def test_conv_explicit_upsample_size(emu_lib):
    """F.interpolate(size=...) with a size that is not 2x (UNet forward_upsample_size path, odd latent sizes): the
    gather kernels use ATen's nearest rule min(floor(i*in/out), in-1)."""
    import torch.nn.functional as F
    from img2img_turbo_amd import ops as O
    g = torch.Generator().manual_seed(0)
    for dtype, cin, tile in ((torch.float32, 32, 20), (torch.bfloat16, 64, 0), (torch.float32, 8, 1)):
        n, h, w, cout, uh, uw = 2, 9, 5, 24, 17, 9
        x = torch.randn(n, cin, h, w, generator=g)
        wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
        b = torch.randn(cout, generator=g) * 0.1
        xq, wq = x.to(dtype).float(), wt.to(dtype).float()
        ref = F.conv2d(F.interpolate(xq, size=(uh, uw), mode="nearest"), wq, b, padding=1)
        x0 = oc.nhwc(x, dtype)
        wp = oc.pack_conv_weight(wt, dtype)
        out = torch.full((n, uh, uw, cout), float("nan"), dtype=dtype)
        opcode, p = O.conv(x0, wp, out, nimg=n, hin=h, win=w, ho=uh, wo=uw, ks=3, stride=1, pad=1, ups=1, N=cout, bias=b, tile=tile, up_size=(uh, uw))
        oc.run_op(emu_lib, opcode, p, dtype, "cpu")
        got = out.float().permute(0, 3, 1, 2)
        assert oc.rel_err(got, ref) < oc.TOL[dtype]
    # halo kernel with an explicit size (wide enough plane)
    n, cin, cout, h, w, uh, uw = 1, 64, 32, 9, 9, 17, 18
    x = torch.randn(n, cin, h, w, generator=g); wt = torch.randn(cout, cin, 3, 3, generator=g) / 24
    ref = F.conv2d(F.interpolate(x.to(torch.bfloat16).float(), size=(uh, uw), mode="nearest"), wt.to(torch.bfloat16).float(), None, padding=1)
    out = torch.full((n, uh, uw, cout), float("nan"), dtype=torch.bfloat16)
    x0, wp = oc.nhwc(x, torch.bfloat16), oc.pack_conv_weight(wt, torch.bfloat16)     # keep alive: ops hold raw pointers
    opcode, p = O.conv(x0, wp, out, nimg=n, hin=h, win=w, ho=uh, wo=uw, ks=3,
                       stride=1, pad=1, ups=1, N=cout, tile=10, up_size=(uh, uw))
    oc.run_op(emu_lib, opcode, p, torch.bfloat16, "cpu")
    assert oc.rel_err(out.float().permute(0, 3, 1, 2), ref) < oc.TOL[torch.bfloat16]



def test_lanczos_resize_u8_is_bit_identical_to_pillow(emu_lib):
    """csrc/resize.hip through the C ABI vs Pillow itself (the implementation the reference's scripts call,
    src/inference_paired.py:38-41, src/inference_unpaired.py:40,53): uint8 HWC batches, down- and up-scaling, one axis only."""
    import numpy as np
    from PIL import Image
    from img2img_turbo_amd.image_ops import lanczos_resize_u8, resize_to_multiple_of_8
    rng = np.random.default_rng(1)
    for (n, h, w, oh, ow) in [(2, 37, 53, 32, 48), (1, 90, 160, 64, 64), (2, 40, 24, 72, 56), (1, 33, 64, 32, 64), (1, 64, 45, 64, 40)]:
        a = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
        a[:, : h // 3] = (a[:, : h // 3] > 127) * 255
        got = lanczos_resize_u8(torch.from_numpy(a), (ow, oh), emu_lib).numpy()
        for i in range(n):
            ref = np.asarray(Image.fromarray(a[i], "RGB").resize((ow, oh), Image.LANCZOS))
            assert np.array_equal(got[i], ref), (n, h, w, oh, ow)
    a = rng.integers(0, 256, (1, 37, 53, 3), dtype=np.uint8)
    got = resize_to_multiple_of_8(torch.from_numpy(a), emu_lib).numpy()
    assert np.array_equal(got[0], np.asarray(Image.fromarray(a[0], "RGB").resize((48, 32), Image.LANCZOS)))


def test_image_prep_options_match_the_reference_transforms(emu_lib):
    """image_ops.apply_image_prep vs the reference's build_transform (src/my_utils/training_utils.py:184-215) restated with
    Pillow: Resize((S, S), LANCZOS); Resize(512, LANCZOS) + CenterCrop(512) with torchvision's size / crop arithmetic."""
    import numpy as np
    from PIL import Image
    from img2img_turbo_amd.image_ops import apply_image_prep
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, (1, 90, 140, 3), dtype=np.uint8)
    t = torch.from_numpy(a)
    pil = Image.fromarray(a[0], "RGB")
    assert np.array_equal(apply_image_prep(t, "resize_256x256", emu_lib)[0].numpy(), np.asarray(pil.resize((256, 256), Image.LANCZOS)))
    assert torch.equal(apply_image_prep(t, "no_resize", emu_lib), t)
    # shorter side (height 90) -> 512, width int(512 * 140 / 90) = 796, centre crop 512: left = round((796 - 512) / 2) = 142
    want = np.asarray(pil.resize((796, 512), Image.LANCZOS))[:, 142:142 + 512]
    got = apply_image_prep(t, "resized_crop_512", emu_lib)[0].numpy()
    assert got.shape == (512, 512, 3) and np.array_equal(got, want)
    with pytest.raises(ValueError):
        apply_image_prep(t, "resize_286_randomcrop_256x256_hflip", emu_lib)


# ---------------------------------------------------------------- wide-tile conv (conv3x3_w32.hip, tile ids 41, 42)
W32_TILES = [41, 42]


@pytest.mark.parametrize("cfg", W32_TILES)
def test_w32_conv_every_tile_config(emu_lib, cfg):
    """32x32x16-MFMA wide-tile conv: GroupNorm+SiLU staged in the MFMA shadow, residual (staged through LDS in the row
    layout), ragged tiles in both plane directions, two slabs (the halo stored after P_8), ragged channel tile (N = 136 on
    BN = 128 / 256), channel tile by XCD with runs of unequal length (18 spatial tiles over 4 XCD groups)."""
    oc.check_conv(emu_lib, "cpu", torch.bfloat16, n=3, cin=128, cout=256, h=20, w=72, gn=True, act=1, groups=8, res=True, tile=cfg)
    oc.check_conv(emu_lib, "cpu", torch.float16, n=2, cin=64, cin2=64, cout=136, h=33, w=65, tile=cfg, seed=3)


def test_w32_conv_three_slabs_one_slab_and_route(emu_lib):
    """Three slabs over a concat seam, one-slab tiles (every slab is a tile's first and last), alpha; the route query names
    the kernel the dispatcher picks."""
    oc.check_conv(emu_lib, "cpu", torch.bfloat16, n=1, cin=64, cin2=128, cout=128, h=16, w=32, gn=True, act=1, groups=8, tile=42)
    oc.check_conv(emu_lib, "cpu", torch.float16, n=1, cin=64, cout=128, h=9, w=32, res=True, alpha=0.5, tile=41)
    oc.check_conv(emu_lib, "cpu", torch.bfloat16, n=2, cin=64, cout=128, h=20, w=40, res=True, tile=42, seed=5)     # one slab = the last slab: residual rows ride in its load slots
    x = torch.zeros(8, 128, 128, 128, dtype=torch.bfloat16)
    w = torch.zeros(128, 9 * 128, dtype=torch.bfloat16)
    out = torch.zeros(8, 128, 128, 128, dtype=torch.bfloat16)
    kw = dict(hin=128, win=128, ho=128, wo=128, ks=3, pad=1)
    _, p = O.conv(x, w, out, nimg=8, **kw)
    assert emu_lib.igemm_route(p, K.BF16) == "conv3x3_w32_kernel"          # 8 x 8 x 4 x 1 = 256 tiles of 16 x 32 x 128
    _, p1 = O.conv(x[:1], w, out[:1], nimg=1, **kw)
    assert emu_lib.igemm_route(p1, K.BF16) == "conv3x3_halo_kernel"         # batch 1: too few wide tiles to fill the chip
    _, p2 = O.conv(x, w, out, nimg=8, tile=20, **kw)
    assert emu_lib.igemm_route(p2, K.BF16) == "igemm_dma_kernel"
    _, p3 = O.conv(x.float(), w.float(), out.float(), nimg=8, **kw)
    assert emu_lib.igemm_route(p3, K.F32) == "conv3x3_halo_kernel"          # exact-f32 parity mode stays on the halo kernel


@pytest.mark.parametrize("cfg", W32_TILES)
def test_w32_subpixel_upsample_conv(emu_lib, cfg):
    """Sub-pixel Upsample2D form on the wide tiles: four parity workgroups per source tile (halo origin shifted by the parity),
    four taps per slab on a two-deep weight ring, outputs scattered to (2y+a, 2x+b); ragged source tiles in both directions,
    two slabs / one slab, ragged channel tile, GroupNorm partial sums with one slot per tile and parity; the route query."""
    oc.check_conv(emu_lib, "cpu", torch.bfloat16, n=2, cin=128, cout=256, h=20, w=40, ups=1, subpix=True, tile=cfg)
    oc.check_conv(emu_lib, "cpu", torch.float16, n=1, cin=64, cout=136, h=9, w=33, ups=1, subpix=True, tile=cfg, seed=4)
    oc.check_conv_gn_part(emu_lib, "cpu", torch.bfloat16, n=1, cin=64, cout=256, h=12, w=40, groups=32, subpix=True, tile=cfg, res=False)
    # the decoder's skip conv folded in as a second contraction (k2_a): 2 slabs and 1 slab of skip channels, ragged tiles, alpha
    oc.check_conv(emu_lib, "cpu", torch.bfloat16, n=2, cin=64, cout=256, h=20, w=40, ups=1, subpix=True, tile=cfg, k2c=128, seed=7)
    oc.check_conv(emu_lib, "cpu", torch.float16, n=1, cin=128, cout=136, h=9, w=33, ups=1, subpix=True, tile=cfg, k2c=64, alpha=0.5, seed=8)
    oc.check_conv(emu_lib, "cpu", torch.bfloat16, n=1, cin=64, cout=128, h=8, w=32, ups=1, subpix=True, tile=cfg, k2c=192, seed=9)     # 3 slabs: ring slot reuse
    x = torch.zeros(8, 64, 64, 256, dtype=torch.bfloat16)
    w = torch.zeros(4 * 256, 4 * 256, dtype=torch.bfloat16)
    out = torch.zeros(8, 128, 128, 256, dtype=torch.bfloat16)
    _, p = O.conv(x, w, out, nimg=8, hin=64, win=64, ho=128, wo=128, ks=3, pad=1, ups=1, N=256, subpix=1)
    assert emu_lib.igemm_route(p, K.BF16) == "conv3x3_w32_kernel<SUBPIX>"
    _, p1 = O.conv(x[:1, :16, :16], w, out[:1, :32, :32], nimg=1, hin=16, win=16, ho=32, wo=32, ks=3, pad=1, ups=1, N=256, subpix=1)
    assert emu_lib.igemm_route(p1, K.BF16) == "conv3x3_halo_kernel<SUBPIX>"      # source plane narrower than a 32-wide tile
    sk, w2 = torch.zeros(1, 32, 32, 128, dtype=torch.bfloat16), torch.zeros(256, 128, dtype=torch.bfloat16)
    op = O.conv(x[:1, :16, :16].contiguous(), w, out[:1, :32, :32].contiguous(), nimg=1, hin=16, win=16, ho=32, wo=32, ks=3, pad=1, ups=1, N=256, subpix=1, k2=(sk, w2, 128))
    with pytest.raises(Exception, match="second contraction"):      # only the wide-tile sub-pixel form takes k2_a: loud, not silent
        oc.run_op(emu_lib, op[0], op[1], torch.bfloat16, "cpu")


@pytest.mark.parametrize("cfg", [41, 42])
def test_w32_conv_epilogue_groupnorm_partials(emu_lib, cfg):
    """The epilogue's GroupNorm partial sums of the STORED output (v_dot2c per channel quad; one slot per tile and group),
    finished by gn_stats."""
    oc.check_conv_gn_part(emu_lib, "cpu", torch.bfloat16, n=2, cin=64, cout=256, h=16, w=64, groups=32, tile=cfg)      # cpg 8
    oc.check_conv_gn_part(emu_lib, "cpu", torch.float16, n=1, cin=64, cout=128, h=20, w=40, groups=32, tile=cfg, res=False)   # cpg 4, ragged tiles
    oc.check_conv_gn_part(emu_lib, "cpu", torch.bfloat16, n=1, cin=64, cout=512, h=8, w=32, groups=32, tile=cfg)       # cpg 16


@pytest.mark.parametrize("cfg", [41, 42])
def test_w32_conv_second_contraction_plain_form(emu_lib, cfg):
    """A ResnetBlock2D's `conv_shortcut(input) + conv2(silu(norm2(h)))` in ONE launch: the 1x1 shortcut over the raw block input
    as the second contraction (k2_a) of the wide-tile 3x3 conv, GroupNorm + SiLU on the first operand only; ragged tiles, one and
    three slabs of the second operand, and the next norm's partial sums from the same epilogue."""
    oc.check_conv(emu_lib, "cpu", torch.bfloat16, n=1, cin=128, cout=256, h=18, w=40, gn=True, act=1, tile=cfg, k2c=64)
    oc.check_conv(emu_lib, "cpu", torch.float16, n=2, cin=64, cout=128, h=9, w=33, gn=True, act=1, tile=cfg, k2c=192, alpha=0.5)
    oc.check_conv(emu_lib, "cpu", torch.bfloat16, n=1, cin=64, cout=128, h=16, w=32, tile=cfg, k2c=128, bias=False)


# ---------------------------------------------------------------- wide GEMM (gemm_w32.hip, tile ids 51..54)
@pytest.mark.parametrize("cfg", [51, 52, 53, 54, 55, 56])
def test_gemm_w32_tiles(emu_lib, cfg):
    """nn.Linear / 1x1 conv on 32x32x16 MFMA: ragged row and column tiles, residual + bias + alpha, K of 1 / 2 / 5 stages
    (shorter than, equal to and longer than the ring), two channel-concatenated sources, both 16-bit types."""
    bn = 160 if cfg in (51, 52, 55) else 128
    oc.check_conv(emu_lib, "cpu", torch.bfloat16, n=1, cin=320, cout=2 * bn + 40, h=9, w=37, ks=1, pad=0, res=True, alpha=0.7, tile=cfg)    # 5 stages, ragged M and N
    oc.check_conv(emu_lib, "cpu", torch.float16, n=2, cin=64, cout=bn, h=8, w=16, ks=1, pad=0, tile=cfg)                                   # 1 stage, exact tiles
    oc.check_conv(emu_lib, "cpu", torch.bfloat16, n=1, cin=64, cin2=128, cout=bn, h=5, w=13, ks=1, pad=0, res=True, bias=False, tile=cfg)  # [c0 | c1], source switch at stage 1
    oc.check_conv(emu_lib, "cpu", torch.bfloat16, n=1, cin=128, cout=96, h=7, w=20, ks=1, pad=0, tile=cfg)                                 # 2 stages, one ragged column tile


@pytest.mark.parametrize("cfg", [51, 52, 53, 54, 55, 56])
def test_gemm_w32_geglu(emu_lib, cfg):
    oc.check_geglu(emu_lib, "cpu", torch.bfloat16, tile=cfg, cff=320, rows=300, cin=128)       # N = 640 packed columns
    oc.check_geglu(emu_lib, "cpu", torch.float16, tile=cfg, cff=64, rows=70, cin=64)           # N = 128: one (ragged for 160) column tile


@pytest.mark.parametrize("cfg", [51, 52, 53, 54])
def test_gemm_w32_conv3x3_gather(emu_lib, cfg):
    """3x3 convolutions as the im2col view of the wide GEMM's A operand (gemm_w32.hip GATHER): the VAE downsamplers'
    F.pad(0,1,0,1) + stride 2, the UNet downsamplers' stride 2 / pad 1, a stride-1 pad-1 conv; one and two 64-channel stages
    per tap, ragged row and column tiles, tiles that span images, residual / alpha, both 16-bit types."""
    bn = 160 if cfg in (51, 52) else 128
    oc.check_conv(emu_lib, "cpu", torch.bfloat16, n=2, cin=64, cout=bn, h=12, w=10, stride=2, asym_pad=True, tile=cfg)             # VAE Downsample2D
    oc.check_conv(emu_lib, "cpu", torch.float16, n=1, cin=128, cout=bn + 40, h=13, w=9, stride=2, pad=1, res=True, alpha=0.6, tile=cfg)   # odd plane, 2 stages / tap
    oc.check_conv(emu_lib, "cpu", torch.bfloat16, n=3, cin=64, cout=96, h=5, w=7, stride=1, pad=1, bias=False, tile=cfg)           # stride 1, tiles span images


@pytest.mark.parametrize("cfg", [51, 52, 53, 54])
def test_gemm_w32_splitk(emu_lib, cfg):
    """Split-K slices of the wide GEMM (grid y) + the shared reduce / epilogue launch: the UNet's small-plane 3x3 convolutions
    (18 stages over 3 slices, over 4 slices of 5 + 5 + 5 + 3), a slice count that leaves the last slice empty (9 stages over
    4 slices of 3), a plain 1x1 with residual and alpha; ragged tiles."""
    bn = 160 if cfg in (51, 52) else 128
    oc.check_conv(emu_lib, "cpu", torch.bfloat16, n=2, cin=128, cout=bn + 8, h=8, w=8, res=True, tile=cfg, splitk=3)
    oc.check_conv(emu_lib, "cpu", torch.float16, n=1, cin=128, cout=bn, h=9, w=7, stride=2, pad=1, tile=cfg, splitk=4)
    oc.check_conv(emu_lib, "cpu", torch.bfloat16, n=1, cin=64, cout=96, h=6, w=6, bias=False, tile=cfg, splitk=4)
    oc.check_conv(emu_lib, "cpu", torch.bfloat16, n=1, cin=320, cout=bn, h=5, w=9, ks=1, pad=0, res=True, alpha=0.7, tile=cfg, splitk=2)
    # a sliced launch has no statistics epilogue (the planner must not attach gn_part to it), and tile 0 keeps split-K ops on the LDS-DMA igemm
    x = torch.zeros(1, 16, 16, 128, dtype=torch.bfloat16)
    w = torch.zeros(128, 9 * 128, dtype=torch.bfloat16)
    out = torch.zeros(1, 16, 16, 128, dtype=torch.bfloat16)
    ws = torch.zeros(2 * 256 * 128)
    mk = lambda **kw: O.conv(x, w, out, nimg=1, hin=16, win=16, ho=16, wo=16, ks=3, pad=1, splitk=2, ws=ws, **kw)[1]
    assert emu_lib.igemm_route(mk(tile=cfg), K.BF16) == "gemm_w32_kernel" and emu_lib.igemm_gn_parts(mk(tile=cfg), K.BF16, 32) == 0
    assert emu_lib.igemm_route(mk(tile=20), K.BF16) == "igemm_dma_kernel"


@pytest.mark.parametrize("cfg", [53, 54])
def test_gemm_w32_gn_partials(emu_lib, cfg):
    """GroupNorm partial sums of the stored output from the wide GEMM's epilogue (128-column tiles): a plain 1x1 and a stride-2
    3x3 whose row tiles (256 / 128 rows) divide the image; finalised by gn_stats(finalize_only) against statistics of the
    stored tensor.  Two column tiles, groups of 4 / 16 / 32 channels."""
    oc.check_conv_gn_part(emu_lib, "cpu", torch.bfloat16, n=2, cin=64, cout=256, h=16, w=16, groups=8, tile=cfg, ks=1)
    oc.check_conv_gn_part(emu_lib, "cpu", torch.float16, n=1, cin=64, cout=128, h=32, w=32, groups=32, tile=cfg, ks=3, stride=2, res=False)
    oc.check_conv_gn_part(emu_lib, "cpu", torch.bfloat16, n=2, cin=128, cout=128, h=32, w=32, groups=8, tile=cfg, ks=3, stride=2)
    # what the kernel cannot do it must decline (the planner then keeps the op on the LDS-DMA igemm): 160-column tiles, ragged row tiles
    x = torch.zeros(1, 16, 16, 64, dtype=torch.bfloat16)
    w = torch.zeros(320, 64, dtype=torch.bfloat16)
    out = torch.zeros(1, 16, 16, 320, dtype=torch.bfloat16)
    assert emu_lib.igemm_gn_parts(O.conv(x, w, out, nimg=1, hin=16, win=16, ho=16, wo=16, ks=1, tile=51)[1], K.BF16, 32) == 0
    x2 = torch.zeros(1, 10, 10, 64, dtype=torch.bfloat16)
    w2 = torch.zeros(128, 64, dtype=torch.bfloat16)
    out2 = torch.zeros(1, 10, 10, 128, dtype=torch.bfloat16)
    assert emu_lib.igemm_gn_parts(O.conv(x2, w2, out2, nimg=1, hin=10, win=10, ho=10, wo=10, ks=1, tile=cfg)[1], K.BF16, 32) == 0


def test_gemm_w32_routing(emu_lib):
    """tile == 0: the dispatcher sends a chip-filling plain 16-bit GEMM to the wide kernel and everything it cannot take
    (fp32, split-K, a GroupNorm prologue, GroupNorm partial sums, odd K) to the other engines; I2I_GEMM_W32=0 switches it off."""
    import os
    x = torch.zeros(32768, 320, dtype=torch.bfloat16)
    w = torch.zeros(320, 320, dtype=torch.bfloat16)
    out = torch.zeros(32768, 320, dtype=torch.bfloat16)
    mk = lambda **kw: O.conv(x, w, out, nimg=1, hin=1, win=32768, ho=1, wo=32768, ks=1, **kw)[1]
    assert emu_lib.igemm_route(mk(), K.BF16) == "gemm_w32_kernel"
    assert emu_lib.igemm_route(mk(), K.F32) == "igemm_dma_kernel"
    assert emu_lib.igemm_route(mk(tile=20), K.BF16) == "igemm_dma_kernel"
    assert emu_lib.igemm_route(mk(splitk=2, ws=torch.zeros(8)), K.BF16) == "igemm_dma_kernel"
    assert emu_lib.igemm_gn_parts(mk(), K.BF16, 32) == 0 and emu_lib.igemm_gn_parts(mk(tile=20), K.BF16, 80) > 0
    small = O.conv(x[:512], w, out[:512], nimg=1, hin=1, win=512, ho=1, wo=512, ks=1)[1]
    assert emu_lib.igemm_route(small, K.BF16) == "igemm_dma_kernel"       # 8 tiles: not worth a wide launch
    os.environ["I2I_GEMM_W32"] = "0"
    try:
        assert emu_lib.igemm_route(mk(), K.BF16) == "igemm_dma_kernel"
    finally:
        del os.environ["I2I_GEMM_W32"]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_layernorm_folded_into_the_wide_gemm(emu_lib, dtype):
    oc.check_ln_gemm(emu_lib, "cpu", dtype)
    oc.check_ln_gemm(emu_lib, "cpu", dtype, nq=320, nv=160, rows=264)                          # to_q | to_k | to_v^T in one launch
    oc.check_ln_gemm(emu_lib, "cpu", dtype, nq=256, nv=128, rows=136, tile=54, cin=64)
    oc.check_ln_gemm(emu_lib, "cpu", dtype, nq=320, geglu=True, tile=55)
    oc.check_ln_gemm(emu_lib, "cpu", dtype, nq=256, rows=300, tile=53, offset=5.0, lora_rank=0)


def test_layernorm_fold_is_refused_off_the_wide_gemm(emu_lib):
    """An op that carries ln_cs must never run on a kernel that would ignore it."""
    x = torch.zeros(64, 64, dtype=torch.float32)
    w = torch.zeros(32, 64, dtype=torch.float32)
    out = torch.zeros(64, 32, dtype=torch.float32)
    cs = torch.zeros(32)
    opcode, p = O.conv(x, w, out, nimg=1, hin=1, win=64, ho=1, wo=64, ks=1, c0=64, lda0=64, N=32, bias=cs, ldc=32)
    p.ln_cs, p.ln_eps = cs.data_ptr(), 1e-5
    prog = K.Program()
    prog.add(opcode, K.F32, p)
    prog.freeze()
    with pytest.raises(K.I2IError):
        emu_lib.run(prog, 0)


def test_gn_stats_flag_boundaries(emu_lib):
    """Both sides of the dtype-aware second-pass ratios (csrc/norm.hip GnRefine: mu^2 / var > 2048 for fp16, 16384 for bf16): just under
    the ratio the one-pass numbers must stay within one rounding step of the storage type, just above it the second pass runs."""
    for dtype, lo, hi in ((torch.bfloat16, 118.0, 136.0), (torch.float16, 43.0, 48.0)):
        for mean in (lo, hi):
            oc.check_gn_stats_offset(emu_lib, "cpu", dtype, h=24, w=20, mean=mean, std=1.0, finalize_only=True, nparts=96)
            oc.check_gn_stats_offset(emu_lib, "cpu", dtype, h=24, w=20, mean=mean, std=1.0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_narrow_input_conv(emu_lib, dtype):
    """conv_narrow.hip (tile 60): the VAE's conv_in shape class -- 3 (padded to 8) input channels into a multiple of 128 output
    channels, 3x3 s1 p1 -- incl. ragged tiles and the GroupNorm partial sums of its output."""
    oc.check_conv(emu_lib, "cpu", dtype, n=2, cin=3, cout=128, h=16, w=64, tile=60)
    oc.check_conv(emu_lib, "cpu", dtype, n=1, cin=3, cout=256, h=9, w=40, tile=60, bias=False, seed=3)       # ragged in both directions, two channel tiles
    oc.check_conv(emu_lib, "cpu", dtype, n=1, cin=8, cout=128, h=8, w=32, tile=60, seed=4)                    # all 8 input channels live
    oc.check_conv_gn_part(emu_lib, "cpu", dtype, n=2, cin=3, cout=128, h=16, w=64, groups=32, tile=60, res=False)
    oc.check_conv_gn_part(emu_lib, "cpu", dtype, n=1, cin=3, cout=128, h=12, w=40, groups=16, tile=60, res=False, seed=2)
