"""Per-op parity on a real MI355X: the hipcc-built library through the C ABI vs PyTorch fp32 (CPU) of the
same op.  Same checks as tests/test_ops_emu.py at larger, tile-filling shapes."""
import pytest
import torch

import opcheck as oc

pytestmark = pytest.mark.gpu
DTYPES = [torch.float32, torch.bfloat16, torch.float16]


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv3x3(gpu_lib, dtype):
    oc.check_conv(gpu_lib, "cuda", dtype, n=2, cin=128, cout=128, h=40, w=48)
    oc.check_conv(gpu_lib, "cuda", dtype, n=2, cin=64, cout=256, h=33, w=17, gn=True, act=1, groups=32, res=True)


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5])
def test_conv_tiles(gpu_lib, tile):
    oc.check_conv(gpu_lib, "cuda", torch.bfloat16, n=1, cin=64, cout=72, h=24, w=24, tile=tile)
    oc.check_conv(gpu_lib, "cuda", torch.float32, n=1, cin=64, cout=72, h=24, w=24, tile=tile)


def test_conv_gathers(gpu_lib):
    for dt in (torch.float32, torch.bfloat16):
        oc.check_conv(gpu_lib, "cuda", dt, stride=2, pad=1, h=32, w=32, cin=64, cout=64)
        oc.check_conv(gpu_lib, "cuda", dt, stride=2, asym_pad=True, h=32, w=32, cin=64, cout=64)
        oc.check_conv(gpu_lib, "cuda", dt, ups=1, h=16, w=16, cin=64, cout=64)
        oc.check_conv(gpu_lib, "cuda", dt, cin=64, cin2=128, gn=True, act=1, groups=32, h=16, w=16, cout=128)
        oc.check_conv(gpu_lib, "cuda", dt, cin=3, cout=128, h=32, w=32)
        oc.check_conv(gpu_lib, "cuda", dt, cin=128, cout=3, h=32, w=32)
        oc.check_conv(gpu_lib, "cuda", dt, ks=1, pad=0, bias=False, res=True, alpha=0.4, cin=128, cout=256, h=16, w=16)
        oc.check_conv(gpu_lib, "cuda", dt, n=5, cin=32, cout=64, h=4, w=4, gn=True, act=1, groups=8)  # tiles span images


@pytest.mark.parametrize("dtype", DTYPES)
def test_geglu_bgemm(gpu_lib, dtype):
    oc.check_geglu(gpu_lib, "cuda", dtype, rows=300, cin=320, cff=1280)
    oc.check_bgemm(gpu_lib, "cuda", dtype, batch=2, heads=5, M=256, N=256, Kd=64)
    oc.check_bgemm(gpu_lib, "cuda", dtype, batch=1, heads=1, M=300, N=200, Kd=512, out_f32=0)


@pytest.mark.parametrize("dtype", DTYPES)
def test_norms(gpu_lib, dtype):
    oc.check_gn_stats(gpu_lib, "cuda", dtype, n=2, c0=128, h=64, w=64, groups=32, nparts=16)
    oc.check_gn_stats(gpu_lib, "cuda", dtype, n=2, c0=1280, c1=640, h=8, w=8, groups=32, nparts=2)
    oc.check_gn_stats(gpu_lib, "cuda", dtype, n=1, c0=320, h=16, w=16, groups=32, nparts=4)
    # sliced single launch (ticket counters, ABI v7): UNet planes at batch 1 and 8, concat input, VAE mid block
    oc.check_gn_stats(gpu_lib, "cuda", dtype, n=1, c0=320, h=64, w=64, groups=32, nparts=20, sliced=True)
    oc.check_gn_stats(gpu_lib, "cuda", dtype, n=8, c0=320, h=64, w=64, groups=32, nparts=20, sliced=True)
    oc.check_gn_stats(gpu_lib, "cuda", dtype, n=2, c0=640, c1=320, h=32, w=32, groups=32, nparts=15, sliced=True)
    oc.check_gn_stats(gpu_lib, "cuda", dtype, n=1, c0=512, h=64, w=64, groups=32, nparts=32, sliced=True)
    oc.check_layernorm(gpu_lib, "cuda", dtype, rows=1000, c=1280)
    oc.check_layernorm(gpu_lib, "cuda", dtype, rows=77, c=320)
    oc.check_layernorm(gpu_lib, "cuda", dtype, rows=4099, c=320, seed=1)    # four rows per wave, ragged last wave
    oc.check_layernorm(gpu_lib, "cuda", dtype, rows=1031, c=640, seed=2)    # two rows per wave
    oc.check_layernorm(gpu_lib, "cuda", dtype, rows=77, c=1024, seed=3)     # the text tower's width
    oc.check_softmax(gpu_lib, "cuda", dtype, rows=500, cols=1024, ldp=1024)
    oc.check_softmax(gpu_lib, "cuda", dtype, rows=500, cols=77, ldp=80)


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention(gpu_lib, dtype):
    oc.check_attention(gpu_lib, "cuda", dtype, batch=2, heads=5, tq=1024, tk=1024)
    oc.check_attention(gpu_lib, "cuda", dtype, batch=2, heads=10, tq=256, tk=77)
    oc.check_attention(gpu_lib, "cuda", dtype, batch=1, heads=1, tq=100, tk=200, spike=True)


@pytest.mark.parametrize("dtype", DTYPES)
def test_boundary_and_latent_ops(gpu_lib, dtype):
    oc.check_boundary(gpu_lib, "cuda", dtype, n=2, h=64, w=48)
    oc.check_latent_ops(gpu_lib, "cuda", dtype, n=3, h=16, w=16)


@pytest.mark.parametrize("dtype", DTYPES)
def test_halo_conv(gpu_lib, dtype):
    """conv3x3.hip forced (tile 10) at tile-filling and ragged shapes, every gather/epilogue mode."""
    oc.check_conv(gpu_lib, "cuda", dtype, n=2, cin=128, cout=128, h=64, w=64, gn=True, act=1, groups=32, res=True, tile=10)
    oc.check_conv(gpu_lib, "cuda", dtype, n=1, cin=256, cout=192, h=20, w=36, gn=True, act=1, groups=32, tile=10)
    oc.check_conv(gpu_lib, "cuda", dtype, n=2, cin=64, cout=64, h=16, w=16, ups=1, tile=10)
    oc.check_conv(gpu_lib, "cuda", dtype, n=1, cin=128, cin2=64, cout=128, h=16, w=32, gn=True, act=1, groups=32, tile=10)
    oc.check_conv(gpu_lib, "cuda", dtype, n=2, cin=128, cout=3, h=32, w=32, gn=True, act=1, groups=32, tile=10)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [11, 12, 13, 14, 15, 16, 17, 18, 19, 31, 32, 33, 34])
def test_halo_conv_every_tile_config(gpu_lib, cfg):
    """Every conv3x3.hip tile configuration on the real LDS-DMA path (the emulator copies synchronously, so
    only the GPU run can see a missing wait): ragged planes, 2-4 slabs, GN+SiLU, residual, concat, upsample."""
    cout = 3 if cfg == 16 else 200
    for dtype in (torch.bfloat16, torch.float32):
        oc.check_conv(gpu_lib, "cuda", dtype, n=2, cin=128, cout=cout, h=40, w=56, gn=True, act=1, groups=32, res=(cfg != 16), tile=cfg)
        oc.check_conv(gpu_lib, "cuda", dtype, n=1, cin=64, cin2=64, cout=cout, h=33, w=17, gn=True, act=1, groups=32, tile=cfg)
        oc.check_conv(gpu_lib, "cuda", dtype, n=3, cin=64, cout=cout, h=16, w=24, ups=1, tile=cfg)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [21, 22, 23, 24, 25])
def test_dma_igemm_every_tile_config(gpu_lib, cfg):
    """gemm_dma.hip on the real LDS-DMA path: K tails, ragged M/N, gathers, long K loops (ring wrap-around)."""
    for dtype in (torch.bfloat16, torch.float32):
        oc.check_conv(gpu_lib, "cuda", dtype, n=1, cin=328, cout=200, h=37, w=23, ks=1, pad=0, res=True, alpha=0.7, tile=cfg)
        oc.check_conv(gpu_lib, "cuda", dtype, n=2, cin=128, cout=136, h=24, w=20, stride=2, pad=1, tile=cfg)
        oc.check_conv(gpu_lib, "cuda", dtype, n=3, cin=256, cout=72, h=8, w=8, res=True, tile=cfg)
        oc.check_conv(gpu_lib, "cuda", dtype, n=1, cin=64, cin2=192, cout=96, h=16, w=16, ks=1, pad=0, tile=cfg)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [20, 22, 24, 25])
def test_dma_igemm_persistent_stream(gpu_lib, cfg):
    """Many more tiles than resident workgroups: the persistent K-step stream (counted vmcnt across tile borders, bias
    ring in LDS, in-loop epilogues) on the real hardware, short K (1, 2, 5 steps), exact and ragged tiles, repeated
    so a missing wait shows up as a mismatch."""
    for rep in range(3):
        oc.check_conv(gpu_lib, "cuda", torch.bfloat16, n=4, cin=64, cout=256, h=128, w=128, ks=1, pad=0, res=True, tile=cfg, seed=rep)     # nk=1, exact
        oc.check_conv(gpu_lib, "cuda", torch.bfloat16, n=2, cin=128, cout=264, h=131, w=127, ks=1, pad=0, tile=cfg, seed=rep)              # nk=2, ragged
        oc.check_conv(gpu_lib, "cuda", torch.float16, n=4, cin=320, cout=640, h=64, w=64, ks=1, pad=0, res=True, alpha=0.5, tile=cfg, seed=rep)   # nk=5
        oc.check_conv(gpu_lib, "cuda", torch.bfloat16, n=2, cin=3, cout=128, h=192, w=192, tile=cfg, seed=rep)                             # narrow gather
    oc.check_conv(gpu_lib, "cuda", torch.float32, n=2, cin=96, cout=256, h=128, w=128, ks=1, pad=0, res=True, tile=cfg)
    oc.check_geglu(gpu_lib, "cuda", torch.bfloat16, tile=cfg, rows=16384, cin=320, cff=1280)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [51, 52, 53, 54, 55, 56])
def test_gemm_w32_on_hardware(gpu_lib, cfg):
    """The wide GEMM (gemm_w32.hip) on real asynchrony: K of 1 / 2 / 5 / 20 stages around its 3-deep ring, exact and ragged
    tiles, residual / bias / alpha, two sources, GEGLU -- at the row counts of the forward, repeated with fresh seeds so a
    missing wait shows up as a mismatch."""
    bn = 160 if cfg in (51, 52, 55) else 128
    for rep in range(3):
        oc.check_conv(gpu_lib, "cuda", torch.bfloat16, n=4, cin=64, cout=2 * bn, h=128, w=128, ks=1, pad=0, res=True, tile=cfg, seed=rep)            # 1 stage
        oc.check_conv(gpu_lib, "cuda", torch.bfloat16, n=2, cin=128, cout=2 * bn + 8, h=131, w=127, ks=1, pad=0, tile=cfg, seed=rep)                # 2 stages, ragged
        oc.check_conv(gpu_lib, "cuda", torch.float16, n=4, cin=320, cout=640, h=64, w=64, ks=1, pad=0, res=True, alpha=0.5, tile=cfg, seed=rep)     # 5 stages
        oc.check_conv(gpu_lib, "cuda", torch.bfloat16, n=8, cin=1280, cout=320, h=32, w=32, ks=1, pad=0, res=True, tile=cfg, seed=rep)              # 20 stages (ff.net.2)
        oc.check_conv(gpu_lib, "cuda", torch.bfloat16, n=2, cin=320, cin2=640, cout=320, h=64, w=64, ks=1, pad=0, tile=cfg, seed=rep)               # [c0 | c1] conv_shortcut
    oc.check_geglu(gpu_lib, "cuda", torch.bfloat16, tile=cfg, rows=16384, cin=320, cff=1280)
    oc.check_geglu(gpu_lib, "cuda", torch.float16, tile=cfg, rows=2048, cin=1280, cff=5120)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [51, 52, 53, 54])
def test_gemm_w32_conv3x3_gather_on_hardware(gpu_lib, cfg):
    """The wide GEMM's 3x3 gather (VAE / UNet stride-2 downsamplers) on real asynchrony at the forward's sizes, repeated with
    fresh seeds; GroupNorm partial sums from its epilogue on the 128-column tiles."""
    bn = 160 if cfg in (51, 52) else 128
    for rep in range(2):
        oc.check_conv(gpu_lib, "cuda", torch.bfloat16, n=2, cin=128, cout=bn, h=128, w=128, stride=2, asym_pad=True, tile=cfg, seed=rep)        # 18 stages
        oc.check_conv(gpu_lib, "cuda", torch.float16, n=4, cin=320, cout=2 * bn, h=32, w=32, stride=2, pad=1, res=True, alpha=0.5, tile=cfg, seed=rep) # 45 stages
        oc.check_conv(gpu_lib, "cuda", torch.bfloat16, n=1, cin=64, cout=bn + 8, h=37, w=51, stride=1, pad=1, tile=cfg, seed=rep)                  # ragged, odd plane
        oc.check_conv(gpu_lib, "cuda", torch.bfloat16, n=8, cin=1280, cout=1280, h=8, w=8, res=True, tile=cfg, splitk=12, seed=rep)                # split-K: 180 stages over 12 slices
        oc.check_conv(gpu_lib, "cuda", torch.float16, n=8, cin=640, cout=640, h=32, w=32, stride=2, pad=1, tile=cfg, splitk=7, seed=rep)            # 90 stages over 7 slices (13 x 6 + 12)
    if cfg in (53, 54):
        oc.check_conv_gn_part(gpu_lib, "cuda", torch.bfloat16, n=2, cin=128, cout=128, h=128, w=128, groups=32, tile=cfg, ks=3, stride=2, res=False)
        oc.check_conv_gn_part(gpu_lib, "cuda", torch.float16, n=2, cin=256, cout=256, h=64, w=64, groups=32, tile=cfg, ks=1)


@pytest.mark.gpu

@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_dma_igemm_small_tile(gpu_lib, dtype):
    """Tile 26 (64 x 32): the UNet's batch-1 projections as one un-sliced launch (256 / 1024 token rows, K = 640 .. 5120)."""
    oc.check_conv(gpu_lib, "cuda", dtype, n=1, cin=1280, cout=1280, h=16, w=16, ks=1, pad=0, res=True, tile=26)
    oc.check_conv(gpu_lib, "cuda", dtype, n=1, cin=5120, cout=1280, h=16, w=16, ks=1, pad=0, res=True, tile=26)
    oc.check_conv(gpu_lib, "cuda", dtype, n=1, cin=640, cout=640, h=32, w=32, ks=1, pad=0, tile=26)
    oc.check_conv(gpu_lib, "cuda", dtype, n=1, cin=1280, cin2=640, cout=640, h=32, w=32, ks=1, pad=0, tile=26)
    oc.check_conv(gpu_lib, "cuda", dtype, n=1, cin=320, cout=328, h=16, w=17, ks=1, pad=0, res=True, tile=26)

@pytest.mark.parametrize("dtype", DTYPES)
def test_dma_igemm_geglu_bgemm_splitk(gpu_lib, dtype):
    oc.check_geglu(gpu_lib, "cuda", dtype, tile=20, rows=300, cin=320, cff=1280)
    oc.check_bgemm(gpu_lib, "cuda", dtype, tile=20, M=300, N=76, Kd=64)
    oc.check_bgemm(gpu_lib, "cuda", dtype, out_f32=0, tile=22, M=130, N=64, Kd=128)
    for sk in (2, 6, 9):
        oc.check_conv(gpu_lib, "cuda", dtype, n=2, cin=512, cout=200, h=8, w=8, res=True, tile=23, splitk=sk)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [12, 13, 17, 18, 31, 32])
def test_conv_epilogue_groupnorm_partials(gpu_lib, cfg):
    for dtype in (torch.bfloat16, torch.float32):
        oc.check_conv_gn_part(gpu_lib, "cuda", dtype, n=2, cin=128, cout=128, h=72, w=40, groups=32, tile=cfg)
        oc.check_conv_gn_part(gpu_lib, "cuda", dtype, n=1, cin=64, cout=512, h=32, w=32, groups=32, tile=cfg, res=False)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", DTYPES)
def test_subpixel_upsample_conv(gpu_lib, dtype):
    oc.check_conv(gpu_lib, "cuda", dtype, n=2, cin=128, cout=200, h=36, w=52, ups=1, res=True, subpix=True)   # 8-row tiles, ragged
    oc.check_conv(gpu_lib, "cuda", dtype, n=8, cin=256, cout=256, h=64, w=64, ups=1, subpix=True)             # 16-row tiles


@pytest.mark.gpu
def test_dma_igemm_epilogue_groupnorm_partials(gpu_lib):
    """GroupNorm partial sums from the LDS-DMA igemm epilogue: VAE skip-conv / conv_in / downsampler shapes, persistent
    stream included."""
    kw = dict(tile=20)
    oc.check_conv_gn_part(gpu_lib, "cuda", torch.bfloat16, n=2, cin=128, cout=256, h=256, w=256, groups=32, ks=1, **kw)           # persistent, 2 n-tiles
    oc.check_conv_gn_part(gpu_lib, "cuda", torch.bfloat16, n=2, cin=128, cout=128, h=128, w=128, groups=32, stride=2, res=False, **kw)
    oc.check_conv_gn_part(gpu_lib, "cuda", torch.float16, n=4, cin=512, cout=512, h=32, w=32, groups=32, ks=1, **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [41, 42])
def test_w32_subpixel_upsample_conv(gpu_lib, cfg):
    """Sub-pixel Upsample2D form on the wide tiles (conv3x3_w32.hip SUBPIX): the decoder's two big upsamplers at their real
    shapes, ragged source tiles, GroupNorm partial sums per tile and parity."""
    oc.check_conv(gpu_lib, "cuda", torch.bfloat16, n=2, cin=128, cout=256, h=36, w=52, ups=1, subpix=True, tile=cfg)           # ragged both ways
    oc.check_conv(gpu_lib, "cuda", torch.float16, n=4, cin=256, cout=256, h=64, w=64, ups=1, subpix=True, tile=cfg)
    oc.check_conv(gpu_lib, "cuda", torch.bfloat16, n=2, cin=512, cout=512, h=64, w=64, ups=1, subpix=True, tile=cfg)           # 8 slabs
    oc.check_conv_gn_part(gpu_lib, "cuda", torch.bfloat16, n=2, cin=128, cout=256, h=40, w=52, groups=32, subpix=True, tile=cfg, res=False)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_subpixel_upsampler_groupnorm_partials(gpu_lib, dtype):
    oc.check_conv_gn_part(gpu_lib, "cuda", dtype, n=2, cin=128, cout=128, h=40, w=52, groups=32, subpix=True)        # 8-row tiles, ragged
    oc.check_conv_gn_part(gpu_lib, "cuda", dtype, n=4, cin=256, cout=256, h=64, w=64, groups=32, subpix=True, res=False)  # 16-row tiles


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_attention_wide_head(gpu_lib, dtype):
    """VAE mid-block attention shape (1 head x 512 over a 64x64 plane) and ragged variants on the real DMA path."""
    oc.check_attention(gpu_lib, "cuda", dtype, batch=2, heads=1, d=512, tq=4096, tk=4096)
    oc.check_attention(gpu_lib, "cuda", dtype, batch=1, heads=1, d=512, tq=1089, tk=1089, spike=True)      # 33x33 plane
    oc.check_attention(gpu_lib, "cuda", dtype, batch=9, heads=1, d=512, tq=200, tk=77)                     # groups > 8: XCD slots
    # keys split over workgroups + merge launch: the batch-1 / batch-4 forms the planner picks (8 and 2 splits), a ragged one
    oc.check_attention(gpu_lib, "cuda", dtype, batch=1, heads=1, d=512, tq=4096, tk=4096, ksplit=8)
    oc.check_attention(gpu_lib, "cuda", dtype, batch=4, heads=1, d=512, tq=4096, tk=4096, ksplit=2, seed=1)
    oc.check_attention(gpu_lib, "cuda", dtype, batch=1, heads=1, d=512, tq=1089, tk=1089, spike=True, ksplit=4)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_attention_dma_large(gpu_lib, dtype):
    oc.check_attention(gpu_lib, "cuda", dtype, batch=2, heads=5, tq=4096, tk=4096)     # UNet level 0 self-attention
    oc.check_attention(gpu_lib, "cuda", dtype, batch=1, heads=2, tq=130, tk=325, spike=True)
    oc.check_attention(gpu_lib, "cuda", dtype, batch=3, heads=20, tq=64, tk=77)
    # key-split + merge launch (batch-1 self-attention): 64 key tiles over 4 / 2 splits, ragged, one empty split
    oc.check_attention(gpu_lib, "cuda", dtype, batch=1, heads=5, tq=4096, tk=4096, ksplit=4)
    oc.check_attention(gpu_lib, "cuda", dtype, batch=1, heads=10, tq=1024, tk=1024, ksplit=2, seed=1)
    oc.check_attention(gpu_lib, "cuda", dtype, batch=1, heads=2, tq=130, tk=325, spike=True, ksplit=4)
    oc.check_attention(gpu_lib, "cuda", dtype, batch=2, heads=1, tq=70, tk=128, ksplit=3)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", DTYPES)
def test_softmax_row_kernel(gpu_lib, dtype):
    oc.check_softmax(gpu_lib, "cuda", dtype, rows=4099, cols=4096, ldp=4096)
    oc.check_softmax(gpu_lib, "cuda", dtype, rows=130, cols=1024, ldp=1032)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", DTYPES)
def test_u8_boundary(gpu_lib, dtype):
    oc.check_u8_boundary(gpu_lib, "cuda", dtype, n=2, h=64, w=48)


@pytest.mark.gpu
def test_lanczos_resize_u8_is_bit_identical_to_pillow(gpu_lib):
    """csrc/resize.hip on the GPU vs Pillow (src/inference_paired.py:38-41: to a multiple of 8; src/inference_unpaired.py:40,53:
    1280x720 frames to 512x512 and back)."""
    import time
    import numpy as np
    from PIL import Image
    from img2img_turbo_amd.image_ops import lanczos_resize_u8
    rng = np.random.default_rng(2)
    for (n, h, w, oh, ow) in [(2, 561, 843, 560, 840), (2, 720, 1280, 512, 512), (2, 512, 512, 720, 1280), (3, 37, 53, 32, 48)]:
        a = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
        a[:, : h // 3] = (a[:, : h // 3] > 127) * 255
        got = lanczos_resize_u8(torch.from_numpy(a).cuda(), (ow, oh), gpu_lib).cpu().numpy()
        for i in range(n):
            assert np.array_equal(got[i], np.asarray(Image.fromarray(a[i], "RGB").resize((ow, oh), Image.LANCZOS))), (n, h, w, oh, ow)
    x = torch.randint(0, 256, (32, 720, 1280, 3), dtype=torch.uint8, device="cuda")
    lanczos_resize_u8(x, (512, 512), gpu_lib)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        y = lanczos_resize_u8(x, (512, 512), gpu_lib)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print("[resize] 32 x 1280x720 -> 512x512: %.3f ms, %.0f GB/s of input+output bytes" % (dt * 1e3, (x.numel() + y.numel()) / dt / 1e9))


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [41, 42])
def test_w32_conv_every_tile_config(gpu_lib, cfg):
    """Wide-tile 32x32x16-MFMA conv (conv3x3_w32.hip) on the real asynchrony: counted vmcnt weight ring, hidden halo loads,
    GroupNorm+SiLU in the MFMA shadow, halo stored after the slab's last step barrier.  Repeated with fresh seeds: a race
    shows up as a sporadic mismatch."""
    for rep in range(3):
        oc.check_conv(gpu_lib, "cuda", torch.bfloat16, n=2, cin=128, cout=256, h=40, w=72, gn=True, act=1, groups=8, res=True, tile=cfg, seed=rep)
        oc.check_conv(gpu_lib, "cuda", torch.float16, n=1, cin=64, cin2=128, cout=136, h=33, w=65, gn=True, act=1, groups=8, tile=cfg, seed=rep)
        oc.check_conv(gpu_lib, "cuda", torch.bfloat16, n=3, cin=256, cout=384, h=24, w=64, tile=cfg, seed=rep)
        oc.check_conv(gpu_lib, "cuda", torch.float16, n=9, cin=64, cout=128, h=64, w=96, res=True, tile=cfg, seed=rep)       # > 1 tile per workgroup


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [41, 42])
def test_w32_conv_second_contraction(gpu_lib, cfg):
    """The second contraction (i2i_igemm_params.k2_a) of the wide-tile conv on real asynchrony, both forms: a resnet's
    conv_shortcut folded into its conv2 (GroupNorm + SiLU on the first operand, raw second operand; the decoder's and the
    encoder's real shapes) and the decoder's skip conv folded into the sub-pixel upsampler; ragged tiles, 1 / 2 / 3 / 8 slabs of
    the second operand (the two ring slots it alternates between are reused), bf16 and fp16, repeated with fresh seeds."""
    for rep in range(3):
        oc.check_conv(gpu_lib, "cuda", torch.bfloat16, n=2, cin=128, cout=128, h=64, w=96, gn=True, act=1, groups=8, tile=cfg, k2c=256, seed=rep)    # up_blocks.3.resnets.0
        oc.check_conv(gpu_lib, "cuda", torch.float16, n=1, cin=256, cout=256, h=40, w=72, gn=True, act=1, groups=8, tile=cfg, k2c=512, seed=rep)     # up_blocks.2.resnets.0, 8 slabs
        oc.check_conv(gpu_lib, "cuda", torch.bfloat16, n=2, cin=256, cout=256, h=33, w=65, gn=True, act=1, groups=8, tile=cfg, k2c=128, alpha=0.5, seed=rep)  # ragged, 2 slabs
        oc.check_conv(gpu_lib, "cuda", torch.float16, n=3, cin=64, cout=128, h=24, w=64, tile=cfg, k2c=64, bias=False, seed=rep)                     # no norm, 1 slab
        oc.check_conv(gpu_lib, "cuda", torch.bfloat16, n=2, cin=128, cout=256, h=36, w=52, ups=1, subpix=True, tile=cfg, k2c=192, seed=rep)          # sub-pixel form, 3 slabs
        oc.check_conv(gpu_lib, "cuda", torch.float16, n=2, cin=256, cout=256, h=32, w=64, ups=1, subpix=True, tile=cfg, k2c=128, seed=rep)
    oc.check_conv_gn_part(gpu_lib, "cuda", torch.bfloat16, n=2, cin=128, cout=256, h=64, w=64, groups=32, tile=cfg, res=False)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [41, 42])
def test_w32_conv_epilogue_groupnorm_partials(gpu_lib, cfg):
    oc.check_conv_gn_part(gpu_lib, "cuda", torch.bfloat16, n=2, cin=128, cout=256, h=64, w=64, groups=32, tile=cfg)
    oc.check_conv_gn_part(gpu_lib, "cuda", torch.float16, n=1, cin=64, cout=128, h=40, w=72, groups=32, tile=cfg, res=False)
    oc.check_conv_gn_part(gpu_lib, "cuda", torch.bfloat16, n=8, cin=128, cout=128, h=64, w=128, groups=32, tile=0)      # auto route: 8 x 4 x 4 = 128 ... halo
    oc.check_conv_gn_part(gpu_lib, "cuda", torch.bfloat16, n=8, cin=64, cout=128, h=128, w=128, groups=32, tile=0)      # auto route: 256 wide tiles


@pytest.mark.gpu
def test_gn_stats_large_offset_second_pass(gpu_lib):
    """GroupNorm statistics with |mean| = 1000 sigma vs F.group_norm (two-pass): <= 1e-3 on the normalised output in fp32,
    through all three routes (single launch, partial + finalize, finalize over one-pass conv-epilogue partials)."""
    oc.check_gn_stats_offset(gpu_lib, "cuda", torch.float32, n=2, c=128, h=64, w=64, groups=32)                      # single launch
    oc.check_gn_stats_offset(gpu_lib, "cuda", torch.float32, n=2, c=128, h=64, w=64, groups=32, nparts=16, sliced=True)   # ... sliced (ABI v7 counters)
    oc.check_gn_stats_offset(gpu_lib, "cuda", torch.float32, n=2, c=128, h=256, w=256, groups=32, nparts=64)         # partial + finalize
    oc.check_gn_stats_offset(gpu_lib, "cuda", torch.float32, n=2, c=128, h=128, w=128, groups=32, nparts=512, finalize_only=True)
    oc.check_gn_stats_offset(gpu_lib, "cuda", torch.float32, n=1, c=256, h=64, w=64, groups=32, mean=-30.0, std=0.02, nparts=16, finalize_only=True)
    oc.check_gn_stats_offset(gpu_lib, "cuda", torch.float16, n=2, c=128, h=64, w=64, groups=32, mean=100.0, std=0.2)                            # 16-bit inputs
    oc.check_gn_stats_offset(gpu_lib, "cuda", torch.bfloat16, n=2, c=128, h=128, w=128, groups=32, mean=30.0, std=0.3, nparts=512, finalize_only=True)
    oc.check_gn_stats_offset(gpu_lib, "cuda", torch.bfloat16, n=2, c=128, h=128, w=128, groups=32, mean=100.0, std=0.3, nparts=512, finalize_only=True)   # above bf16's flag ratio
    oc.check_gn_stats_offset(gpu_lib, "cuda", torch.float16, n=2, c=320, h=64, w=64, groups=32, mean=200.0, std=0.2)          # cpg 10 (UNet): 4-byte pieces
    oc.check_gn_stats_offset(gpu_lib, "cuda", torch.float16, n=1, c=512, h=64, w=64, groups=32, mean=200.0, std=0.2, nparts=16)   # cpg 16: 16-byte pieces


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_layernorm_folded_into_the_wide_gemm(gpu_lib, dtype):
    """LayerNorm + linear in one launch (i2i_igemm_params.ln_cs) incl. the merge that folds the LayerNorm weight: every tile
    configuration, the to_q | to_k | to_v^T form (transposed column range), the GEGLU form, SD-Turbo's three widths, rows on a DC offset."""
    for tile in (51, 52, 53, 54):
        wide = 160 if tile in (51, 52) else 128
        oc.check_ln_gemm(gpu_lib, "cuda", dtype, rows=1000, cin=320, nq=4 * wide, tile=tile)
        oc.check_ln_gemm(gpu_lib, "cuda", dtype, rows=1000, cin=320, nq=4 * wide, nv=2 * wide, tile=tile, seed=1)
    oc.check_ln_gemm(gpu_lib, "cuda", dtype, rows=4096, cin=320, nq=640, nv=320, tile=50)
    oc.check_ln_gemm(gpu_lib, "cuda", dtype, rows=1024, cin=640, nq=1280, nv=640, tile=50)
    oc.check_ln_gemm(gpu_lib, "cuda", dtype, rows=264, cin=1280, nq=2560, nv=1280, tile=50)
    oc.check_ln_gemm(gpu_lib, "cuda", dtype, rows=64, cin=1280, nq=1280, tile=50)                       # cross-attention to_q at batch 1
    oc.check_ln_gemm(gpu_lib, "cuda", dtype, rows=777, cin=320, nq=2560, geglu=True, tile=50)           # -> the two-workgroups-per-CU form
    oc.check_ln_gemm(gpu_lib, "cuda", dtype, rows=520, cin=1280, nq=10240, geglu=True, tile=50)
    oc.check_ln_gemm(gpu_lib, "cuda", dtype, rows=512, cin=640, nq=640, tile=50, offset=8.0, seed=3)


def test_gn_stats_flag_boundaries(gpu_lib):
    """Both sides of the dtype-aware second-pass ratios (csrc/norm.hip GnRefine) with epilogue-style partial sums (512 parts)."""
    for dtype, lo, hi in ((torch.bfloat16, 118.0, 136.0), (torch.float16, 43.0, 48.0)):
        for mean in (lo, hi):
            oc.check_gn_stats_offset(gpu_lib, "cuda", dtype, h=64, w=64, mean=mean, std=1.0, finalize_only=True, nparts=512)
            oc.check_gn_stats_offset(gpu_lib, "cuda", dtype, h=64, w=64, mean=mean, std=1.0, sliced=True, nparts=20)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_narrow_input_conv(gpu_lib, dtype):
    """conv_narrow.hip: the VAE's conv_in (3 -> 128 at full resolution) as a write-bound kernel of its own; auto route (tile 0) on a
    plane of at least 256 tiles, forced (tile 60) on ragged ones; GroupNorm partial sums of the output."""
    oc.check_conv(gpu_lib, "cuda", dtype, n=2, cin=3, cout=128, h=256, w=256, tile=0)
    oc.check_conv(gpu_lib, "cuda", dtype, n=1, cin=3, cout=256, h=50, w=72, tile=60, bias=False, seed=3)
    oc.check_conv_gn_part(gpu_lib, "cuda", dtype, n=2, cin=3, cout=128, h=128, w=128, groups=32, tile=0, res=False)
    oc.check_conv_gn_part(gpu_lib, "cuda", dtype, n=1, cin=3, cout=128, h=44, w=72, groups=32, tile=60, res=False, seed=5)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gn_apply_one_and_two_sources(gpu_lib, dtype):
    oc.check_gn_apply(gpu_lib, "cuda", dtype, n=2, c=320, h=64, w=64)
    oc.check_gn_apply(gpu_lib, "cuda", dtype, n=2, c=1280, c1=640, h=32, w=32, seed=1)     # up-block resnet input
    oc.check_gn_apply(gpu_lib, "cuda", dtype, n=1, c=320, c1=320, h=63, w=65, act=0, seed=2)
