"""The DMA kernels again under the emulator's ASYNCHRONOUS memory model (tests/emu/hip_emu.h, I2I_EMU_ASYNC=1): LDS-DMA copies
and compiler-hidden loads are queued per lane and land only at the s_waitcnt vmcnt(N) that retires them -- the LATEST completion
the in-order counter allows, where the default emulator run is the EARLIEST (everything lands at issue).  A hand-counted wait that
is one operation too generous, or data consumed before the wait that covers it, passes the default run and fails here; the GPU
then only has to confirm what both extreme schedules already agree on.  (Test infrastructure only; the product never loads it.)"""
import pytest
import torch

import opcheck as oc


@pytest.fixture
def async_lib(emu_lib, monkeypatch):
    monkeypatch.setenv("I2I_EMU_ASYNC", "1")
    return emu_lib


@pytest.mark.parametrize("cfg", [12, 13, 14, 16, 17, 18, 31, 34])
def test_halo_conv_waits(async_lib, cfg):
    oc.check_conv(async_lib, "cpu", torch.bfloat16, n=1, cin=128, cout=72, h=20, w=24, gn=True, act=1, res=True, tile=cfg)   # 2 slabs: hand-over + ring wrap
    oc.check_conv(async_lib, "cpu", torch.float32, n=1, cin=96, cout=40, h=9, w=17, tile=cfg)                                 # 3 slabs of 32


@pytest.mark.parametrize("cfg", [41, 42])
@pytest.mark.parametrize("order", [None, "0", "1", "7"])
def test_w32_conv_waits_and_barriers(async_lib, cfg, order, monkeypatch):
    """Wide-tile conv (conv3x3_w32.hip): counted vmcnt of the weight ring + hidden halo loads, and the single slab-end
    barrier that publishes the halo stored after P_8, on the latest-completion memory model with run-ahead wave orders."""
    if order is not None:
        monkeypatch.setenv("I2I_EMU_ORDER", order)
    oc.check_conv(async_lib, "cpu", torch.bfloat16, n=2, cin=192, cout=136, h=18, w=72, gn=True, act=1, res=True, tile=cfg)   # 3 slabs, ragged tiles
    oc.check_conv_gn_part(async_lib, "cpu", torch.float16, n=1, cin=64, cout=128, h=16, w=32, groups=32, tile=cfg)
    oc.check_conv(async_lib, "cpu", torch.bfloat16, n=1, cin=192, cout=136, h=12, w=40, ups=1, subpix=True, tile=cfg)      # sub-pixel form: 4 taps, 2-deep ring, 3 slabs
    oc.check_conv(async_lib, "cpu", torch.bfloat16, n=1, cin=64, cout=128, h=9, w=40, ups=1, subpix=True, tile=cfg, k2c=128)       # + the folded skip conv (2 slabs)
    oc.check_conv(async_lib, "cpu", torch.bfloat16, n=1, cin=128, cout=128, h=18, w=40, gn=True, act=1, tile=cfg, k2c=128)          # resnet conv2 + folded conv_shortcut


def test_subpixel_and_partials_waits(async_lib):
    oc.check_conv(async_lib, "cpu", torch.bfloat16, n=1, cin=128, cout=72, h=12, w=20, ups=1, res=True, subpix=True)
    oc.check_conv_gn_part(async_lib, "cpu", torch.bfloat16, n=1, cin=128, cout=64, h=16, w=16, groups=8, tile=13)


@pytest.mark.parametrize("wgs", [0, 1, 3])
def test_dma_igemm_waits(async_lib, wgs, monkeypatch):
    monkeypatch.setenv("I2I_PERSIST_WGS", str(wgs))
    for cfg in (22, 23, 24, 25, 26):
        oc.check_conv(async_lib, "cpu", torch.bfloat16, n=2, cin=320, cout=200, h=12, w=23, ks=1, pad=0, res=True, tile=cfg)      # 5 K steps, ragged
        oc.check_conv(async_lib, "cpu", torch.float16, n=2, cin=64, cout=136, h=16, w=16, ks=1, pad=0, tile=cfg)                 # 1 K step, exact tiles
    oc.check_conv(async_lib, "cpu", torch.bfloat16, n=2, cin=64, cout=72, h=12, w=10, stride=2, pad=1, tile=20)
    oc.check_conv(async_lib, "cpu", torch.bfloat16, n=2, cin=128, cout=72, h=4, w=4, res=True, tile=23, splitk=3)
    oc.check_geglu(async_lib, "cpu", torch.bfloat16, tile=24, cff=128, rows=200)
    oc.check_conv_gn_part(async_lib, "cpu", torch.bfloat16, n=2, cin=64, cout=128, h=16, w=16, groups=32, tile=20, ks=1)


def test_attention_waits(async_lib):
    oc.check_attention(async_lib, "cpu", torch.bfloat16, batch=1, heads=2, tq=130, tk=325, spike=True)        # d=64 ring of 3, 6 key tiles
    oc.check_attention(async_lib, "cpu", torch.bfloat16, batch=1, heads=2, tq=130, tk=325, spike=True, ksplit=2)   # ring phases of a split that starts at tile 3
    oc.check_attention(async_lib, "cpu", torch.float16, batch=1, heads=1, d=512, tq=70, tk=77, spike=True)    # wide head, double buffer


def test_model_is_sensitive_to_one_operation(async_lib, monkeypatch):
    """Self-test of the model: every wait one operation too generous (I2I_EMU_WAIT_BIAS=1) must break the kernels."""
    monkeypatch.setenv("I2I_EMU_WAIT_BIAS", "1")
    with pytest.raises(AssertionError):
        oc.check_conv(async_lib, "cpu", torch.bfloat16, n=1, cin=128, cout=128, h=16, w=32, tile=13)
    with pytest.raises(AssertionError):
        oc.check_conv(async_lib, "cpu", torch.bfloat16, n=2, cin=320, cout=200, h=12, w=23, ks=1, pad=0, tile=25)
    with pytest.raises(AssertionError):
        oc.check_attention(async_lib, "cpu", torch.bfloat16, batch=1, heads=1, tq=64, tk=325)
    with pytest.raises(AssertionError):
        oc.check_conv(async_lib, "cpu", torch.bfloat16, n=1, cin=128, cout=128, h=16, w=32, tile=42)
    with pytest.raises(AssertionError):
        oc.check_conv(async_lib, "cpu", torch.bfloat16, n=1, cin=448, cout=160, h=9, w=37, ks=1, pad=0, tile=51)


@pytest.mark.slow
def test_whole_forward_under_the_async_model(async_lib):
    """The planned program end to end (tiny architecture, bf16: every DMA kernel in its real sequence) on the adversarial schedule."""
    from oracle import TINY_UNET, TINY_VAE
    from oracle.pipeline import pix2pix_forward
    from oracle.synth import make_inputs, make_pix2pix_weights
    from img2img_turbo_amd.pix2pix_turbo import Pix2Pix_Turbo
    from img2img_turbo_amd.weights import GeneratorWeights
    mw = make_pix2pix_weights(TINY_UNET, TINY_VAE, seed=1)
    x, cap, eps, _ = make_inputs("canny", 1, 64, 64, TINY_UNET.cross_attention_dim)
    ref = pix2pix_forward(mw, x, cap, eps)
    w = GeneratorWeights(mw.unet, mw.vae, mw.unet_arch, mw.vae_arch, mw.unet_scaling, mw.vae_scaling, mw.vae_b2a)
    model = Pix2Pix_Turbo(weights=w, device="cpu", dtype=torch.bfloat16, lib=async_lib)
    out = model(x, caption_enc=cap, eps=eps)
    assert (out.float() - ref).abs().max().item() < 0.25


# ---- barriers: run-ahead wave scheduling (I2I_EMU_ORDER): each wave runs a whole barrier interval ahead of the next one ----
@pytest.mark.parametrize("order", ["0", "1", "7"])
def test_barriers_under_wave_skew(emu_lib, order, monkeypatch):
    """The emulator's default sweep advances all waves in lock step, which hides a missing barrier.  With I2I_EMU_ORDER set, a
    wave runs until it blocks at a BLOCK barrier before the next wave starts (ascending / descending / shuffled leaders) -- on
    top of the latest-completion memory model.  Every LDS pipeline again."""
    monkeypatch.setenv("I2I_EMU_ORDER", order)
    monkeypatch.setenv("I2I_EMU_ASYNC", "1")
    lib = emu_lib
    for cfg in (12, 13, 14, 17, 34):
        oc.check_conv(lib, "cpu", torch.bfloat16, n=1, cin=128, cout=72, h=20, w=24, gn=True, act=1, res=True, tile=cfg)
    oc.check_conv(lib, "cpu", torch.bfloat16, n=1, cin=128, cout=72, h=12, w=20, ups=1, res=True, subpix=True)
    oc.check_conv_gn_part(lib, "cpu", torch.bfloat16, n=1, cin=128, cout=64, h=16, w=16, groups=8, tile=13)
    monkeypatch.setenv("I2I_PERSIST_WGS", "2")
    for cfg in (22, 24, 25):
        oc.check_conv(lib, "cpu", torch.bfloat16, n=2, cin=320, cout=200, h=12, w=23, ks=1, pad=0, res=True, tile=cfg)
    oc.check_conv(lib, "cpu", torch.bfloat16, n=2, cin=128, cout=72, h=4, w=4, res=True, tile=23, splitk=3)
    oc.check_conv_gn_part(lib, "cpu", torch.bfloat16, n=2, cin=64, cout=128, h=16, w=16, groups=32, tile=20, ks=1)
    oc.check_attention(lib, "cpu", torch.bfloat16, batch=1, heads=2, tq=130, tk=325, spike=True)
    oc.check_attention(lib, "cpu", torch.float16, batch=1, heads=1, d=512, tq=70, tk=77, spike=True)
    oc.check_gn_stats(lib, "cpu", torch.bfloat16, c0=128, groups=32, h=40, w=32, nparts=1100, n=1)
    oc.check_layernorm(lib, "cpu", torch.bfloat16)
    oc.check_softmax(lib, "cpu", torch.bfloat16)


def test_skew_notices_a_forgotten_barrier(emu_lib, monkeypatch):
    """Self-test: with every thread's 5th barrier dropped (a K-step barrier of the pipelines) the run-ahead schedule must fail."""
    monkeypatch.setenv("I2I_EMU_ORDER", "1")
    monkeypatch.setenv("I2I_EMU_ASYNC", "1")
    monkeypatch.setenv("I2I_EMU_DROP_BARRIER", "5")
    with pytest.raises(AssertionError):
        oc.check_conv(emu_lib, "cpu", torch.bfloat16, n=1, cin=128, cout=128, h=16, w=32, tile=13)
    with pytest.raises(AssertionError):
        oc.check_conv(emu_lib, "cpu", torch.bfloat16, n=2, cin=320, cout=200, h=12, w=23, ks=1, pad=0, tile=25)
    with pytest.raises(AssertionError):
        oc.check_conv(emu_lib, "cpu", torch.bfloat16, n=1, cin=448, cout=160, h=9, w=37, ks=1, pad=0, tile=51)


@pytest.mark.parametrize("cfg", [51, 52, 53, 54, 55, 56])
@pytest.mark.parametrize("order", [None, "0", "1", "7"])
def test_gemm_w32_waits_and_barriers(async_lib, cfg, order, monkeypatch):
    """Wide GEMM (gemm_w32.hip): the counted vmcnt of the 3-deep operand ring (window pieces spread over four k16 steps, dummy
    pieces past the last stage) and the one barrier per stage, on the latest-completion memory model with run-ahead wave orders."""
    if order is not None:
        monkeypatch.setenv("I2I_EMU_ORDER", order)
    bn = 160 if cfg in (51, 52, 55) else 128
    oc.check_conv(async_lib, "cpu", torch.bfloat16, n=1, cin=448, cout=bn + 40, h=9, w=37, ks=1, pad=0, res=True, tile=cfg)     # 7 stages: the ring wraps twice
    oc.check_conv(async_lib, "cpu", torch.bfloat16, n=1, cin=64, cin2=128, cout=bn, h=5, w=13, ks=1, pad=0, tile=cfg)           # 3 stages = the ring, two sources
    oc.check_conv(async_lib, "cpu", torch.float16, n=1, cin=128, cout=bn, h=4, w=16, ks=1, pad=0, tile=cfg)                     # 2 stages: shorter than the ring
    oc.check_geglu(async_lib, "cpu", torch.bfloat16, tile=cfg, cff=160, rows=150, cin=256)
    if cfg <= 54:       # the 3x3 gather (18 stages: the running tap counter across the ring's wraps, zero-block pieces at the borders)
        oc.check_conv(async_lib, "cpu", torch.bfloat16, n=1, cin=128, cout=bn, h=9, w=11, stride=2, asym_pad=True, tile=cfg)
    if cfg <= 54:       # split-K slices start mid-K (the tap counter of the gather starts at the slice's first stage)
        oc.check_conv(async_lib, "cpu", torch.bfloat16, n=1, cin=128, cout=bn, h=8, w=8, res=True, tile=cfg, splitk=4)
    if cfg in (53, 54):
        oc.check_conv_gn_part(async_lib, "cpu", torch.bfloat16, n=1, cin=64, cout=128, h=32, w=32, groups=8, tile=cfg, ks=3, stride=2)
