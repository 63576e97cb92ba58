"""Row f2: native CLIP text tower vs the CPU oracle (oracle/clip.py, itself pinned against ``transformers`` in
tests/test_oracle_kats.py).  Emulator on the CPU, real kernels on the GPU."""
import pytest
import torch

from oracle.clip import SD_TURBO_CLIP, TINY_CLIP, clip_text_forward, make_clip_weights

from img2img_turbo_amd.text_encoder import ClipTextArch, ClipTextEncoder, arch_from_state_dict


def _ids(arch, B, seed=0):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(3, arch.vocab_size, (B, arch.max_positions), generator=g)
    ids[:, 0] = 0
    ids[0, 11:] = 2          # a short prompt: eos padding
    return ids


def _product_arch(a):
    return ClipTextArch(a.vocab_size, a.hidden_size, a.intermediate_size, a.num_layers, a.num_heads, a.max_positions, a.hidden_act, a.layer_norm_eps)


@pytest.mark.slow
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-3), (torch.bfloat16, 0.25)])
def test_text_encoder_tiny_emu(emu_lib, dtype, tol):
    sd = make_clip_weights(TINY_CLIP, seed=5)
    ids = _ids(TINY_CLIP, 2)
    ref = clip_text_forward(sd, TINY_CLIP, ids)
    assert arch_from_state_dict(sd, num_heads=2) == _product_arch(TINY_CLIP)
    enc = ClipTextEncoder(sd, arch=_product_arch(TINY_CLIP), device="cpu", dtype=dtype, lib=emu_lib)
    out = enc(ids)[0].float()
    assert out.shape == ref.shape
    assert (out - ref).abs().max().item() < tol
    # the causal mask is real: changing a LATER token must not change earlier positions
    ids2 = ids.clone(); ids2[:, 50] = 7
    out2 = enc(ids2)[0].float()
    assert torch.equal(out2[:, :50], out[:, :50]) and not torch.equal(out2[:, 50:], out[:, 50:])
    # the tower as a plan file for hosts that are not Python: "ids" -> "ctx", the same bits as the module
    import os
    import tempfile
    from img2img_turbo_amd.plan_file import export_text_plan
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "clip.i2iplan")
        info = export_text_plan(enc, 2, path)
        assert set(info["io"]) == {"ids", "ctx"} and info["io"]["ids"] == 2 * 77 * 8
        h = emu_lib.plan_load(path)
        try:
            emu_lib.plan_write(h, "ids", ids2.reshape(-1).contiguous())
            emu_lib.plan_run(h)
            got = emu_lib.plan_read(h, "ctx", torch.empty(2 * 77, out.shape[-1], dtype=dtype))
        finally:
            emu_lib.plan_destroy(h)
        assert torch.equal(got.float().view(out2.shape), out2)


@pytest.mark.gpu
def test_text_encoder_sd_turbo_gpu(gpu_lib):
    """Full-size tower (23 layers, 1024 wide, 340 M parameters, seeded synthetic weights) on the GPU vs the oracle."""
    sd = make_clip_weights(SD_TURBO_CLIP, seed=5)
    ids = _ids(SD_TURBO_CLIP, 2)
    ref = clip_text_forward(sd, SD_TURBO_CLIP, ids)
    for dtype, tol in ((torch.float32, 2e-3), (torch.bfloat16, 0.35)):
        enc = ClipTextEncoder(sd, device="cuda", dtype=dtype)
        out = enc(ids.cuda())[0].float().cpu()
        err = (out - ref).abs().max().item()
        print("[parity] CLIP text tower %s: max-abs %.3e (ref max %.2f)" % (dtype, err, ref.abs().max().item()))
        assert err < tol
        out_b = enc(ids.cuda())[0].float().cpu()       # graph replay: bit-identical
        assert torch.equal(out, out_b)


@pytest.mark.gpu
def test_prompt_tokens_through_native_text_encoder_gpu(gpu_lib):
    """src/pix2pix_turbo.py:190-199 end to end: prompt_tokens -> text tower -> UNet cross-attention -> image, with the
    native text encoder attached to the generator class, vs the oracle pipeline fed by the oracle text tower."""
    from oracle import SD_TURBO_UNET, SD_TURBO_VAE
    from oracle.pipeline import pix2pix_forward
    from oracle.synth import make_inputs, make_pix2pix_weights
    from img2img_turbo_amd.pix2pix_turbo import Pix2Pix_Turbo
    from img2img_turbo_amd.weights import GeneratorWeights
    mw = make_pix2pix_weights(SD_TURBO_UNET, SD_TURBO_VAE, seed=1234 + 1)
    sd = make_clip_weights(SD_TURBO_CLIP, seed=5)
    ids = _ids(SD_TURBO_CLIP, 1, seed=3)
    x, _, eps, _ = make_inputs("canny", 1, 128, 128, SD_TURBO_UNET.cross_attention_dim, seed=1)
    cap = clip_text_forward(sd, SD_TURBO_CLIP, ids)
    ref = pix2pix_forward(mw, x, cap, eps)
    gw = GeneratorWeights(mw.unet, mw.vae, mw.unet_arch, mw.vae_arch, mw.unet_scaling, mw.vae_scaling, mw.vae_b2a)
    enc = ClipTextEncoder(sd, device="cuda", dtype=torch.float32)
    model = Pix2Pix_Turbo(weights=gw, device="cuda", dtype=torch.float32, text_encoder=enc)
    out = model(x.cuda(), prompt_tokens=ids.cuda(), eps=eps.cuda()).cpu()
    err = (out - ref).abs().max().item()
    print("[parity] SD-Turbo 128x128 fp32 with native text tower: max-abs %.3e" % err)
    assert err < 1e-3
