"""Host-side logic on the CPU: C-ABI surface, checkpoint readers (both reference layouts), packer folds,
data-parallel sharding with a 2-process gloo group."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

from oracle import TINY_UNET, TINY_VAE
from oracle.nn import Weights
from oracle.synth import (make_cyclegan_weights, make_pix2pix_weights, split_cyclegan_checkpoint, split_pix2pix_checkpoint)

from img2img_turbo_amd import _capi, dp
from img2img_turbo_amd.packer import Packer
from img2img_turbo_amd.weights import from_cyclegan_checkpoint, from_pix2pix_checkpoint, load_checkpoint_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_declared_symbol():
    """The hipcc-built product library loads (no GPU needed) and exports every function include/i2i_turbo.h declares."""
    hdr = open(os.path.join(ROOT, "include", "i2i_turbo.h")).read()
    declared = set(re.findall(r"\b(i2i_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 19
    lib = _capi.Library()           # product build; raises loudly if missing
    assert lib.backend == "gfx950"
    for name in declared:
        assert hasattr(lib.lib, name), name
    assert set(_capi.EXPORTS) == declared
    assert lib.lib.i2i_sizeof_op() == ctypes.sizeof(_capi.Op)


def test_bad_arguments_return_errors_not_crashes():
    lib = _capi.Library()
    p = _capi.IgemmParams()
    assert lib.lib.i2i_igemm(ctypes.addressof(p), _capi.BF16, None) == -1
    assert b"null operand" in lib.lib.i2i_last_error()
    with pytest.raises(_capi.I2IError):
        _capi.Library("/nonexistent/libi2i_turbo.so")       # no silent fallback when the extension is missing


def _same(a, b):
    assert set(a) == set(b), (sorted(set(a) ^ set(b))[:5])
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_pix2pix_checkpoint_roundtrip(tmp_path):
    for sketch in (False, True):
        mw = make_pix2pix_weights(TINY_UNET, TINY_VAE, seed=5, sketch=sketch)
        base_unet, base_vae, ckpt = split_pix2pix_checkpoint(mw)
        assert all(("lora" in k) or ("conv_in" in k) for k in ckpt["state_dict_unet"])       # save_model's filter (:227)
        assert all(("lora" in k) or ("skip" in k) for k in ckpt["state_dict_vae"])
        f = tmp_path / f"p2p_{sketch}.pkl"
        torch.save(ckpt, f)
        gw = from_pix2pix_checkpoint(base_unet, base_vae, load_checkpoint_file(f), TINY_UNET, TINY_VAE)
        _same(gw.unet, mw.unet)
        _same(gw.vae, mw.vae)
        assert gw.unet_scaling == mw.unet_scaling and gw.vae_scaling == mw.vae_scaling and gw.is_twin_conv == sketch


def test_cyclegan_checkpoint_roundtrip(tmp_path):
    mw = make_cyclegan_weights(TINY_UNET, TINY_VAE, rank_unet=16)
    base_unet, base_vae, ckpt = split_cyclegan_checkpoint(mw, rank_unet=16)
    f = tmp_path / "cg.pkl"
    torch.save(ckpt, f)
    gw = from_cyclegan_checkpoint(base_unet, load_checkpoint_file(f), TINY_UNET, TINY_VAE)
    _same(gw.unet, mw.unet)
    _same(gw.vae, mw.vae)
    _same(gw.vae_b2a, mw.vae_b2a)
    assert gw.unet_scaling == {"default_encoder": 1.0, "default_decoder": 1.0, "default_others": 1.0}


def test_packer_merge_and_layouts():
    mw = make_pix2pix_weights(TINY_UNET, TINY_VAE, seed=6)
    r = 0.4
    pk = Packer(mw.vae, mw.vae_scaling, torch.float32, "cpu", r)
    W = Weights(mw.vae, {k: v * r for k, v in mw.vae_scaling.items()})
    name = "decoder.up_blocks.2.resnets.0.conv1"
    wm, bm = W.merged(name)
    got = pk.conv(name)
    o, i, kh, kw = wm.shape
    assert got["w"].shape == (o, kh * kw * i) and got["ks"] == 3
    assert torch.allclose(got["w"].view(o, kh, kw, i).permute(0, 3, 1, 2), wm, atol=1e-5)      # [O][KH][KW][I]
    assert torch.allclose(got["b"], bm)
    # 3-channel conv_in is padded to 8 input channels with zeros
    cin = pk.conv("encoder.conv_in")
    assert cin["w"].shape[1] == 9 * 8 and (cin["w"].view(-1, 3, 3, 8)[..., 3:] == 0).all()
    # encoder conv_out o quant_conv composition == applying them in sequence
    x = torch.randn(1, TINY_VAE.block_out_channels[-1], 5, 5)
    wc, bc = W.merged("encoder.conv_out")
    wq, bq = W.merged("quant_conv")
    ref = torch.nn.functional.conv2d(torch.nn.functional.conv2d(x, wc, bc, padding=1), wq, bq)
    eo = pk.encoder_out()
    wfold = eo["w"].view(8, 3, 3, -1).permute(0, 3, 1, 2)
    assert torch.allclose(torch.nn.functional.conv2d(x, wfold, eo["b"], padding=1), ref, atol=1e-4)
    # GEGLU interleave: packed row blocks are [16 value | 16 gate]
    pu = Packer(mw.unet, mw.unet_scaling, torch.float32, "cpu", 1.0)
    nm = "down_blocks.0.attentions.0.transformer_blocks.0.ff.net.0.proj"
    wg, bg = Weights(mw.unet, mw.unet_scaling).merged(nm)
    gp = pu.geglu_linear(nm)
    half = wg.shape[0] // 2
    assert torch.allclose(gp["w"][:16], wg[:16], atol=1e-6) and torch.allclose(gp["w"][16:32], wg[half:half + 16], atol=1e-6)
    assert torch.allclose(gp["b"][32:48], bg[16:32])
    # time embedding fold: bias of conv1 = conv1.bias + time_emb_proj(silu(temb))
    from oracle.unet import time_embedding
    Wu = Weights(mw.unet, mw.unet_scaling)
    temb = time_embedding(Wu, TINY_UNET)
    pre = "down_blocks.0.resnets.0"
    tw, tb = Wu.base(pre + ".time_emb_proj")
    want = Wu.merged(pre + ".conv1")[1] + torch.nn.functional.linear(torch.nn.functional.silu(temb), tw, tb)[0]
    assert torch.allclose(pu.resnet_conv1(pre, TINY_UNET)["b"], want, atol=1e-5)


def test_shard_bounds_cover_exactly():
    for total in (1, 7, 8, 32, 33):
        for world in (1, 2, 3, 8):
            spans = [dp.shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


_WORKER = r"""
import os, sys, torch
sys.path.insert(0, %r)
from img2img_turbo_amd import dp
rank, world, local = dp.init_from_env("gloo")
total = 5                                     # ragged: 3 + 2
full = torch.arange(total * 3 * 2 * 2, dtype=torch.float32).reshape(total, 3, 2, 2)
mine = dp.shard(full, rank, world) * 2.0      # "forward" of this rank's images
out = dp.gather_images(mine, total, dst=0)
m = dp.max_over_ranks(float(rank + 1), "cpu")
dp.barrier()
if rank == 0:
    assert torch.equal(out, full * 2.0), out
    assert m == float(world)
    print("DP_OK")
else:
    assert out is None
"""


def test_data_parallel_two_process_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29731", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=120)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "DP_OK" in outs[0]
