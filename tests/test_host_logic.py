"""Host-side logic on the CPU: C-ABI surface, checkpoint readers (both reference layouts), packer folds,
data-parallel sharding with a 2-process gloo group."""
import ctypes
import json
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


def test_packer_merge_and_layouts(emu_lib):
    """Layouts + the DEVICE-side LoRA merge (csrc/lora_merge.hip, here on the emulator twin) against the oracle's host merge,
    at r = 0.4 from the start and again after set_scale() moved r."""
    mw = make_pix2pix_weights(TINY_UNET, TINY_VAE, seed=6)
    r = 0.4
    pk = Packer(mw.vae, mw.vae_scaling, torch.float32, "cpu", emu_lib, r, r)
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
    pu = Packer(mw.unet, mw.unet_scaling, torch.float32, "cpu", emu_lib, 1.0, 1.0)
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
# persistent gather (what bench.py's timed loop and a serving loop use): buffers allocated once, called every step
even = torch.zeros(2, 3, 2, 2)
g = dp.OutputGather(even, 4, dst=0)
slabs = []
for step in range(3):
    even.fill_(10.0 * step + rank)          # the plan's static output buffer is rewritten in place every step
    slab = g()
    if rank == 0:
        slabs.append(slab.data_ptr())
        imgs = g.images()
        assert imgs.shape == (4, 3, 2, 2) and torch.all(imgs[:2] == 10.0 * step) and torch.all(imgs[2:] == 10.0 * step + 1), imgs
# double-buffered form (overlap=True: the gather of step i may run beside the replay of step i+1): the plan's buffer is
# rewritten right after the call -- BEFORE the images of that step are consumed -- and two staging buffers / slabs alternate
g2 = dp.OutputGather(even, 4, dst=0, overlap=True)
slabs2 = []
kept = []
for step in range(5):
    even.fill_(100.0 * step + rank)
    assert g2() is None                       # overlap mode hands out no slab: it is rewritten two steps later (read through images())
    even.fill_(-1.0)                          # the next replay overwrites the static buffer: the staged copy must be what travels
    if rank == 0:
        slabs2.append(g2.slab.data_ptr())
        imgs = g2.images()
        assert imgs.data_ptr() != g2.slab.data_ptr(), "images() of the double-buffered gather must be a copy"
        kept.append(imgs)
        assert torch.all(imgs[:2] == 100.0 * step) and torch.all(imgs[2:] == 100.0 * step + 1), (step, imgs)
if rank == 0:                                 # ... and the copies survive the reuse of both slabs
    for step, imgs in enumerate(kept):
        assert torch.all(imgs[:2] == 100.0 * step) and torch.all(imgs[2:] == 100.0 * step + 1), (step, imgs)
# ragged shards (5 images over 2 ranks: 3 + 2) through the double-buffered gather over 5 steps: both staging buffers / slabs are
# reused twice, the short shard's padding row never reaches images()
lo, hi = dp.shard_bounds(total, rank, world)
rag = torch.zeros(hi - lo, 3, 2, 2)
g3 = dp.OutputGather(rag, total, dst=0, overlap=True)
assert g3.sizes == [3, 2]
for step in range(5):
    rag.copy_(full[lo:hi] + 1000.0 * step)
    g3()
    rag.fill_(-7.0)
    imgs = g3.images()
    if rank == 0:
        assert imgs.shape == (total, 3, 2, 2) and torch.equal(imgs, full + 1000.0 * step), (step, imgs)
    else:
        assert imgs is None
dp.barrier()
if rank == 0:
    assert torch.equal(out, full * 2.0), out
    assert m == float(world)
    assert len(set(slabs)) == 1, "the gather must reuse its buffers"
    assert len(set(slabs2)) == 2 and slabs2[0] == slabs2[2] == slabs2[4] and slabs2[1] == slabs2[3], slabs2
    print("DP_OK")
else:
    assert out is None and g.images() is None and g2.images() is None
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


def _unpack_conv(pw, ks):
    w = pw["w"].float()
    return w.view(w.shape[0], ks, ks, -1).permute(0, 3, 1, 2)


def test_device_lora_remerge_r_sweep(emu_lib):
    """set_scale(r) must leave every packed layer equal to the oracle's host merge at that r: plain convs, the sub-pixel
    upsampler form (linear in the 3x3 kernel), stacked q|k rows, GEGLU-interleaved rows, conv_out o quant_conv, the skip
    convs (which also carry gamma) and the TwinConv fold -- in place, same device addresses (src/pix2pix_turbo.py:206-217)."""
    from img2img_turbo_amd.packer import subpixel_weights
    mw = make_pix2pix_weights(TINY_UNET, TINY_VAE, seed=9, sketch=True)
    for dtype, tol in ((torch.float32, 2e-6), (torch.bfloat16, 1.0 / 128)):
        pv = Packer(mw.vae, mw.vae_scaling, dtype, "cpu", emu_lib)
        pu = Packer(mw.unet, mw.unet_scaling, dtype, "cpu", emu_lib)
        names = dict(conv="decoder.up_blocks.1.resnets.0.conv1", up="decoder.up_blocks.0.upsamplers.0.conv", skip="decoder.skip_conv_2",
                     q="down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q", k="down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_k",
                     ff="down_blocks.0.attentions.0.transformer_blocks.0.ff.net.0.proj")
        got = dict(conv=pv.conv(names["conv"]), up=pv.conv_subpixel(names["up"]), skip=pv.conv(names["skip"], gamma=True),
                   enc=pv.encoder_out(), qk=pu.stacked_linear([names["q"], names["k"]]), ff=pu.geglu_linear(names["ff"]), twin=pu.twin_conv_in())
        ptrs = {k: v["w"].data_ptr() for k, v in got.items()}
        for r in (1.0, 0.4, 0.0, 0.73, 1.0):
            pv.set_scale(r)
            pu.set_scale(r)
            Wv = Weights(mw.vae, {k: v * r for k, v in mw.vae_scaling.items()})
            Wu = Weights(mw.unet, {k: v * r for k, v in mw.unet_scaling.items()})

            def close(a, b, what):
                assert torch.allclose(a.float(), b, atol=tol * max(1.0, float(b.abs().max())), rtol=0), (what, r, dtype, float((a.float() - b).abs().max()))
            close(_unpack_conv(got["conv"], 3), Wv.merged(names["conv"])[0], "conv")
            wu = Wv.merged(names["up"])[0]
            close(got["up"]["w"], subpixel_weights(wu).reshape(4 * wu.shape[0], -1), "subpixel")
            close(_unpack_conv(got["skip"], 1), Wv.merged(names["skip"])[0] * r, "skip*gamma")
            wc, bc = Wv.merged("encoder.conv_out")
            wq = Wv.merged("quant_conv")[0][:, :, 0, 0]
            close(_unpack_conv(got["enc"], 3), torch.einsum("po,oiyx->piyx", wq, wc), "conv_out o quant_conv")
            close(got["qk"]["w"], torch.cat([Wu.merged(names["q"])[0], Wu.merged(names["k"])[0]], 0), "q|k")
            wf = Wu.merged(names["ff"])[0]
            half = wf.shape[0] // 2
            close(got["ff"]["w"][:16], wf[:16], "geglu value rows")
            close(got["ff"]["w"][16:32], wf[half:half + 16], "geglu gate rows")
            w1, w2 = Wu.merged("conv_in.conv_in_pretrained")[0], Wu.merged("conv_in.conv_in_curr")[0]
            tw = _unpack_conv(got["twin"], 3)[:, :w1.shape[1]]
            close(tw, w1 * (1 - r) + w2 * r, "TwinConv")
        assert ptrs == {k: v["w"].data_ptr() for k, v in got.items()}, "a re-merge must not move the packed tensors"


def test_save_model_roundtrip_and_handles(tmp_path, emu_lib):
    """Pix2Pix_Turbo.save_model writes the reference's dict (src/pix2pix_turbo.py:221-229) and it loads back to the same
    canonical weights; the module exposes the attributes callers touch (.unet, .vae, .sched, .timesteps)."""
    from img2img_turbo_amd.pix2pix_turbo import Pix2Pix_Turbo
    from img2img_turbo_amd.weights import GeneratorWeights
    mw = make_pix2pix_weights(TINY_UNET, TINY_VAE, seed=5)
    base_unet, base_vae, ckpt = split_pix2pix_checkpoint(mw)
    gw = from_pix2pix_checkpoint(base_unet, base_vae, ckpt, TINY_UNET, TINY_VAE)
    model = Pix2Pix_Turbo(weights=gw, device="cpu", dtype=torch.float32, lib=emu_lib)
    assert model.lora_rank_unet == ckpt["rank_unet"] and model.target_modules_vae == ckpt["vae_lora_target_modules"]
    f = tmp_path / "model_1001.pkl"
    model.save_model(f)
    sd = load_checkpoint_file(f)
    assert set(sd) == {"unet_lora_target_modules", "vae_lora_target_modules", "rank_unet", "rank_vae", "state_dict_unet", "state_dict_vae"}
    _same(sd["state_dict_unet"], ckpt["state_dict_unet"])
    _same(sd["state_dict_vae"], ckpt["state_dict_vae"])
    gw2 = from_pix2pix_checkpoint(base_unet, base_vae, sd, TINY_UNET, TINY_VAE)
    _same(gw2.unet, mw.unet)
    _same(gw2.vae, mw.vae)
    assert set(model.unet.state_dict()) == set(mw.unet) and set(model.vae.state_dict()) == set(mw.vae)
    assert model.unet.enable_xformers_memory_efficient_attention() is None and model.set_eval() is model
    assert model.timesteps.tolist() == [999]
    with pytest.raises(NotImplementedError):
        model.set_train()


def test_load_sd_turbo_base_from_safetensors(tmp_path):
    """Row f4: the SD-Turbo snapshot reader on synthetic safetensors files laid out as ``from_pretrained(subfolder=...)``
    expects them -- fp32 and the ``.fp16`` variant, and the legacy VAE attention key names (query/key/value/proj_attn with
    [C, C, 1, 1] conv-style weights in old checkpoints are NOT assumed: diffusers renames linear-shaped tensors)."""
    from safetensors.torch import save_file
    from img2img_turbo_amd.weights import load_sd_turbo_base
    mw = make_pix2pix_weights(TINY_UNET, TINY_VAE, seed=3)
    base_unet, base_vae, _ = split_pix2pix_checkpoint(mw)
    base_vae = {k: v for k, v in base_vae.items() if "skip_conv" not in k}          # the hub VAE has no skip convs
    legacy = {"to_q": "query", "to_k": "key", "to_v": "value", "to_out.0": "proj_attn"}
    vae_legacy = {}
    for k, v in base_vae.items():
        for new, old in legacy.items():
            if f".attentions.0.{new}." in k:
                k = k.replace(f".{new}.", f".{old}.")
        vae_legacy[k] = v.contiguous()
    assert any(".query." in k for k in vae_legacy)
    for variant, cast in (("", torch.float32), (".fp16", torch.float16)):
        root = tmp_path / ("snap" + variant)
        (root / "unet").mkdir(parents=True)
        (root / "vae").mkdir()
        save_file({k: v.to(cast).contiguous() for k, v in base_unet.items()}, str(root / "unet" / f"diffusion_pytorch_model{variant}.safetensors"))
        save_file({k: v.to(cast) for k, v in vae_legacy.items()}, str(root / "vae" / f"diffusion_pytorch_model{variant}.safetensors"))
        unet, vae = load_sd_turbo_base(str(root))
        assert set(unet) == set(base_unet) and set(vae) == set(base_vae)
        tol = 0 if cast == torch.float32 else 1e-3
        for k in base_unet:
            assert unet[k].dtype == torch.float32 and torch.allclose(unet[k], base_unet[k], atol=tol, rtol=tol), k
        for k in base_vae:
            assert torch.allclose(vae[k], base_vae[k], atol=tol, rtol=tol), k
    with pytest.raises(FileNotFoundError):
        load_sd_turbo_base(str(tmp_path / "nowhere"))


def test_pretrained_name_constructor_with_local_snapshot(tmp_path, emu_lib, monkeypatch):
    """``Pix2Pix_Turbo(pretrained_name="edge_to_image")`` as written in src/inference_paired.py:31 works once
    I2I_SD_TURBO_DIR points at a local snapshot and ckpt_folder holds the .pkl (the reference downloads both)."""
    from safetensors.torch import save_file
    from img2img_turbo_amd.pix2pix_turbo import Pix2Pix_Turbo
    import img2img_turbo_amd.pix2pix_turbo as P
    mw = make_pix2pix_weights(TINY_UNET, TINY_VAE, seed=4)
    base_unet, base_vae, ckpt = split_pix2pix_checkpoint(mw)
    root = tmp_path / "sd-turbo"
    (root / "unet").mkdir(parents=True)
    (root / "vae").mkdir()
    save_file({k: v.contiguous() for k, v in base_unet.items()}, str(root / "unet" / "diffusion_pytorch_model.safetensors"))
    save_file({k: v.contiguous() for k, v in base_vae.items() if "skip_conv" not in k}, str(root / "vae" / "diffusion_pytorch_model.safetensors"))
    ck = tmp_path / "checkpoints"
    ck.mkdir()
    torch.save(ckpt, ck / "edge_to_image_loras.pkl")
    monkeypatch.setenv("I2I_SD_TURBO_DIR", str(root))
    monkeypatch.setattr(P, "from_pix2pix_checkpoint", lambda u, v, c: from_pix2pix_checkpoint(u, v, c, TINY_UNET, TINY_VAE))
    model = Pix2Pix_Turbo(pretrained_name="edge_to_image", ckpt_folder=str(ck), device="cpu", dtype=torch.float32, lib=emu_lib)
    _same(model.weights.unet, mw.unet)
    _same(model.weights.vae, mw.vae)
    monkeypatch.delenv("I2I_SD_TURBO_DIR")
    with pytest.raises(ValueError):
        Pix2Pix_Turbo(pretrained_name="edge_to_image", ckpt_folder=str(ck), device="cpu", lib=emu_lib)


def test_bench_spawns_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` with no torchrun environment must become the launcher (one rank per GPU on 127.0.0.1) instead
    of asserting; under torch.distributed.run (WORLD_SIZE set) it must NOT re-launch."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    calls = []

    class Launched(Exception):
        pass

    def fake_execv(exe, argv):
        calls.append(argv)
        raise Launched()
    monkeypatch.setattr(os, "execv", fake_execv)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"])
    with pytest.raises(Launched):
        bench.main()
    argv = calls[0]
    assert argv[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in argv and argv[argv.index("--nproc-per-node") + 1] == "4"
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1" and argv[-6:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    # inside a torchrun rank whose WORLD_SIZE disagrees with --gpus: a clear error, no second launch
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    with pytest.raises((SystemExit, AssertionError, RuntimeError, ValueError)):
        bench.main()
    assert len(calls) == 1


_DP_E2E_WORKER = r"""
import os, sys, torch
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests")); sys.path.insert(0, os.path.join(%r, "tests", "emu"))
import build_emu
from img2img_turbo_amd import _capi, dp
from img2img_turbo_amd.arch import TINY_UNET, TINY_VAE
from img2img_turbo_amd.pix2pix_turbo import Pix2Pix_Turbo
from img2img_turbo_amd.synth import make_pix2pix_weights
rank, world, local = dp.init_from_env("gloo")
lib = _capi.Library(build_emu.build())
w = make_pix2pix_weights(TINY_UNET, TINY_VAE, seed=1)               # replicated weights: every rank builds the same
g = torch.Generator().manual_seed(3)
total = 2                                                           # one image per rank (ragged shard bounds: test_data_parallel_two_process_gloo)
torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))           # two ranks share the host cores
x = (torch.rand(total, 1, 64, 64, generator=g) < 0.08).float().expand(total, 3, 64, 64).contiguous()
cap = torch.randn(1, 77, TINY_UNET.cross_attention_dim, generator=g)
eps = torch.randn(total, 4, 8, 8, generator=g)
model = Pix2Pix_Turbo(weights=w, device="cpu", dtype=torch.float32, lib=lib)
lo, hi = dp.shard_bounds(total, rank, world)
mine = model(x[lo:hi], caption_enc=cap, eps=eps[lo:hi])             # this rank's contiguous batch shard, no data-path collective
out = dp.gather_images(mine, total, dst=0)
dp.barrier()
if rank == 0:
    ref = model(x, caption_enc=cap, eps=eps)                         # the whole batch on one rank
    assert out.shape == ref.shape and torch.equal(out, ref), float((out - ref).abs().max())
    print("DP_E2E_OK")
"""


@pytest.mark.slow
def test_data_parallel_forward_two_ranks_gloo(tmp_path, emu_lib):
    """Row (e) end to end on the CPU: two ranks (gloo) each run the planned forward (emulated kernels) on their batch shard with
    replicated weights, the finished images are gathered to rank 0 and equal the single-rank forward of the whole batch bit for
    bit (images are independent: per-sample norms and attention)."""
    script = tmp_path / "dp_e2e.py"
    script.write_text(_DP_E2E_WORKER % (ROOT, ROOT, ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29741", WORLD_SIZE="2")
    env.pop("I2I_EMU_ASYNC", None)
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "DP_E2E_OK" in outs[0]


_SHARED_WEIGHTS_WORKER = r"""
import hashlib, os, sys, torch
import torch.distributed as dist
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests")); sys.path.insert(0, os.path.join(%r, "tests", "emu"))
import build_emu
from img2img_turbo_amd import _capi, dp
from img2img_turbo_amd.arch import TINY_UNET, TINY_VAE
from img2img_turbo_amd.packer import Packer
from img2img_turbo_amd.synth import make_cyclegan_weights
rank, world, local = dp.init_from_env("gloo")
built = []
def build():
    built.append(rank)
    return make_cyclegan_weights(TINY_UNET, TINY_VAE, seed=11, rank_unet=16)
w = dp.shared_weights(build, rank, world, tag="test")
assert built == ([0] if rank == 0 else []), built                   # only rank 0 ran the builder
ref = make_cyclegan_weights(TINY_UNET, TINY_VAE, seed=11, rank_unet=16)
for a, b in ((w.unet, ref.unet), (w.vae, ref.vae), (w.vae_b2a, ref.vae_b2a)):
    assert a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a)
assert w.unet_arch == ref.unet_arch and w.vae_arch == ref.vae_arch and w.unet_scaling == ref.unet_scaling and w.vae_scaling == ref.vae_scaling
# ... and what every rank PACKS from them (device-side LoRA merge on the emulated kernels) is the same bytes
lib = _capi.Library(build_emu.build())
pk = Packer(w.unet, w.unet_scaling, torch.bfloat16, "cpu", lib, 0.7, 0.7)
h = hashlib.sha256()
for name in ("down_blocks.0.resnets.0.conv1", "mid_block.attentions.0.proj_in"):
    h.update(pk.conv(name)["w"].view(torch.uint8).numpy().tobytes())
t = "down_blocks.1.attentions.0.transformer_blocks.0"
for ent in (pk.ln_linear([t + ".attn1.to_q", t + ".attn1.to_k", t + ".attn1.to_v"], t + ".norm1"), pk.ln_linear([t + ".ff.net.0.proj"], t + ".norm3", geglu=True)):
    for k in ("w", "b", "cs"):
        h.update(ent[k].contiguous().view(torch.uint8).numpy().tobytes())
mine = torch.tensor(list(h.digest()), dtype=torch.uint8)
all_ = [torch.zeros_like(mine) for _ in range(world)]
dist.all_gather(all_, mine)
assert all(torch.equal(all_[0], x) for x in all_), "ranks packed different bytes"
assert not [f for f in os.listdir("/dev/shm") if f.startswith("i2i_test_")], "the shared file must be unlinked once every rank mapped it"
dp.barrier()
if rank == 0:
    print("SHARED_WEIGHTS_OK")
"""


@pytest.mark.slow
def test_shared_weights_one_builder_all_ranks_identical(tmp_path, emu_lib):
    """N > 1 set-up: the weights are built ONCE (rank 0) and shared through a /dev/shm safetensors file; every rank
    then holds bit-identical host weights and packs bit-identical device tensors (incl. the LayerNorm-folded layers)."""
    script = tmp_path / "shared_w.py"
    script.write_text(_SHARED_WEIGHTS_WORKER % (ROOT, ROOT, ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29763", WORLD_SIZE="2")
    env.pop("I2I_EMU_ASYNC", None)
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "SHARED_WEIGHTS_OK" in outs[0]


@pytest.mark.slow
def test_bench_two_ranks_on_the_emulator(emu_lib, extra=()):
    """`bench.py --gpus 2` end to end without a GPU: two gloo ranks on the CPU wave emulator (bench.py --emulate, a test hook) go
    through the same rendezvous, batch sharding, replay + double-buffered / serial gather, barrier + max-over-ranks timing
    and JSON line as the N > 1 runs the driver launches on RCCL.  Checked: ONE line, whole-job value, global batch, weak scaling,
    per-rank compute times, ms_gather >= 0 up to timer noise."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29757", WORLD_SIZE="2")
    env.pop("I2I_EMU_ASYNC", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0", "--batch", "1", "--emulate", emu_lib.path] + list(extra)
    procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-2000:] for o in outs]
    assert not any(l.startswith("{") for l in outs[1][0].splitlines()), "only rank 0 prints the JSON line"
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1, outs[0][0]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["warmup"] == 0 and rec["scaling"] == "weak" and rec["higher_is_better"] is True
    assert rec["config"]["global_batch"] == 2 and "dp2" in rec["config"]["parallelism"]
    assert abs(rec["value"] - 2 * 1e3 / rec["ms_per_step"]) < 1e-2 * rec["value"], "value is the whole-job rate (all ranks' images / max-over-ranks time)"
    assert len(rec["ms_compute_per_rank"]) == 2 and all(t > 0 for t in rec["ms_compute_per_rank"])
    assert rec["ms_gather"] > -0.25 * rec["ms_per_step"], rec            # step - compute: >= 0 up to run-to-run noise of two separate loops
    assert ("serial" in rec["gather"]) == bool(extra)
    assert "roofline" not in rec and rec["data"].startswith("EMULATED")


def test_plan_file_export_cli(emu_lib, tmp_path):
    """`python -m img2img_turbo_amd.plan_file` (the exporter a user with checkpoints runs once): writes a loadable file whose boundary
    buffers have the sizes the C host expects; here on the tiny architecture and the emulator library."""
    from img2img_turbo_amd import plan_file
    out = tmp_path / "cli.i2iplan"
    plan_file.main(["--out", str(out), "--synthetic", "--arch", "tiny", "--batch", "2", "--size", "64", "72", "--dtype", "f32", "--stochastic", "--gamma", "0.4",
                    "--device", "cpu", "--lib", emu_lib.path])
    h = emu_lib.plan_load(out)
    try:
        assert emu_lib.plan_io(h, "x")[1] == 2 * 3 * 64 * 72 * 4 and emu_lib.plan_io(h, "out")[1] == 2 * 3 * 64 * 72 * 4
        assert emu_lib.plan_io(h, "eps")[1] == 2 * 4 * 8 * 9 * 4 and emu_lib.plan_io(h, "noise")[1] == 2 * 4 * 8 * 9 * 4
    finally:
        emu_lib.plan_destroy(h)
    # a file of another ABI version / with a foreign header is refused with a message, not loaded
    from img2img_turbo_amd import _capi as K
    blob = bytearray(out.read_bytes())
    bad_abi = bytearray(blob); bad_abi[8:12] = (K.ABI_VERSION - 1).to_bytes(4, "little")
    for name, b in (("abi", bad_abi), ("magic", b"NOTAPLAN" + bytes(blob[8:])), ("short", bytes(blob[:20]))):
        pth = tmp_path / (name + ".i2iplan")
        pth.write_bytes(bytes(b))
        with pytest.raises(K.I2IError) as e:
            emu_lib.plan_load(pth)
        assert ("export it again" in str(e.value)) == (name == "abi"), str(e.value)
    with pytest.raises(K.I2IError):
        emu_lib.plan_load(tmp_path / "does_not_exist.i2iplan")


def test_product_and_oracle_twins_of_arch_and_synth_agree():
    """The product never imports oracle/, so the architecture tables and the synthetic-weight generator exist twice
    (img2img_turbo_amd/{arch,synth}.py for bench.py / smoke(), oracle/{arch,synth}.py for the checker).  They must stay the
    same: every architecture constant, every state-dict key and every tensor bit, for both models."""
    import oracle
    from oracle.synth import make_cyclegan_weights as o_cg, make_pix2pix_weights as o_p2p
    from img2img_turbo_amd import arch as parch
    from img2img_turbo_amd.synth import make_cyclegan_weights as p_cg, make_pix2pix_weights as p_p2p
    for name in ("TINY_UNET", "TINY_VAE", "SD_TURBO_UNET", "SD_TURBO_VAE"):
        assert vars(getattr(oracle, name)) == vars(getattr(parch, name)), name
    a, b = o_p2p(oracle.TINY_UNET, oracle.TINY_VAE, seed=3, sketch=True), p_p2p(parch.TINY_UNET, parch.TINY_VAE, seed=3, sketch=True)
    c, d = o_cg(oracle.TINY_UNET, oracle.TINY_VAE, rank_unet=16), p_cg(parch.TINY_UNET, parch.TINY_VAE, rank_unet=16)
    for x, y in ((a, b), (c, d)):
        for part in ("unet", "vae", "vae_b2a"):
            sx, sy = getattr(x, part, None), getattr(y, part, None)
            assert (sx is None) == (sy is None), part
            if sx is not None:
                assert list(sx) == list(sy), part
                assert all(torch.equal(sx[k], sy[k]) for k in sx), part
        assert x.unet_scaling == y.unet_scaling and x.vae_scaling == y.vae_scaling


def test_wide_gemm_tile_rule_for_the_unet_convs():
    """The planner's (tile, K slices) choice for a 3x3 conv on the wide GEMM reproduces what the sweeps measured best or within 10 % of it
    (profiles/r4h_bench_ops_splitk_w32.log, r4k_bench_ops_unet_planes_w32.log), keeps every slice at >= 8 stages, and declines ops that would
    put fewer than ~100 workgroups on the chip (they stay on the LDS-DMA igemm's split-K)."""
    from img2img_turbo_amd.plan import ForwardPlan
    rule = ForwardPlan._w32_splitk_cfg
    measured = {                                    # (rows, N, K) at batch 8 -> (tile, slices)
        (512, 1280, 11520): (54, 6),                # 1280 -> 1280 @ 8 x 8: 0.034 ms (LDS-DMA igemm 0.050)
        (512, 1280, 23040): (54, 6),                # 2560 -> 1280 @ 8 x 8: 0.049 (0.084)
        (2048, 1280, 11520): (51, 4),               # 1280 -> 1280 @ 16 x 16: 0.081 (0.157; halo conv 0.125)
        (2048, 640, 5760): (52, 4),                 # 640 @ 32 x 32 stride 2: 0.035 (0.055)
        (8192, 640, 5760): (52, 1),                 # 640 -> 640 @ 32 x 32: 0.072 (halo conv 0.090)
        (8192, 640, 11520): (52, 1),                # 1280 -> 640 @ 32 x 32: 0.125 (0.175)
        (32768, 320, 2880): (51, 1),                # 320 -> 320 @ 64 x 64: 0.065 (0.091)
        (32768, 320, 8640): (51, 1),                # 960 -> 320 @ 64 x 64: 0.162 (0.240)
    }
    for (m, n, k), want in measured.items():
        assert rule(m, n, k) == want, (m, n, k, rule(m, n, k))
    for m, n, k in [(512, 1280, 11520), (1024, 640, 5760), (2048, 1280, 23040), (4096, 320, 2880), (256, 1280, 11520)]:
        cfg, sk = rule(m, n, k, 1)
        assert cfg in (51, 52, 54) and sk >= 1 and (k // 64) // sk >= 8, (m, n, k, cfg, sk)
    assert rule(64, 1280, 11520) == (54, 22) and rule(64, 320, 2880) == (0, 0)      # 10 x 22 = 220 workgroups vs 3 x 5 = 15: declined
    assert rule(512, 100, 1152) == (0, 0) and rule(512, 1280, 100) == (0, 0)          # not a 64-stage K / too narrow
