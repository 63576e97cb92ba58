"""End-to-end parity on the CPU: the full planned program (tiny architecture, real topology) through the
emulated HIP kernels vs the pure-PyTorch oracle."""
import pytest
import torch

from oracle import TINY_UNET, TINY_VAE
from oracle.pipeline import cyclegan_forward, pix2pix_forward
from oracle.synth import make_cyclegan_weights, make_inputs, make_pix2pix_weights

from img2img_turbo_amd.cyclegan_turbo import CycleGAN_Turbo
from img2img_turbo_amd.pix2pix_turbo import Pix2Pix_Turbo
from img2img_turbo_amd.weights import GeneratorWeights


def as_product_weights(mw):
    return GeneratorWeights(mw.unet, mw.vae, mw.unet_arch, mw.vae_arch, mw.unet_scaling, mw.vae_scaling, mw.vae_b2a)


@pytest.mark.slow
def test_pix2pix_deterministic_fp32(emu_lib):
    mw = make_pix2pix_weights(TINY_UNET, TINY_VAE, seed=1)
    x, cap, eps, _ = make_inputs("canny", 2, 64, 64, TINY_UNET.cross_attention_dim)
    ref = pix2pix_forward(mw, x, cap, eps)
    model = Pix2Pix_Turbo(weights=as_product_weights(mw), device="cpu", dtype=torch.float32, lib=emu_lib)
    out = model(x, caption_enc=cap, eps=eps)
    err = (out - ref).abs().max().item()
    assert err < 1e-3, err


@pytest.mark.slow
def test_pix2pix_stochastic_twinconv_bf16(emu_lib, monkeypatch):
    mw = make_pix2pix_weights(TINY_UNET, TINY_VAE, seed=2, sketch=True)
    x, cap, eps, nm = make_inputs("sketch", 1, 64, 64, TINY_UNET.cross_attention_dim)
    ref = pix2pix_forward(mw, x, cap, eps, deterministic=False, r=0.4, noise_map=nm)
    model = Pix2Pix_Turbo(weights=as_product_weights(mw), device="cpu", dtype=torch.bfloat16, lib=emu_lib)
    out = model(x, caption_enc=cap, eps=eps, deterministic=False, r=0.4, noise_map=nm)
    err = (out.float() - ref).abs().max().item()
    assert err < 0.25, err   # bf16 end-to-end, x14.6 scheduler amplification
    # the batch-8 route of the UNet's small-plane / stride-2 3x3 convolutions (wide GEMM with the im2col gather and K slices,
    # materialised GroupNorm) reached with the tiny model by lowering the planner's row / workgroup thresholds
    monkeypatch.setenv("I2I_W32_SPLITK_MIN_ROWS", "1")
    monkeypatch.setenv("I2I_W32_SPLITK_MIN_WGS", "1")
    model2 = Pix2Pix_Turbo(weights=as_product_weights(mw), device="cpu", dtype=torch.bfloat16, lib=emu_lib)
    out2 = model2(x, caption_enc=cap, eps=eps, deterministic=False, r=0.4, noise_map=nm)
    plan2 = list(model2._plans.values())[0]
    sliced = [p for (_o, _d, p, _l), k in zip(plan2.prog.ops, plan2.op_kernel) if k == "gemm_w32_kernel" and getattr(p, "ks", 0) == 3]
    assert len(sliced) >= 6 and any(p.splitk > 1 for p in sliced), len(sliced)
    err2 = (out2.float() - ref).abs().max().item()
    assert err2 < 0.25, err2
    # the batch-1 route of the UNet's projections (64 x 32 tiles of the LDS-DMA igemm, no K slices), reached the same way
    monkeypatch.setenv("I2I_SMALL_TILE_MIN_TILES", "1")
    monkeypatch.setenv("I2I_ATT_KSPLIT64_MIN_TK", "64")       # ... and the key-split self-attention (2 splits of the 64-key tiles + the merge launch)
    monkeypatch.setenv("I2I_ATT_KSPLIT64", "2")
    model3 = Pix2Pix_Turbo(weights=as_product_weights(mw), device="cpu", dtype=torch.bfloat16, lib=emu_lib)
    out3 = model3(x, caption_enc=cap, eps=eps, deterministic=False, r=0.4, noise_map=nm)
    plan3 = list(model3._plans.values())[0]
    small = [p for (_o, _d, p, _l) in plan3.prog.ops if getattr(p, "tile", 0) == 26]
    assert len(small) >= 10 and all(p.splitk <= 1 for p in small), len(small)
    assert any(getattr(p, "ksplit", 0) == 2 and p.d == 64 for (_o, _d, p, _l) in plan3.prog.ops if hasattr(p, "tq")), "no key-split attention in the plan"
    err3 = (out3.float() - ref).abs().max().item()
    assert err3 < 0.25, err3
    _check_plan_file_round_trip(emu_lib, model, x, cap, eps, nm, out)


@pytest.mark.slow
def test_mixed_precision_plan_fp16_unet_bf16_vae(emu_lib):
    """Per-network precision (Pix2Pix_Turbo(dtype=bf16, unet_dtype=fp16)): the VAE's ops are recorded as bf16, the UNet's as fp16, the two
    meet in fp32 (posterior latents in, eps-prediction out).  The error against the fp32 oracle lies between the all-bf16 and the
    all-fp16 plan's -- the UNet's error is what the 1-step scheduler multiplies by 14.6."""
    from img2img_turbo_amd import _capi as K
    mw = make_pix2pix_weights(TINY_UNET, TINY_VAE, seed=3)
    x, cap, eps, _ = make_inputs("canny", 1, 64, 64, TINY_UNET.cross_attention_dim)
    ref = pix2pix_forward(mw, x, cap, eps)
    errs = {}
    for name, kw in (("bf16", dict(dtype=torch.bfloat16)), ("mixed", dict(dtype=torch.bfloat16, unet_dtype=torch.float16)), ("f16", dict(dtype=torch.float16))):
        model = Pix2Pix_Turbo(weights=as_product_weights(mw), device="cpu", lib=emu_lib, **kw)
        out = model(x, caption_enc=cap, eps=eps)
        errs[name] = ((out.float() - ref) ** 2).mean().sqrt().item()
        if name == "mixed":
            plan = list(model._plans.values())[0]
            dts = {}
            for (opc, dt, _p, label) in plan.prog.ops:
                net = "vae" if label.startswith(("encoder", "decoder", "input", "output", "ddpm")) else "unet"
                dts.setdefault(net, set()).add(dt)
            assert dts["vae"] == {K.BF16} and dts["unet"] == {K.F16}, dts
            assert plan.ctx.dtype == torch.float16 and plan.out.dtype == torch.bfloat16
    assert errs["f16"] < errs["mixed"] < errs["bf16"], errs


def _check_plan_file_round_trip(lib, model, x, cap, eps, nm, out_python, c_host=True, plan=None):
    """The whole-forward entry for hosts that are not Python (include/i2i_turbo.h i2i_plan_*): the planned forward is written to a plan
    file, loaded by the C library into its OWN buffers (nothing of the Python plan is shared: every pointer is relocated), fed through
    i2i_plan_write, run, read back -- and equals the Python replay bit for bit.  Also: the file refuses a truncated tail and unknown
    buffer names."""
    import os
    import tempfile
    from img2img_turbo_amd import _capi as K
    from img2img_turbo_amd.plan_file import export_plan
    plan = plan or list(model._plans.values())[0]
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "tiny.i2iplan")
        info = export_plan(plan, path)
        assert info["ops"] == plan.prog.n and set(info["io"]) == {"x", "ctx", "eps", "out"} | ({"noise"} if nm is not None else set()) and info["data_bytes"] > 0
        h = lib.plan_load(path)
        try:
            lib.plan_write(h, "x", x.to(plan.x_in.dtype).contiguous())
            lib.plan_write(h, "ctx", cap.to(plan.ctx.dtype).reshape(plan.ctx.shape).contiguous())
            lib.plan_write(h, "eps", eps.to(plan.eps.dtype).contiguous())
            if nm is not None:
                lib.plan_write(h, "noise", nm.to(plan.noise.dtype).expand_as(plan.noise).contiguous())
            lib.plan_run(h)
            got = lib.plan_read(h, "out", torch.empty_like(plan.out, device="cpu"))
            assert torch.equal(got.float(), out_python.float()), float((got.float() - out_python.float()).abs().max())      # (same bits: the model returns them as fp32)
            lib.plan_run(h)          # a second run from the state the first one left (recycled buffers, ticket counters)
            got2 = lib.plan_read(h, "out", torch.empty_like(plan.out, device="cpu"))
            assert torch.equal(got2.float(), got.float())
            with pytest.raises(K.I2IError):
                lib.plan_io(h, "no_such_buffer")
            with pytest.raises(K.I2IError):
                lib.plan_write(h, "eps", torch.zeros(3))
        finally:
            lib.plan_destroy(h)
        # the same file through a host that is not Python: examples/plan_host.c, built with gcc against the emulator library
        import shutil
        import subprocess
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        if c_host and shutil.which("gcc"):
            exe = os.path.join(d, "plan_host")
            subprocess.run(["gcc", "-O2", "-Wall", "-Werror", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "plan_host.c"), "-o", exe,
                            "-L", os.path.dirname(lib.path), "-l" + os.path.basename(lib.path)[3:-3], "-Wl,-rpath," + os.path.dirname(lib.path)], check=True)
            files = {}
            for name, t in (("x", x.to(plan.x_in.dtype)), ("ctx", cap.to(plan.ctx.dtype).reshape(plan.ctx.shape)), ("eps", eps.to(plan.eps.dtype)),
                            ("noise", nm.to(plan.noise.dtype).expand_as(plan.noise))):
                files[name] = os.path.join(d, name + ".bin")
                with open(files[name], "wb") as f:
                    f.write(t.contiguous().view(torch.uint8).numpy().tobytes())
            outp = os.path.join(d, "out.bin")
            env = dict(os.environ)
            env.pop("I2I_EMU_ASYNC", None)
            r = subprocess.run([exe, path, files["x"], files["ctx"], files["eps"], outp, files["noise"]], capture_output=True, text=True, env=env)
            assert r.returncode == 0, r.stderr
            with open(outp, "rb") as f:
                host_out = torch.frombuffer(bytearray(f.read()), dtype=plan.out.dtype).reshape(plan.out.shape)
            assert torch.equal(host_out.float(), out_python.float()), "the C host's images differ from the Python replay"
        with open(path, "rb") as f:
            blob = f.read()
        with open(path, "wb") as f:
            f.write(blob[:len(blob) - 100])
        with pytest.raises(K.I2IError):
            lib.plan_load(path)


@pytest.mark.slow
def test_pix2pix_r_sweep_one_plan_fp32(emu_lib):
    """The per-request gamma sweep of gradio_sketch2image.py:67-91 on ONE model: every r re-merges the packed weights on the
    device (no new packer, no new plan), and each output matches the oracle's unmerged-LoRA forward at that r."""
    mw = make_pix2pix_weights(TINY_UNET, TINY_VAE, seed=2, sketch=True)
    x, cap, eps, nm = make_inputs("sketch", 1, 64, 64, TINY_UNET.cross_attention_dim)
    model = Pix2Pix_Turbo(weights=as_product_weights(mw), device="cpu", dtype=torch.float32, lib=emu_lib)
    plans, packers = None, None
    for r in (0.4, 1.0, 0.7, 0.4):
        ref = pix2pix_forward(mw, x, cap, eps, deterministic=False, r=r, noise_map=nm)
        out = model(x, caption_enc=cap, eps=eps, deterministic=False, r=r, noise_map=nm)
        assert (out - ref).abs().max().item() < 1e-3, r
        if plans is None:
            plans, packers = list(model._plans.values()), list(model._packers.values())
        assert list(model._plans.values()) == plans and list(model._packers.values()) == packers


@pytest.mark.slow
def test_cyclegan_b2a_fp32(emu_lib):
    mw = make_cyclegan_weights(TINY_UNET, TINY_VAE, rank_unet=16)
    x, cap, eps, _ = make_inputs("photo", 1, 64, 64, TINY_UNET.cross_attention_dim)
    ref = cyclegan_forward(mw, x, cap, eps, direction="b2a")
    model = CycleGAN_Turbo(weights=as_product_weights(mw), device="cpu", dtype=torch.float32, lib=emu_lib)
    out = model(x, direction="b2a", caption_emb=cap, eps=eps)
    err = (out - ref).abs().max().item()
    assert err < 1e-3, err
    # the static entry point the reference's callers use (src/cyclegan_turbo.py:199-207, src/train_cyclegan_turbo.py:181)
    out2 = CycleGAN_Turbo.forward_with_networks(x, "b2a", model.vae_enc, model.unet, model.vae_dec, model.sched, model.timesteps, cap, eps=eps)
    assert torch.equal(out, out2)
    other = CycleGAN_Turbo(weights=as_product_weights(mw), device="cpu", dtype=torch.float32, lib=emu_lib)
    with pytest.raises(ValueError):
        CycleGAN_Turbo.forward_with_networks(x, "b2a", model.vae_enc, other.unet, model.vae_dec, model.sched, model.timesteps, cap)
    assert any(k.startswith("vae_b2a.") for k in model.vae_enc.state_dict()) and any(k.startswith("vae.") for k in model.vae_dec.state_dict())
    # the CycleGAN plan (b2a: the second VAE's encoder, the first's decoder) as a plan file through the C library
    _check_plan_file_round_trip(emu_lib, model, x, cap, eps, None, out, c_host=False)


@pytest.mark.slow
def test_pix2pix_halo_everywhere_fp32(emu_lib):
    """Same forward with every eligible 3x3 conv on the halo kernel (tiny planes default to the split-K igemm)."""
    mw = make_pix2pix_weights(TINY_UNET, TINY_VAE, seed=1)
    x, cap, eps, _ = make_inputs("canny", 1, 64, 64, TINY_UNET.cross_attention_dim)
    ref = pix2pix_forward(mw, x, cap, eps)
    model = Pix2Pix_Turbo(weights=as_product_weights(mw), device="cpu", dtype=torch.float32, lib=emu_lib, plan_options=dict(halo_min_tiles=0))
    out = model(x, caption_enc=cap, eps=eps)
    assert (out - ref).abs().max().item() < 1e-3


@pytest.mark.slow
def test_pix2pix_u8_io_matches_float_io(emu_lib):
    """forward_u8 (uint8 HWC in/out, pre/post-processing inside the boundary kernels) == forward on to_tensor'd input,
    post-processed the way the reference callers do (src/inference_paired.py:50,72)."""
    mw = make_pix2pix_weights(TINY_UNET, TINY_VAE, seed=1)
    _, cap, eps, _ = make_inputs("canny", 1, 64, 64, TINY_UNET.cross_attention_dim)
    g = torch.Generator().manual_seed(5)
    img = (torch.rand(1, 64, 64, 1, generator=g) < 0.1).to(torch.uint8).expand(1, 64, 64, 3).contiguous() * 255
    model = Pix2Pix_Turbo(weights=as_product_weights(mw), device="cpu", dtype=torch.float32, lib=emu_lib)
    out_f = model(img.permute(0, 3, 1, 2).float() / 255.0, caption_enc=cap, eps=eps)
    exp = ((out_f * 0.5 + 0.5).clamp(0, 1) * 255.0).to(torch.uint8).permute(0, 2, 3, 1)
    out_u8 = model.forward_u8(img, caption_enc=cap, eps=eps)
    assert out_u8.dtype == torch.uint8 and out_u8.shape == (1, 64, 64, 3)
    assert (out_u8.int() - exp.int()).abs().max() <= 1
    # the uint8-boundary plan as a plan file ("x" and "out" are uint8 NHWC there)
    _check_plan_file_round_trip(emu_lib, model, img, cap, eps, None, out_u8, c_host=False, plan=[p for p in model._plans.values() if p.u8_io][0])


@pytest.mark.slow
def test_sketch_u8_binarisation_matches_the_script(emu_lib):
    """src/inference_paired.py:57-66 (sketch_to_image_stochastic branch): ``c_t = (F.to_tensor(img) < 0.5).float()`` then the
    stochastic forward.  forward_u8(..., sketch=True) does the threshold inside the uint8 boundary kernel."""
    mw = make_pix2pix_weights(TINY_UNET, TINY_VAE, seed=2, sketch=True)
    _, cap, eps, nm = make_inputs("sketch", 1, 64, 64, TINY_UNET.cross_attention_dim)
    g = torch.Generator().manual_seed(7)
    img = torch.randint(0, 256, (1, 64, 64, 3), generator=g, dtype=torch.uint8)          # a grey-level "sketch"
    c_t = ((img.permute(0, 3, 1, 2).float() / 255.0) < 0.5).float()
    ref = pix2pix_forward(mw, c_t, cap, eps, deterministic=False, r=0.4, noise_map=nm)
    model = Pix2Pix_Turbo(weights=as_product_weights(mw), device="cpu", dtype=torch.float32, lib=emu_lib)
    out = model.forward_u8(img, caption_enc=cap, eps=eps, deterministic=False, r=0.4, noise_map=nm, sketch=True)
    exp = ((ref.clamp(-1, 1) * 0.5 + 0.5).clamp(0, 1) * 255.0).to(torch.uint8).permute(0, 2, 3, 1)
    assert (out.int() - exp.int()).abs().max().item() <= 1


@pytest.mark.slow
def test_pix2pix_odd_latent_size_fp32(emu_lib):
    """72 x 88 input (multiples of 8, not of 64): latent 9 x 11, UNet levels 9x11 -> 5x6 -> 3x3 -> 2x2 with explicit
    upsample sizes (row f3; the reference resizes to multiples of 8 only: src/inference_paired.py:38-41)."""
    mw = make_pix2pix_weights(TINY_UNET, TINY_VAE, seed=1)
    x, cap, eps, _ = make_inputs("canny", 1, 72, 88, TINY_UNET.cross_attention_dim)
    ref = pix2pix_forward(mw, x, cap, eps)
    model = Pix2Pix_Turbo(weights=as_product_weights(mw), device="cpu", dtype=torch.float32, lib=emu_lib)
    out = model(x, caption_enc=cap, eps=eps)
    assert out.shape == ref.shape
    assert (out - ref).abs().max().item() < 1e-3



@pytest.mark.slow
def test_buffer_recycling_is_invisible(emu_lib, monkeypatch):
    """The planner recycles activation buffers as soon as the static program no longer reads them (plan.Pool).  The same program
    built with recycling switched off (every intermediate in its own buffer) must give bit-identical outputs: an op reading a
    buffer that was re-lent too early would show here.  Deterministic and stochastic programs, odd plane sizes."""
    from img2img_turbo_amd import plan as plan_mod
    mw = make_pix2pix_weights(TINY_UNET, TINY_VAE, seed=2, sketch=True)
    x, cap, eps, nm = make_inputs("sketch", 2, 72, 88, TINY_UNET.cross_attention_dim)
    kw = dict(caption_enc=cap, eps=eps, deterministic=False, r=0.6, noise_map=nm)
    shared = Pix2Pix_Turbo(weights=as_product_weights(mw), device="cpu", dtype=torch.bfloat16, lib=emu_lib)
    a = shared(x, **kw)
    n_shared = len(next(iter(shared._plans.values())).pool.all)
    monkeypatch.setattr(plan_mod.Pool, "put", lambda self, t: None)
    fresh = Pix2Pix_Turbo(weights=as_product_weights(mw), device="cpu", dtype=torch.bfloat16, lib=emu_lib)
    b = fresh(x, **kw)
    n_fresh = len(next(iter(fresh._plans.values())).pool.all)
    assert n_fresh > 2 * n_shared, (n_fresh, n_shared)          # the switch really removed the sharing
    assert torch.equal(a, b)


@pytest.mark.slow
def test_two_plans_with_different_r_interleave_and_released_plan_refuses(emu_lib):
    """r (LoRA scale / skip gamma / TwinConv fold) is device state shared by every plan of a model.  A caller holding a
    stochastic plan (r = 0.4) and a deterministic one and alternating stage() + run() -- what bench.py does with replay() --
    gets each plan's own r: the plan re-applies it before it executes.  A released plan refuses to run."""
    mw = make_pix2pix_weights(TINY_UNET, TINY_VAE, seed=2, sketch=True)
    x, cap, eps, nm = make_inputs("sketch", 1, 64, 64, TINY_UNET.cross_attention_dim)
    model = Pix2Pix_Turbo(weights=as_product_weights(mw), device="cpu", dtype=torch.float32, lib=emu_lib, use_graph=False)
    ref_s = pix2pix_forward(mw, x, cap, eps, deterministic=False, r=0.4, noise_map=nm)
    ref_s7 = pix2pix_forward(mw, x, cap, eps, deterministic=False, r=0.7, noise_map=nm)
    ps = model.get_plan(1, 64, 64, stochastic=True, r=0.4)
    for r_other, ref_other in ((0.7, ref_s7),):
        model.get_plan(1, 64, 64, stochastic=True, r=r_other)         # same cached plan object, now requested at another r ...
        assert model.get_plan(1, 64, 64, stochastic=True, r=r_other) is ps
        model.stage(ps, x, cap, eps, nm)
        ps.run()
        assert (ps.out - ref_other).abs().max().item() < 1e-3
    ps = model.get_plan(1, 64, 64, stochastic=True, r=0.4)            # ... and back
    model.set_lora_scale(1.0)                                         # someone else moved the device state in between
    model.stage(ps, x, cap, eps, nm)
    ps.run()
    assert (ps.out - ref_s).abs().max().item() < 1e-3
    # Exporting a plan file reads the SHARED packed weights back: the stochastic plan exported after a deterministic plan moved the device
    # state to r = 1 must still carry ITS r (export_plan re-merges through plan._prepare), i.e. the loaded file reproduces the r = 0.4
    # replay bit for bit -- and a released plan refuses to export.
    import os
    import tempfile
    from img2img_turbo_amd.plan_file import export_plan
    out_s = ps.out.clone()
    pd = model.get_plan(1, 64, 64)                                    # deterministic plan: running it re-merges at r = 1
    model.stage(pd, x, cap, eps, None)
    pd.run()
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "stochastic_after_deterministic.i2iplan")
        export_plan(ps, path)
        h = emu_lib.plan_load(path)
        try:
            emu_lib.plan_write(h, "x", x.contiguous())
            emu_lib.plan_write(h, "ctx", cap.to(ps.ctx.dtype).reshape(ps.ctx.shape).contiguous())
            emu_lib.plan_write(h, "eps", eps.contiguous())
            emu_lib.plan_write(h, "noise", nm.expand_as(ps.noise).contiguous())
            emu_lib.plan_run(h)
            got = emu_lib.plan_read(h, "out", torch.empty_like(ps.out))
        finally:
            emu_lib.plan_destroy(h)
        assert torch.equal(got, out_s), float((got - out_s).abs().max())
        model.release_plans()
        with pytest.raises(RuntimeError):
            ps.run()
        with pytest.raises(RuntimeError):
            export_plan(ps, path)


@pytest.mark.slow
@pytest.mark.parametrize("per_image_prompts", [False, True])
def test_cross_attention_kv_of_all_modules_in_two_launches(emu_lib, monkeypatch, per_image_prompts):
    """The K / V^T projections of every attn2 module read the same text states: the planner stacks their weights and runs
    TWO GEMMs for the whole UNet (plan._cross_kv).  Same bits as one to_k / to_v launch per module (same K order per output
    element), 2 x modules - 2 fewer ops; with one shared prompt and with one prompt per image.  (The LoRA re-merge of the
    stacked row blocks at a new r is what test_pix2pix_r_sweep_one_plan_fp32 runs on the default, merged, plan.)"""
    B = 2 if per_image_prompts else 1
    mw = make_pix2pix_weights(TINY_UNET, TINY_VAE, seed=5, sketch=True)
    x, cap, eps, nm = make_inputs("sketch", B, 64, 64, TINY_UNET.cross_attention_dim)
    if per_image_prompts:
        cap = torch.cat([cap, cap.flip(1) * 0.5], 0)
    outs, nops = [], []
    for flag in ("1", "0"):
        monkeypatch.setenv("I2I_CROSS_KV_MERGED", flag)
        model = Pix2Pix_Turbo(weights=as_product_weights(mw), device="cpu", dtype=torch.bfloat16, lib=emu_lib)
        outs.append(model(x, caption_enc=cap, eps=eps, deterministic=False, r=0.7, noise_map=nm))
        nops.append(len(next(iter(model._plans.values())).prog.ops))
    n_attn2 = len({k[:k.index(".to_k.")] for k in mw.unet if ".attn2.to_k." in k})
    assert nops[1] - nops[0] == 2 * n_attn2 - 2, (nops, n_attn2)
    assert torch.equal(outs[0], outs[1])
    ref = pix2pix_forward(mw, x, cap, eps, deterministic=False, r=0.7, noise_map=nm)
    assert (outs[0].float() - ref).abs().max().item() < 0.25
