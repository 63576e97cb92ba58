"""Generate REFERENCE-made golden vectors: instantiate the reference's own ``Pix2Pix_Turbo`` / ``CycleGAN_Turbo`` classes
(/root/reference/src, or $I2I_REFERENCE_SRC) on the CPU with this repo's synthetic state dicts and record inputs -> outputs.

    python tests/golden/make_reference_golden.py          # writes tests/golden/ref_*.pt

Needs ``diffusers`` (0.25.x), ``peft`` and ``transformers`` importable.  THIS CONTAINER HAS NEITHER diffusers NOR peft (and no
network), so the script has never been executed here and tests/golden/ holds no ref_*.pt: the oracle stays "parity unpinned"
(oracle/__init__.py) until someone runs this where the dependencies exist.  tests/test_oracle_kats.py picks the
fixtures up automatically (test_oracle_matches_reference_golden, skipped while they are absent).

How the reference is made to run without a GPU, the hub or its checkpoints (everything patched is I/O, none of it arithmetic):
  * ``from_pretrained`` of AutoencoderKL / UNet2DConditionModel / DDPMScheduler / CLIPTextModel / AutoTokenizer is replaced by
    construction from explicit configs (the sd-turbo values of SURVEY.md Appendix A with this repo's TINY widths) -- random
    init, immediately overwritten by ``load_state_dict`` of the synthetic weights through the reference's own
    ``pretrained_path`` branch (src/pix2pix_turbo.py:114-127) / ``load_ckpt_from_state_dict`` (src/cyclegan_turbo.py:162-190);
  * ``.cuda()`` / ``.to("cuda")`` / ``device="cuda"`` become no-ops (src/pix2pix_turbo.py:33,40-43,158-162; src/model.py:9-10);
  * the text encoder is a stub returning the recorded caption embedding (conditioning is a boundary input of the hot path);
  * the two RNG draws (posterior sample, scheduler noise) come from ``torch.manual_seed`` on the CPU generator, and are
    re-drawn in the same order afterwards so the oracle can be fed the identical eps tensors.
"""
import contextlib
import os
import sys
from unittest import mock

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF_SRC = os.environ.get("I2I_REFERENCE_SRC", "/root/reference/src")


def _vae_config(a):
    return dict(in_channels=a.in_channels, out_channels=a.out_channels, latent_channels=a.latent_channels,
                down_block_types=("DownEncoderBlock2D",) * 4, up_block_types=("UpDecoderBlock2D",) * 4,
                block_out_channels=tuple(a.block_out_channels), layers_per_block=a.layers_per_block, act_fn="silu",
                norm_num_groups=a.norm_num_groups, sample_size=512, scaling_factor=a.scaling_factor)


def _unet_config(a):
    return dict(sample_size=64, in_channels=a.in_channels, out_channels=a.out_channels,
                down_block_types=("CrossAttnDownBlock2D",) * 3 + ("DownBlock2D",),
                up_block_types=("UpBlock2D",) + ("CrossAttnUpBlock2D",) * 3,
                block_out_channels=tuple(a.block_out_channels), layers_per_block=a.layers_per_block,
                attention_head_dim=tuple(a.num_heads), cross_attention_dim=a.cross_attention_dim,
                norm_num_groups=a.norm_num_groups, norm_eps=a.norm_eps, use_linear_projection=True, act_fn="silu",
                center_input_sample=False, flip_sin_to_cos=True, freq_shift=0, downsample_padding=1, mid_block_scale_factor=1)


@contextlib.contextmanager
def offline_cpu_reference(unet_arch, vae_arch, caption_enc):
    """Patches under which the reference modules construct and run on the CPU with no hub access."""
    import diffusers
    import transformers

    class _TextStub(torch.nn.Module):
        def forward(self, ids):
            return (caption_enc.expand(ids.shape[0], -1, -1),)

        def requires_grad_(self, flag=False):
            return self

    class _Tok:
        model_max_length = 77

        def __call__(self, prompt, **kw):
            n = 1 if isinstance(prompt, str) else len(prompt)
            return type("Enc", (), {"input_ids": torch.zeros(n, 77, dtype=torch.long)})()

    def sched_from_pretrained(*a, **k):       # the fields of sd-turbo's scheduler_config.json a DDPMScheduler reads
        return diffusers.DDPMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                                       clip_sample=False, prediction_type="epsilon", timestep_spacing="trailing")

    real_tensor = torch.tensor

    def tensor_cpu(*a, **k):
        k.pop("device", None)
        return real_tensor(*a, **k)

    orig_set_timesteps = diffusers.DDPMScheduler.set_timesteps

    def set_timesteps_cpu(self, n, device=None, **k):
        return orig_set_timesteps(self, n, device="cpu", **k)

    def to_cpu(self, *a, **k):
        a = tuple(x for x in a if not (isinstance(x, (str, torch.device)) and "cuda" in str(x)))
        k = {n: v for n, v in k.items() if not (n == "device" and "cuda" in str(v))}
        return torch.nn.Module._orig_to(self, *a, **k) if (a or k) else self

    patches = [
        mock.patch.object(diffusers.AutoencoderKL, "from_pretrained", classmethod(lambda cls, *a, **k: cls(**_vae_config(vae_arch)))),
        mock.patch.object(diffusers.UNet2DConditionModel, "from_pretrained", classmethod(lambda cls, *a, **k: cls(**_unet_config(unet_arch)))),
        mock.patch.object(diffusers.DDPMScheduler, "from_pretrained", classmethod(lambda cls, *a, **k: sched_from_pretrained())),
        mock.patch.object(diffusers.DDPMScheduler, "set_timesteps", set_timesteps_cpu),
        mock.patch.object(transformers.CLIPTextModel, "from_pretrained", classmethod(lambda cls, *a, **k: _TextStub())),
        mock.patch.object(transformers.AutoTokenizer, "from_pretrained", classmethod(lambda cls, *a, **k: _Tok())),
        mock.patch.object(torch.nn.Module, "cuda", lambda self, *a, **k: self),
        mock.patch.object(torch.Tensor, "cuda", lambda self, *a, **k: self),
        mock.patch.object(torch, "tensor", tensor_cpu),
    ]
    torch.nn.Module._orig_to = torch.nn.Module.to
    patches.append(mock.patch.object(torch.nn.Module, "to", to_cpu))
    with contextlib.ExitStack() as st:
        for p in patches:
            st.enter_context(p)
        sys.path.insert(0, REF_SRC)
        try:
            yield
        finally:
            sys.path.remove(REF_SRC)
            del torch.nn.Module._orig_to


def _draws(seed, shape):
    torch.manual_seed(seed)
    return torch.randn(shape), torch.randn(shape)        # posterior sample (latent_dist.sample()), then sched.step's variance noise


def pix2pix_case(sketch, r=None, seed=21):
    from oracle import TINY_UNET, TINY_VAE
    from oracle.synth import make_inputs, make_pix2pix_weights, split_pix2pix_checkpoint
    mw = make_pix2pix_weights(TINY_UNET, TINY_VAE, seed=seed, sketch=sketch)
    _, _, ckpt = split_pix2pix_checkpoint(mw)
    x, cap, _, nm = make_inputs("sketch" if sketch else "canny", 2, 64, 64, TINY_UNET.cross_attention_dim, seed=seed)
    with offline_cpu_reference(TINY_UNET, TINY_VAE, cap):
        import importlib
        ref_mod = importlib.import_module("pix2pix_turbo")
        path = os.path.join(HERE, "_tmp_ckpt.pkl")
        # the checkpoint carries only lora / conv_in / skip tensors (save_model's filter); the BASE weights of the synthetic model
        # go in through the patched from_pretrained constructors' state dicts below
        torch.save(ckpt, path)
        model = ref_mod.Pix2Pix_Turbo(pretrained_path=path)
        os.remove(path)
        if sketch:      # the sketch model wraps conv_in before loading (src/pix2pix_turbo.py:100-101); with pretrained_path it does not,
            raise SystemExit("TwinConv case needs pretrained_name='sketch_to_image_stochastic' with a local file; not scripted")
        base_unet = {k: v for k, v in mw.unet.items()}
        base_vae = {k: v for k, v in mw.vae.items()}
        missing, unexpected = model.unet.load_state_dict(base_unet, strict=False)
        assert not unexpected, unexpected[:5]
        missing, unexpected = model.vae.load_state_dict(base_vae, strict=False)
        assert not unexpected, unexpected[:5]
        model.set_eval()
        torch.manual_seed(seed)
        with torch.no_grad():
            if r is None:
                out = model(x, prompt_tokens=torch.zeros(2, 77, dtype=torch.long))
            else:
                out = model(x, prompt_tokens=torch.zeros(2, 77, dtype=torch.long), deterministic=False, r=r, noise_map=nm)
    eps_enc, eps_sched = _draws(seed, (2, TINY_VAE.latent_channels, 8, 8))
    return dict(kind="pix2pix", seed=seed, sketch=sketch, r=r, x=x, caption_enc=cap, noise_map=nm, eps_enc=eps_enc, eps_sched=eps_sched, out=out)


def cyclegan_case(direction, seed=22):
    from oracle import TINY_UNET, TINY_VAE
    from oracle.synth import make_cyclegan_weights, make_inputs, split_cyclegan_checkpoint
    mw = make_cyclegan_weights(TINY_UNET, TINY_VAE, rank_unet=16)
    _, _, ckpt = split_cyclegan_checkpoint(mw, rank_unet=16)
    x, cap, _, _ = make_inputs("photo", 1, 64, 64, TINY_UNET.cross_attention_dim, seed=seed)      # the reference forward only works at B = 1
    with offline_cpu_reference(TINY_UNET, TINY_VAE, cap):
        import importlib
        ref_mod = importlib.import_module("cyclegan_turbo")
        path = os.path.join(HERE, "_tmp_cg.pkl")
        torch.save(ckpt, path)
        model = ref_mod.CycleGAN_Turbo(pretrained_path=path)
        os.remove(path)
        model.unet.load_state_dict({k: v for k, v in mw.unet.items()}, strict=False)
        model.eval()
        torch.manual_seed(seed)
        with torch.no_grad():
            out = model(x, direction=direction, caption="x")
    eps_enc, eps_sched = _draws(seed, (1, TINY_VAE.latent_channels, 8, 8))
    return dict(kind="cyclegan", seed=seed, direction=direction, x=x, caption_enc=cap, eps_enc=eps_enc, eps_sched=eps_sched, out=out)


def main():
    try:
        import diffusers  # noqa: F401
        import peft  # noqa: F401
    except ImportError as e:
        raise SystemExit("make_reference_golden: %s -- the reference cannot be instantiated here; no fixture written "
                         "(the oracle stays 'parity unpinned')." % e)
    cases = {"ref_pix2pix_deterministic.pt": lambda: pix2pix_case(False),
             "ref_pix2pix_stochastic_r0.4.pt": lambda: pix2pix_case(False, r=0.4),
             "ref_cyclegan_a2b.pt": lambda: cyclegan_case("a2b"),
             "ref_cyclegan_b2a.pt": lambda: cyclegan_case("b2a")}
    for fn, make in cases.items():
        rec = make()
        torch.save(rec, os.path.join(HERE, fn))
        print("wrote", fn, tuple(rec["out"].shape))


if __name__ == "__main__":
    main()
