"""CLIP text tower (the ``text_encoder`` of SD-Turbo: OpenCLIP ViT-H/14 text model in HF layout) -- test infrastructure.

Restates ``transformers`` ``CLIPTextModel.forward`` (modeling_clip.py: CLIPTextEmbeddings, CLIPEncoderLayer with
pre-LayerNorm, causal self-attention, ``hidden_act`` MLP, final_layer_norm) over a flat HF-keyed state dict, the way
the reference consumes it: ``text_encoder(tokens)[0]`` = last_hidden_state (src/pix2pix_turbo.py:192-196).

PINNED: unlike the rest of the oracle this sub-path can be checked against the real implementation -- ``transformers``
is installed in the build container -- see tests/test_oracle_kats.py::test_clip_oracle_matches_transformers.
"""
import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass(frozen=True)
class ClipTextArch:
    vocab_size: int = 49408
    hidden_size: int = 1024
    intermediate_size: int = 4096
    num_layers: int = 23
    num_heads: int = 16
    max_positions: int = 77
    hidden_act: str = "gelu"           # SD-2.x text tower; SD-1.x (ViT-L) uses quick_gelu
    layer_norm_eps: float = 1e-5


SD_TURBO_CLIP = ClipTextArch()
TINY_CLIP = ClipTextArch(vocab_size=1000, hidden_size=128, intermediate_size=512, num_layers=2, num_heads=2, max_positions=77)


def _act(x, name):
    if name == "gelu":
        return F.gelu(x)
    if name == "quick_gelu":
        return x * torch.sigmoid(1.702 * x)
    raise ValueError(name)


@torch.no_grad()
def clip_text_forward(sd, arch: ClipTextArch, input_ids):
    """input_ids int64 [B, T<=77] -> last_hidden_state fp32 [B, T, hidden]."""
    p = "text_model." if any(k.startswith("text_model.") for k in sd) else ""
    B, T = input_ids.shape
    x = sd[p + "embeddings.token_embedding.weight"][input_ids] + sd[p + "embeddings.position_embedding.weight"][:T][None]
    H, d = arch.num_heads, arch.hidden_size // arch.num_heads
    mask = torch.full((T, T), float("-inf")).triu(1)        # causal: token t attends to tokens <= t
    for i in range(arch.num_layers):
        L = f"{p}encoder.layers.{i}."
        h = F.layer_norm(x, (arch.hidden_size,), sd[L + "layer_norm1.weight"], sd[L + "layer_norm1.bias"], arch.layer_norm_eps)
        q = F.linear(h, sd[L + "self_attn.q_proj.weight"], sd[L + "self_attn.q_proj.bias"]).view(B, T, H, d).transpose(1, 2)
        k = F.linear(h, sd[L + "self_attn.k_proj.weight"], sd[L + "self_attn.k_proj.bias"]).view(B, T, H, d).transpose(1, 2)
        v = F.linear(h, sd[L + "self_attn.v_proj.weight"], sd[L + "self_attn.v_proj.bias"]).view(B, T, H, d).transpose(1, 2)
        a = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(d) + mask, -1) @ v
        a = a.transpose(1, 2).reshape(B, T, arch.hidden_size)
        x = x + F.linear(a, sd[L + "self_attn.out_proj.weight"], sd[L + "self_attn.out_proj.bias"])
        h = F.layer_norm(x, (arch.hidden_size,), sd[L + "layer_norm2.weight"], sd[L + "layer_norm2.bias"], arch.layer_norm_eps)
        h = _act(F.linear(h, sd[L + "mlp.fc1.weight"], sd[L + "mlp.fc1.bias"]), arch.hidden_act)
        x = x + F.linear(h, sd[L + "mlp.fc2.weight"], sd[L + "mlp.fc2.bias"])
    return F.layer_norm(x, (arch.hidden_size,), sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"], arch.layer_norm_eps)


def make_clip_weights(arch: ClipTextArch, seed=0):
    """Seeded synthetic weights with the HF state-dict keys (``text_model.`` prefix as in transformers 4.x checkpoints)."""
    g = torch.Generator().manual_seed(seed)
    C, I = arch.hidden_size, arch.intermediate_size
    sd = {"text_model.embeddings.token_embedding.weight": torch.randn(arch.vocab_size, C, generator=g) * 0.02,
          "text_model.embeddings.position_embedding.weight": torch.randn(arch.max_positions, C, generator=g) * 0.01}

    def lin(name, o, i):
        sd[name + ".weight"] = torch.randn(o, i, generator=g) / math.sqrt(i)
        sd[name + ".bias"] = torch.randn(o, generator=g) * 0.02

    def ln(name):
        sd[name + ".weight"] = 1 + 0.1 * torch.randn(C, generator=g)
        sd[name + ".bias"] = 0.1 * torch.randn(C, generator=g)

    for i in range(arch.num_layers):
        L = f"text_model.encoder.layers.{i}."
        ln(L + "layer_norm1"); ln(L + "layer_norm2")
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            lin(L + "self_attn." + nm, C, C)
        lin(L + "mlp.fc1", I, C); lin(L + "mlp.fc2", C, I)
    ln("text_model.final_layer_norm")
    return sd
