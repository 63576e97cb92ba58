"""Native CLIP text tower (row f2): ``text_encoder(tokens)[0]`` of the reference (src/pix2pix_turbo.py:192-196,
src/cyclegan_turbo.py:248-253) on the MI355X kernels.

SD-Turbo's text encoder is the OpenCLIP ViT-H/14 text model in HF layout (23 pre-LN layers, width 1024, 16 heads of 64,
GELU MLP of 4096, causal attention over 77 tokens, final LayerNorm).  It re-uses the generator's kernels: LDS-DMA GEMMs
for the projections (q|k stacked, GELU fused into fc1's epilogue, residual adds fused into out_proj / fc2), the
flash-attention kernel with a causal mask, LayerNorm, plus one embedding-gather kernel.  The tokenizer stays on the
host (string processing); this class takes token ids.

Usage is that of ``transformers.CLIPTextModel`` as the reference uses it::

    enc = ClipTextEncoder(state_dict, device="cuda", dtype=torch.bfloat16)      # HF keys, with or without "text_model."
    caption_enc = enc(input_ids)[0]                                             # [B, 77, 1024]
"""
import math
from dataclasses import dataclass

import torch

from . import _capi as K
from . import ops as O


@dataclass(frozen=True)
class ClipTextArch:
    vocab_size: int = 49408
    hidden_size: int = 1024
    intermediate_size: int = 4096
    num_layers: int = 23
    num_heads: int = 16
    max_positions: int = 77
    hidden_act: str = "gelu"
    layer_norm_eps: float = 1e-5


SD_TURBO_CLIP = ClipTextArch()


def arch_from_state_dict(sd, num_heads=None, hidden_act="gelu"):
    p = "text_model." if any(k.startswith("text_model.") for k in sd) else ""
    tok = sd[p + "embeddings.token_embedding.weight"]
    pos = sd[p + "embeddings.position_embedding.weight"]
    n = 1 + max(int(k[len(p + "encoder.layers."):].split(".")[0]) for k in sd if k.startswith(p + "encoder.layers."))
    inter = sd[p + "encoder.layers.0.mlp.fc1.weight"].shape[0]
    C = tok.shape[1]
    return ClipTextArch(vocab_size=tok.shape[0], hidden_size=C, intermediate_size=inter, num_layers=n,
                        num_heads=num_heads or C // 64, max_positions=pos.shape[0], hidden_act=hidden_act)


class _Plan:
    """One flat program for a fixed batch size (77-token rows), replayable as a hipGraph."""

    def __init__(self, enc, B):
        a, dt, dev = enc.arch, enc.dtype, enc.device
        C, I, T, H = a.hidden_size, a.intermediate_size, a.max_positions, a.num_heads
        d = C // H
        rows = B * T
        epc = 4 if dt == torch.float32 else 8
        ldvt = (T + epc - 1) // epc * epc
        new = lambda *shape, dtype=dt: torch.zeros(*shape, dtype=dtype, device=dev)   # noqa: E731
        self.ids = torch.zeros(rows, dtype=torch.int64, device=dev)
        x, x2, h = new(rows, C), new(rows, C), new(rows, C)
        qk, att, vt, mlp = new(rows, 2 * C), new(rows, C), new(B, C, ldvt), new(rows, I)
        self.out = new(rows, C)
        self._keep = [x, x2, h, qk, att, vt, mlp]
        prog = K.Program()
        dtc = O.DT[dt]
        w = enc.w
        add = lambda op, label: prog.add(op[0], dtc, op[1], label)   # noqa: E731
        lin = lambda src, wt, b, dst, cin, n, **kw: O.conv(src, wt, dst, nimg=1, hin=1, win=rows, ho=1, wo=rows, ks=1, c0=cin, lda0=cin,   # noqa: E731
                                                           N=n, bias=b, ldc=n, **kw)
        add(O.embed(self.ids, w["tok"], w["pos"], x, rows=rows, T=T, c=C), "embeddings")
        act = {"gelu": 1, "quick_gelu": 2}[a.hidden_act]
        for i in range(a.num_layers):
            L = w["layers"][i]
            add(O.layernorm(x, h, L["ln1_w"], L["ln1_b"], rows=rows, c=C, eps=a.layer_norm_eps), f"layers.{i}.layer_norm1")
            add(lin(h, L["qk_w"], L["qk_b"], qk, C, 2 * C), f"layers.{i}.q|k_proj")
            add(O.bgemm(L["v_w"], h, vt, M=C, N=T, Kdim=C, lda=C, ldb=C, ldc=ldvt, batch=B, heads=1, a_bs=(0, 0), b_bs=(T * C, 0),
                        c_bs=(C * ldvt, 0), bias=L["v_b"], bias_mode=2), f"layers.{i}.v_proj^T")
            add(O.attention(qk, qk[:, C:], vt, att, batch=B, heads=H, d=d, tq=T, tk=T, ldq=2 * C, ldk=2 * C, ldvt=ldvt, ldo=C,
                            q_bs=T * 2 * C, k_bs=T * 2 * C, vt_bs=C * ldvt, o_bs=T * C, scale=1.0 / math.sqrt(d), causal=1), f"layers.{i}.sdpa")
            add(lin(att, L["o_w"], L["o_b"], x2, C, C, res=x, ldr=C), f"layers.{i}.out_proj+res")
            add(O.layernorm(x2, h, L["ln2_w"], L["ln2_b"], rows=rows, c=C, eps=a.layer_norm_eps), f"layers.{i}.layer_norm2")
            add(lin(h, L["fc1_w"], L["fc1_b"], mlp, C, I, act_out=act), f"layers.{i}.fc1+{a.hidden_act}")
            add(lin(mlp, L["fc2_w"], L["fc2_b"], x, I, C, res=x2, ldr=C), f"layers.{i}.fc2+res")
        add(O.layernorm(x, self.out, w["lnf_w"], w["lnf_b"], rows=rows, c=C, eps=a.layer_norm_eps), "final_layer_norm")
        prog.freeze()
        self.prog, self.graph, self.B, self.T, self.C = prog, None, B, T, C


class ClipTextEncoder(torch.nn.Module):
    def __init__(self, state_dict, arch: ClipTextArch = None, device="cuda", dtype=torch.float32, lib=None, use_graph=True):
        super().__init__()
        self.lib = lib or K.default_library()
        self.device, self.dtype = torch.device(device), dtype
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        if self.lib.backend == "gfx950" and self.device.type != "cuda":
            raise K.I2IError("the gfx950 kernel library needs a CUDA/HIP device, got %s" % device)
        self.arch = arch or arch_from_state_dict(state_dict)
        if self.arch.hidden_size // self.arch.num_heads != 64:
            raise K.I2IError("the fused attention kernel supports head dim 64 (CLIP ViT-H / ViT-L text towers)")
        self.use_graph = use_graph and self.lib.backend == "gfx950"
        p = "text_model." if any(k.startswith("text_model.") for k in state_dict) else ""
        sd = state_dict
        up = lambda t, dt=None: t.detach().to(dt or dtype).contiguous().to(self.device)    # noqa: E731
        f32 = torch.float32
        layers = []
        for i in range(self.arch.num_layers):
            L = f"{p}encoder.layers.{i}."
            g = lambda n: sd[L + n]     # noqa: E731
            layers.append(dict(
                ln1_w=up(g("layer_norm1.weight"), f32), ln1_b=up(g("layer_norm1.bias"), f32),
                ln2_w=up(g("layer_norm2.weight"), f32), ln2_b=up(g("layer_norm2.bias"), f32),
                qk_w=up(torch.cat([g("self_attn.q_proj.weight"), g("self_attn.k_proj.weight")], 0)),
                qk_b=up(torch.cat([g("self_attn.q_proj.bias"), g("self_attn.k_proj.bias")], 0), f32),
                v_w=up(g("self_attn.v_proj.weight")), v_b=up(g("self_attn.v_proj.bias"), f32),
                o_w=up(g("self_attn.out_proj.weight")), o_b=up(g("self_attn.out_proj.bias"), f32),
                fc1_w=up(g("mlp.fc1.weight")), fc1_b=up(g("mlp.fc1.bias"), f32),
                fc2_w=up(g("mlp.fc2.weight")), fc2_b=up(g("mlp.fc2.bias"), f32)))
        self.w = dict(tok=up(sd[p + "embeddings.token_embedding.weight"]), pos=up(sd[p + "embeddings.position_embedding.weight"]),
                      lnf_w=up(sd[p + "final_layer_norm.weight"], f32), lnf_b=up(sd[p + "final_layer_norm.bias"], f32), layers=layers)
        self._plans = {}

    @torch.no_grad()
    def forward(self, input_ids):
        """input_ids int64 [B, 77] -> (last_hidden_state [B, 77, hidden],) like ``CLIPTextModel(ids)``."""
        assert input_ids.dim() == 2 and input_ids.shape[1] == self.arch.max_positions, \
            "pad / truncate to max_length=%d as the reference's tokenizer call does" % self.arch.max_positions
        import contextlib
        B = input_ids.shape[0]
        # launches go to the device the weights live on, whatever the caller's current device is
        with (torch.cuda.device(self.device) if self.device.type == "cuda" else contextlib.nullcontext()):
            plan = self._plans.get(B)
            if plan is None:
                plan = self._plans[B] = _Plan(self, B)
            plan.ids.copy_(input_ids.reshape(-1).to(self.device))
            stream = torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else 0
            if self.use_graph:
                if plan.graph is None:
                    plan.graph = self.lib.graph_create(plan.prog)
                self.lib.graph_launch(plan.graph, stream)
            else:
                self.lib.run(plan.prog, stream)
            return (plan.out.view(B, plan.T, plan.C).clone(),)
