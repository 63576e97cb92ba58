"""Weight packer: canonical (diffusers/peft-keyed) weights -> kernel-ready device tensors.

Everything here happens once per (weights, dtype, r): pure tensor plumbing on the host, then one upload.
  * LoRA merge        W' = W + sum_adapters (lora_alpha/r * weight) * B.A     (peft merge; the reference
                      keeps adapters as side branches: src/pix2pix_turbo.py:69,74,206-207)
  * TwinConv fold     W = pre*(1-r) + cur*r                                   (src/pix2pix_turbo.py:16-26)
  * time-embedding    t = 999 is fixed (src/pix2pix_turbo.py:160) so time_emb_proj(silu(temb)) is a
                      per-resnet constant added to conv1's bias
  * conv_out o quant_conv of the VAE encoder composed into one 3x3 conv (both linear, nothing between)
  * layouts           conv OIHW -> [O][KH][KW][I] (k contiguous, I padded to 8); q|k stacked; GEGLU rows
                      interleaved [16 value | 16 gate]; biases / norm affine fp32
"""
import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F


def _pad8(c):
    return (c + 7) // 8 * 8


def subpixel_weights(w):
    """Sub-pixel form of ``conv3x3(nearest_upsample_2x(x))`` (diffusers Upsample2D.forward): the output pixel
    (2y+a, 2x+b) only sees source pixels (y + a - 1 + r, x + b - 1 + c), r, c in {0, 1}, so per output parity the
    nine taps collapse onto a 2x2 kernel whose weights are sums of the original taps.

    ``w``: OIHW fp32 [O, I, 3, 3]  ->  [4 parities (a*2+b)][O][2][2][I] fp32 (k order (r, c, i), as the kernels read).
    """
    o, i, kh, kw = w.shape
    assert kh == 3 and kw == 3
    R = ((0, 1, 1), (0, 0, 1))     # R[a][dy] = source row offset r hit by tap dy for output parity a
    out = torch.zeros(2, 2, o, 2, 2, i, dtype=torch.float32)
    for a in range(2):
        for b in range(2):
            for dy in range(3):
                for dx in range(3):
                    out[a, b, :, R[a][dy], R[b][dx], :] += w[:, :, dy, dx].float()
    return out.reshape(4, o, 2, 2, i)


class Packer:
    def __init__(self, sd: Dict[str, torch.Tensor], scaling: Dict[str, float], dtype, device, r: float = 1.0):
        self.sd = sd
        self.scaling = {k: v * r for k, v in scaling.items()}
        self.dtype = dtype
        self.device = device
        self.r = r
        self._adapters = {}
        for k in sd:
            i = k.find(".lora_A.")
            if i >= 0:
                self._adapters.setdefault(k[:i], []).append(k[i + len(".lora_A."):-len(".weight")])
        self.cache = {}
        self.nbytes = 0

    # ------------------------------------------------------------------ host-side algebra (fp32)
    def has(self, name):
        return (name + ".weight") in self.sd or (name + ".base_layer.weight") in self.sd

    def merged(self, name) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """(W', bias) in fp32 with every adapter of ``name`` merged."""
        if name + ".base_layer.weight" in self.sd:
            w, b = self.sd[name + ".base_layer.weight"], self.sd.get(name + ".base_layer.bias")
        else:
            w, b = self.sd[name + ".weight"], self.sd.get(name + ".bias")
        w = w.float()
        for ad in self._adapters.get(name, []):
            A = self.sd[f"{name}.lora_A.{ad}.weight"].float()
            B = self.sd[f"{name}.lora_B.{ad}.weight"].float()
            s = self.scaling.get(ad, 1.0)
            if w.dim() == 4:
                dw = (B[:, :, 0, 0] @ A.reshape(A.shape[0], -1)).reshape(w.shape)
            else:
                dw = B @ A
            w = w + s * dw
        return w, (None if b is None else b.float())

    def _up(self, t, dtype=None):
        t = t.to(dtype or self.dtype).contiguous().to(self.device)
        self.nbytes += t.numel() * t.element_size()
        return t

    @staticmethod
    def _conv_to_k(w, split=None):
        """OIHW -> [O][KH*KW*(sum of padded source widths)]; ``split`` = channels of source 0 (concat)."""
        o, i, kh, kw = w.shape
        parts = [w] if not split else [w[:, :split], w[:, split:]]
        outs = []
        for pw in parts:
            ci = pw.shape[1]
            t = torch.zeros(o, kh, kw, _pad8(ci))
            t[..., :ci] = pw.permute(0, 2, 3, 1)
            outs.append(t)
        return torch.cat(outs, dim=-1).reshape(o, -1)

    # ------------------------------------------------------------------ packed layers
    def conv(self, name, split=None, extra_bias=None, w_override=None):
        """-> dict(w=[N][K] dtype, b=fp32 or None, n=N, ks=k).  ``split``: channel count of concat source 0."""
        key = ("conv", name, split)
        if key not in self.cache:
            w, b = self.merged(name) if w_override is None else w_override
            if extra_bias is not None:
                b = extra_bias if b is None else b + extra_bias
            if w.dim() == 2:
                w = w[:, :, None, None]
            self.cache[key] = dict(w=self._up(self._conv_to_k(w, split)), b=None if b is None else self._up(b, torch.float32),
                                   n=w.shape[0], ks=w.shape[2])
        return self.cache[key]

    def conv_subpixel(self, name):
        """Upsample2D conv in sub-pixel form: dict(w=[4*N][4*I] dtype, b, n, ks=3, subpix=True) (see subpixel_weights)."""
        key = ("conv_subpix", name)
        if key not in self.cache:
            w, b = self.merged(name)
            o, i = w.shape[0], w.shape[1]
            assert i % 8 == 0
            self.cache[key] = dict(w=self._up(subpixel_weights(w).reshape(4 * o, 4 * i)), b=None if b is None else self._up(b, torch.float32),
                                   n=o, ks=3, subpix=True)
        return self.cache[key]

    def twin_conv_in(self):
        """UNet conv_in as TwinConv folded at this r (src/pix2pix_turbo.py:23-26)."""
        w1, b1 = self.merged("conv_in.conv_in_pretrained")
        w2, b2 = self.merged("conv_in.conv_in_curr")
        r = self.r
        return self.conv("conv_in", w_override=(w1 * (1 - r) + w2 * r, b1 * (1 - r) + b2 * r))

    def stacked_linear(self, names):
        """Rows of several linears stacked (q|k projections share one GEMM)."""
        key = ("stack",) + tuple(names)
        if key not in self.cache:
            ws, bs = zip(*(self.merged(n) for n in names))
            w = torch.cat(ws, 0)
            b = None if bs[0] is None else torch.cat(bs, 0)
            self.cache[key] = dict(w=self._up(w), b=None if b is None else self._up(b, torch.float32), n=w.shape[0], ks=1)
        return self.cache[key]

    def geglu_linear(self, name):
        """ff.net.0.proj with rows (and bias) interleaved per 16: [value 16 | gate 16]."""
        key = ("geglu", name)
        if key not in self.cache:
            w, b = self.merged(name)
            half = w.shape[0] // 2
            assert half % 16 == 0
            idx = torch.arange(half).reshape(-1, 16)
            idx = torch.cat([idx, idx + half], 1).reshape(-1)
            self.cache[key] = dict(w=self._up(w[idx]), b=self._up(b[idx], torch.float32), n=w.shape[0], ks=1)
        return self.cache[key]

    def norm(self, name):
        key = ("norm", name)
        if key not in self.cache:
            self.cache[key] = (self._up(self.sd[name + ".weight"].float(), torch.float32),
                               self._up(self.sd[name + ".bias"].float(), torch.float32))
        return self.cache[key]

    def small_f32(self, name):
        """A tiny conv/linear kept in fp32 for in-register use (post_quant_conv)."""
        key = ("f32", name)
        if key not in self.cache:
            w, b = self.merged(name)
            self.cache[key] = (self._up(w.reshape(w.shape[0], -1), torch.float32), self._up(b, torch.float32))
        return self.cache[key]

    # ------------------------------------------------------------------ folds
    def time_embedding(self, arch):
        """temb at the fixed timestep: Linear2(silu(Linear1([cos | sin]))) -> [1, 4*C0] fp32 (host)."""
        if "temb" not in self.cache:
            c0 = arch.block_out_channels[0]
            half = c0 // 2
            freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
            arg = float(arch.timestep) * freqs
            e = torch.cat([torch.cos(arg), torch.sin(arg)])[None]
            w1, b1 = self.merged("time_embedding.linear_1")
            w2, b2 = self.merged("time_embedding.linear_2")
            self.cache["temb"] = F.linear(F.silu(F.linear(e, w1, b1)), w2, b2)
        return self.cache["temb"]

    def resnet_conv1(self, prefix, arch=None, split=None):
        """conv1 with the constant time-embedding projection folded into its bias (UNet resnets)."""
        extra = None
        if arch is not None and self.has(prefix + ".time_emb_proj"):
            w, b = self.merged(prefix + ".time_emb_proj")
            extra = F.linear(F.silu(self.time_embedding(arch)), w, b)[0]
        return self.conv(prefix + ".conv1", split=split, extra_bias=extra)

    def encoder_out(self):
        """quant_conv (1x1) o encoder.conv_out (3x3) as one 3x3 conv: W'[p,i,y,x] = sum_o Wq[p,o] Wc[o,i,y,x]."""
        wc, bc = self.merged("encoder.conv_out")
        wq, bq = self.merged("quant_conv")
        wq2 = wq[:, :, 0, 0]
        w = torch.einsum("po,oiyx->piyx", wq2, wc)
        b = wq2 @ bc + bq
        return self.conv("encoder.conv_out+quant_conv", w_override=(w, b))
