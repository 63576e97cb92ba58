"""Weight packer: canonical (diffusers/peft-keyed) weights -> kernel-ready device tensors.

Layout work happens once per (weights, dtype) on the host (pure tensor plumbing) followed by one upload; the LoRA merge
itself runs ON THE DEVICE (csrc/lora_merge.hip) into the packed tensors the forward reads, so a new LoRA scale / skip
gamma ``r`` (the reference's per-call ``set_adapters(..., [r])`` / ``decoder.gamma = r``, src/pix2pix_turbo.py:206-207,217)
is one streaming pass over the adapted layers: no re-pack, no re-upload, same device addresses (plans and captured
hipGraphs stay valid).
  * LoRA merge        W' = W + r * sum_adapters (lora_alpha/rank) * B.A       (peft merge; the reference keeps adapters
                      as side branches: src/pix2pix_turbo.py:69,74,206-207).  Per adapted layer the device keeps the fp32
                      master W in packed layout, A (scaling folded in, adapters concatenated along rank) and B.
  * skip gamma        skip_conv_i(skip * gamma) = (gamma * W') skip: folded into the skip-conv weights by the same kernel
  * TwinConv fold     W = pre*(1-r) + cur*r (tiny; re-folded on the host per r)     (src/pix2pix_turbo.py:16-26)
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
    def __init__(self, sd: Dict[str, torch.Tensor], scaling: Dict[str, float], dtype, device, lib, r: float = 1.0, gamma: float = 1.0):
        from . import _capi
        self.sd = sd
        self.scaling = dict(scaling)          # adapter -> lora_alpha / rank (the per-call factor r is applied on the device)
        self.dtype = dtype
        self.device = device
        self.lib = lib
        self._dt = {torch.float32: _capi.F32, torch.bfloat16: _capi.BF16, torch.float16: _capi.F16}[dtype]
        self.r, self.gamma = float(r), float(gamma)
        self.rg = torch.tensor([self.r, self.gamma], dtype=torch.float32, device=device)   # read by the merge kernel and the posterior op
        self._adapters = {}
        for k in sd:
            i = k.find(".lora_A.")
            if i >= 0:
                self._adapters.setdefault(k[:i], []).append(k[i + len(".lora_A."):-len(".weight")])
        self.cache = {}
        self.nbytes = 0
        self._merges = []        # (LoraMergeParams, tensors kept alive): one per adapted packed row block
        self._refolds = []       # callables re-folding the few host-side constants that depend on r (TwinConv)

    # ------------------------------------------------------------------ host-side algebra (fp32)
    def has(self, name):
        return (name + ".weight") in self.sd or (name + ".base_layer.weight") in self.sd

    def base(self, name) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """(W, bias) of the base layer in fp32 (no adapter)."""
        if name + ".base_layer.weight" in self.sd:
            w, b = self.sd[name + ".base_layer.weight"], self.sd.get(name + ".base_layer.bias")
        else:
            w, b = self.sd[name + ".weight"], self.sd.get(name + ".bias")
        return w.float(), (None if b is None else b.float())

    def lora(self, name):
        """(A, B) of ``name`` with every adapter concatenated along rank and lora_alpha/rank folded into A, or None.
        A keeps the base weight's trailing dims ([R, I, kh, kw] / [R, I]); B is [O, R]."""
        ads = self._adapters.get(name, [])
        if not ads:
            return None
        As, Bs = [], []
        for ad in ads:
            A = self.sd[f"{name}.lora_A.{ad}.weight"].float()
            B = self.sd[f"{name}.lora_B.{ad}.weight"].float()
            As.append(A * self.scaling.get(ad, 1.0))
            Bs.append(B.reshape(B.shape[0], B.shape[1]))
        return torch.cat(As, 0), torch.cat(Bs, 1)

    def merged(self, name) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """(W', bias) in fp32 merged ON THE HOST at the current r: only for the tiny folds (time embedding, TwinConv,
        post_quant_conv) and for tests; the packed layers are merged by the device kernel."""
        w, b = self.base(name)
        ab = self.lora(name)
        if ab is not None:
            A, B = ab
            w = w + self.r * (B @ A.reshape(A.shape[0], -1)).reshape(w.shape)
        return w, b

    def _up(self, t, dtype=None):
        t = t.to(dtype or self.dtype).contiguous().to(self.device)
        if t.data_ptr() % 16:      # (CPU device only: a memory-mapped safetensors view is 8-byte aligned; the kernels read 16-byte vectors)
            t = t.clone()
        self.nbytes += t.numel() * t.element_size()
        return t

    def _pack_rows(self, w0, A=None, B=None, use_gamma=False, dst=None, row0=0, ln=None):
        """Packed [N][K] fp32 host rows (+ optional LoRA factors A [R][K], B [N][R]) -> rows ``row0..`` of the device tensor
        the kernels read.  Adapted (or gamma-scaled) rows keep their fp32 master on the device and are (re)merged there.
        ``ln`` = dict(gamma, beta [K] device fp32, bias0 = host fp32 [N] or None, cs, bias_out = device fp32 vectors of the whole
        tensor): fold the LayerNorm in front of the layer into these rows (i2i_lora_merge_params.kscale ..), always on the device."""
        from . import _capi
        n, k = w0.shape
        if dst is None:
            dst = torch.empty(n, k, dtype=self.dtype, device=self.device)
            self.nbytes += dst.numel() * dst.element_size()
        view = dst[row0:row0 + n]
        if A is None and not use_gamma and ln is None:
            view.copy_(w0.to(self.dtype))
            return dst
        assert k % 4 == 0
        p = _capi.LoraMergeParams()
        w0d = self._up(w0, torch.float32)
        keep = [dst, w0d, self.rg]
        p.dst, p.w0, p.N, p.K = view.data_ptr(), w0d.data_ptr(), n, k
        p.rank, p.use_gamma, p.rg = 0, int(use_gamma), self.rg.data_ptr()
        if ln is not None:
            b0 = None if ln.get("bias0") is None else self._up(ln["bias0"], torch.float32)
            p.kscale, p.kshift, p.bias0 = ln["gamma"].data_ptr(), ln["beta"].data_ptr(), (0 if b0 is None else b0.data_ptr())
            p.colsum, p.bias_out = ln["cs"][row0:].data_ptr(), ln["bias_out"][row0:].data_ptr()
            keep += [ln["gamma"], ln["beta"], ln["cs"], ln["bias_out"], b0]
        if A is not None:
            Ad, Bd = self._up(A.reshape(A.shape[0], -1), torch.float32), self._up(B, torch.float32)
            assert Ad.shape == (A.shape[0], k) and Bd.shape == (n, A.shape[0]), (Ad.shape, Bd.shape, n, k)
            p.a, p.b, p.rank = Ad.data_ptr(), Bd.data_ptr(), A.shape[0]
            keep += [Ad, Bd]
        self._merges.append((p, keep))
        self.lib.check(self.lib.lib.i2i_lora_merge(C_addr(p), self._dt, self._stream()))
        return dst

    def _stream(self):
        import ctypes
        if torch.device(self.device).type == "cuda":
            return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        return None

    def set_scale(self, r: float, gamma: Optional[float] = None):
        """New LoRA scale r (and skip gamma, default = r as the reference sets both: src/pix2pix_turbo.py:206-207,217):
        re-merge every adapted layer on the device, re-fold the host-side constants that depend on r.  Asynchronous on the
        current stream; a no-op when nothing changed."""
        gamma = r if gamma is None else gamma
        if float(r) == self.r and float(gamma) == self.gamma:
            return
        self.r, self.gamma = float(r), float(gamma)
        self.rg.copy_(torch.tensor([self.r, self.gamma], dtype=torch.float32))
        st = self._stream()
        for p, _ in self._merges:
            self.lib.check(self.lib.lib.i2i_lora_merge(C_addr(p), self._dt, st))
        for f in self._refolds:
            f()

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
    @staticmethod
    def _scaled(w, b, ab, c):
        """The layer whose output is ``c`` times the layer's: rows of W, the bias and the rows of lora_B scaled (fp32, before any
        rounding: the packed weights are rounded once either way)."""
        if c is None:
            return w, b, ab
        return w * c, (None if b is None else b * c), (None if ab is None else (ab[0], ab[1] * c))

    def conv(self, name, split=None, extra_bias=None, gamma=False, out_scale=None):
        """-> dict(w=[N][K] dtype, b=fp32 or None, n=N, ks=k).  ``split``: channel count of concat source 0;
        ``gamma``: the skip-conv weights carry the decoder's skip gamma (src/model.py:41-43); ``out_scale``: a constant factor on
        the layer's output folded into W, bias and lora_B (attention: to_q carries scale * log2(e), plan.attention_block)."""
        key = ("conv", name, split, bool(gamma), extra_bias is not None, out_scale)     # a second call with other flags must not get the first packing
        if key not in self.cache:
            w, b = self.base(name)
            ab = self.lora(name)
            w, b, ab = self._scaled(w, b, ab, out_scale)
            if extra_bias is not None:
                b = extra_bias if b is None else b + extra_bias
            if w.dim() == 2:
                w = w[:, :, None, None]
            A = B = None
            if ab is not None:
                A, B = ab
                A = self._conv_to_k(A if A.dim() == 4 else A[:, :, None, None], split)
            self.cache[key] = dict(w=self._pack_rows(self._conv_to_k(w, split), A, B, use_gamma=gamma),
                                   b=None if b is None else self._up(b, torch.float32), n=w.shape[0], ks=w.shape[2])
        return self.cache[key]

    def bias_sum(self, name_a, name_b):
        """fp32 device vector bias_a + bias_b (either may be absent): the bias of a launch that computes both layers at once
        (a resnet's conv2 with its conv_shortcut folded in as a second contraction).  Biases carry no adapter."""
        key = ("bias_sum", name_a, name_b)
        if key not in self.cache:
            ba, bb = self.base(name_a)[1], self.base(name_b)[1]
            tot = ba if bb is None else (bb if ba is None else ba + bb)
            self.cache[key] = None if tot is None else self._up(tot, torch.float32)
        return self.cache[key]

    def conv_subpixel(self, name):
        """Upsample2D conv in sub-pixel form: dict(w=[4*N][4*I] dtype, b, n, ks=3, subpix=True) (see subpixel_weights).
        The form is linear in the 3x3 kernel, so lora_A (itself a 3x3 conv) goes through the same map per parity."""
        key = ("conv_subpix", name)
        if key not in self.cache:
            w, b = self.base(name)
            ab = self.lora(name)
            o, i = w.shape[0], w.shape[1]
            assert i % 8 == 0
            w4 = subpixel_weights(w).reshape(4, o, 4 * i)
            dst = None
            for par in range(4):
                A = B = None
                if ab is not None:
                    A, B = subpixel_weights(ab[0])[par].reshape(ab[0].shape[0], 4 * i), ab[1]
                d = self._pack_rows(w4[par], A, B, dst=dst if dst is not None else torch.empty(4 * o, 4 * i, dtype=self.dtype, device=self.device), row0=par * o)
                dst = d
            self.nbytes += dst.numel() * dst.element_size()
            self.cache[key] = dict(w=dst, b=None if b is None else self._up(b, torch.float32), n=o, ks=3, subpix=True)
        return self.cache[key]

    def twin_conv_in(self):
        """UNet conv_in as TwinConv folded at the current r (src/pix2pix_turbo.py:23-26); re-folded by set_scale."""
        key = ("conv", "conv_in", None)
        if key not in self.cache:
            def fold():
                w1, b1 = self.merged("conv_in.conv_in_pretrained")
                w2, b2 = self.merged("conv_in.conv_in_curr")
                r = self.r
                return self._conv_to_k(w1 * (1 - r) + w2 * r), b1 * (1 - r) + b2 * r
            w, b = fold()
            ent = dict(w=self._up(w), b=self._up(b, torch.float32), n=w.shape[0], ks=3)

            def refold():
                w, b = fold()
                ent["w"].copy_(w.to(self.dtype))
                ent["b"].copy_(b)
            self._refolds.append(refold)
            self.cache[key] = ent
        return self.cache[key]

    def stacked_linear(self, names, scale0=None):
        """Rows of several linears stacked (q|k projections share one GEMM).  ``scale0``: output factor of the FIRST layer (see conv)."""
        key = ("stack", scale0) + tuple(names)
        if key not in self.cache:
            parts = []
            for i, n in enumerate(names):
                w, b, ab = self._scaled(*self.base(n), self.lora(n), scale0 if i == 0 else None)
                parts.append(((w, b), ab))
            rows = sum(p[0][0].shape[0] for p in parts)
            k = parts[0][0][0].shape[1]
            dst = torch.empty(rows, k, dtype=self.dtype, device=self.device)
            self.nbytes += dst.numel() * dst.element_size()
            r0 = 0
            for (w, _), ab in parts:
                self._pack_rows(w, ab[0] if ab else None, ab[1] if ab else None, dst=dst, row0=r0)
                r0 += w.shape[0]
            bs = [p[0][1] for p in parts]
            b = None if bs[0] is None else torch.cat(bs, 0)
            self.cache[key] = dict(w=dst, b=None if b is None else self._up(b, torch.float32), n=rows, ks=1)
        return self.cache[key]

    def ln_linear(self, names, norm_name, geglu=False, scale0=None):
        """Linear layer(s) with the LayerNorm in front of them folded in (diffusers BasicTransformerBlock: norm1 -> attn1.to_q / to_k /
        to_v, norm2 -> attn2.to_q, norm3 -> ff.net.0.proj): rows of ``names`` stacked (GEGLU rows interleaved per 16 as geglu_linear),
        W' = W * gamma[k] written by the device-side merge together with the two per-row vectors the consumer GEMM needs
        (i2i_igemm_params.ln_cs): -> dict(w, b = bias' = bias + W.beta, cs = row sums of the stored W', n, ks=1).  ``scale0``: output factor
        of the FIRST layer (see conv)."""
        key = ("ln", norm_name, bool(geglu), scale0) + tuple(names)
        if key not in self.cache:
            gamma, beta = self.norm(norm_name)
            parts = []
            for i, n in enumerate(names):
                w, b, ab = self._scaled(*self.base(n), self.lora(n), scale0 if i == 0 else None)
                parts.append(((w, b), ab))
            rows = sum(p[0][0].shape[0] for p in parts)
            k = parts[0][0][0].shape[1]
            assert gamma.numel() == k, (norm_name, gamma.shape, k)
            dst = torch.empty(rows, k, dtype=self.dtype, device=self.device)
            cs = torch.zeros(rows, dtype=torch.float32, device=self.device)
            bout = torch.zeros(rows, dtype=torch.float32, device=self.device)
            self.nbytes += dst.numel() * dst.element_size() + 8 * rows
            r0 = 0
            for (w, b), ab in parts:
                A, B = (ab if ab else (None, None))
                if geglu:
                    half = w.shape[0] // 2
                    assert half % 16 == 0
                    idx = torch.arange(half).reshape(-1, 16)
                    idx = torch.cat([idx, idx + half], 1).reshape(-1)
                    w, b, B = w[idx], (None if b is None else b[idx]), (None if B is None else B[idx])
                self._pack_rows(w, A, B, dst=dst, row0=r0, ln=dict(gamma=gamma, beta=beta, bias0=b, cs=cs, bias_out=bout))
                r0 += w.shape[0]
            self.cache[key] = dict(w=dst, b=bout, cs=cs, n=rows, ks=1)
        return self.cache[key]

    def geglu_linear(self, name):
        """ff.net.0.proj with rows (and bias) interleaved per 16: [value 16 | gate 16]."""
        key = ("geglu", name)
        if key not in self.cache:
            w, b = self.base(name)
            ab = self.lora(name)
            half = w.shape[0] // 2
            assert half % 16 == 0
            idx = torch.arange(half).reshape(-1, 16)
            idx = torch.cat([idx, idx + half], 1).reshape(-1)
            self.cache[key] = dict(w=self._pack_rows(w[idx], ab[0] if ab else None, ab[1][idx] if ab else None),
                                   b=self._up(b[idx], torch.float32), n=w.shape[0], ks=1)
        return self.cache[key]

    def norm(self, name):
        key = ("norm", name)
        if key not in self.cache:
            self.cache[key] = (self._up(self.sd[name + ".weight"].float(), torch.float32),
                               self._up(self.sd[name + ".bias"].float(), torch.float32))
        return self.cache[key]

    def small_f32(self, name):
        """A tiny conv/linear kept in fp32 for in-register use (post_quant_conv; never a LoRA target)."""
        key = ("f32", name)
        if key not in self.cache:
            assert name not in self._adapters, name
            w, b = self.base(name)
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
            for n in ("time_embedding.linear_1", "time_embedding.linear_2"):
                assert n not in self._adapters, n     # not a LoRA target in either reference model (r-independent fold)
            w1, b1 = self.base("time_embedding.linear_1")
            w2, b2 = self.base("time_embedding.linear_2")
            self.cache["temb"] = F.linear(F.silu(F.linear(e, w1, b1)), w2, b2)
        return self.cache["temb"]

    def resnet_conv1(self, prefix, arch=None, split=None):
        """conv1 with the constant time-embedding projection folded into its bias (UNet resnets)."""
        extra = None
        if arch is not None and self.has(prefix + ".time_emb_proj"):
            assert prefix + ".time_emb_proj" not in self._adapters
            w, b = self.base(prefix + ".time_emb_proj")
            extra = F.linear(F.silu(self.time_embedding(arch)), w, b)[0]
        return self.conv(prefix + ".conv1", split=split, extra_bias=extra)

    def encoder_out(self):
        """quant_conv (1x1) o encoder.conv_out (3x3) as one 3x3 conv: W'[p,i,y,x] = sum_o Wq[p,o] Wc[o,i,y,x]; conv_out's
        adapter composes the same way (B' = Wq.B), quant_conv itself is never a LoRA target."""
        key = ("conv", "encoder.conv_out+quant_conv", None)
        if key not in self.cache:
            assert "quant_conv" not in self._adapters
            wc, bc = self.base("encoder.conv_out")
            wq, bq = self.base("quant_conv")
            wq2 = wq[:, :, 0, 0]
            w = torch.einsum("po,oiyx->piyx", wq2, wc)
            b = wq2 @ bc + bq
            ab = self.lora("encoder.conv_out")
            A = B = None
            if ab is not None:
                A, B = self._conv_to_k(ab[0]), wq2 @ ab[1]
            self.cache[key] = dict(w=self._pack_rows(self._conv_to_k(w), A, B), b=self._up(b, torch.float32), n=w.shape[0], ks=3)
        return self.cache[key]


def C_addr(p):
    import ctypes
    return ctypes.addressof(p)
