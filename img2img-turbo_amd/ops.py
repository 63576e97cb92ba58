"""Op-descriptor builders: torch tensors (as raw pointers + shapes) -> i2i_op parameter structs.

Pure plumbing.  Every function returns ``(opcode, params)`` for ``_capi.Program.add``; nothing here
computes.  Activations are NHWC / token-major tensors whose last-dim stride is 1.
"""
import torch

from . import _capi as K

DT = {torch.float32: K.F32, torch.bfloat16: K.BF16, torch.float16: K.F16}
TORCH_DT = {K.F32: torch.float32, K.BF16: torch.bfloat16, K.F16: torch.float16}


def ptr(t):
    return 0 if t is None else t.data_ptr()


def _keep(p, *tensors):
    """The descriptors carry raw pointers only; pin the tensors to the descriptor so a temporary passed by a caller
    cannot be freed while a program still references it."""
    p._keep = tuple(t for t in tensors if t is not None)
    return p


def conv(x0, w, out, *, nimg, hin, win, ho, wo, ks, stride=1, pad=0, ups=0, x1=None, c0=None, c1=0,
         lda0=None, lda1=None, N=None, ldb=None, gn_ss=None, act=0, bias=None, bias_mode=None, alpha=1.0,
         res=None, ldr=None, ldc=None, geglu=0, out_f32=0, tile=0, splitk=0, ws=None, subpix=0, up_size=None, act_out=0, k2=None):
    """Implicit-GEMM conv / linear over NHWC sources.  ``w`` is packed [N][ks*ks*(c0+c1)]."""
    p = K.IgemmParams()
    c0 = x0.shape[-1] if c0 is None else c0
    p.a0, p.a1 = ptr(x0), ptr(x1)
    p.c0, p.c1 = c0, c1
    p.lda0 = lda0 if lda0 is not None else x0.shape[-1]
    p.lda1 = (lda1 if lda1 is not None else x1.shape[-1]) if x1 is not None else 0
    p.nimg, p.hin, p.win, p.ho, p.wo = nimg, hin, win, ho, wo
    p.ks, p.stride, p.pad, p.ups = ks, stride, pad, ups
    p.b = ptr(w)
    p.K = ks * ks * (c0 + c1)
    p.ldb = ldb if ldb is not None else w.shape[-1]
    p.M = nimg * ho * wo
    p.N = N if N is not None else w.shape[0]
    p.gn_ss, p.act = ptr(gn_ss), act
    p.bias = ptr(bias)
    p.bias_mode = (1 if bias is not None else 0) if bias_mode is None else bias_mode
    p.alpha = alpha
    p.res = ptr(res)
    p.ldr = (ldr if ldr is not None else res.shape[-1]) if res is not None else 0
    p.c = ptr(out)
    p.ldc = ldc if ldc is not None else out.shape[-1]
    p.zcount, p.zh_count = 1, 1
    p.geglu, p.out_f32, p.tile = geglu, out_f32, tile
    p.splitk, p.ws, p.subpix = splitk, ptr(ws), subpix
    p.up_h, p.up_w = up_size if up_size else (0, 0)
    p.act_out = act_out
    if k2 is not None:      # (tensor [nimg][ho][wo][ld], weights [N][ldb], channels): second contraction, see i2i_igemm_params.k2_a
        k2a, k2b, k2c = k2
        p.k2_a, p.k2_b, p.k2_c, p.k2_lda, p.k2_ldb = ptr(k2a), ptr(k2b), k2c, k2a.shape[-1], k2b.shape[-1]
    return K.OP_IGEMM, _keep(p, x0, x1, w, out, gn_ss, bias, res, ws, *(k2[:2] if k2 is not None else ()))


def bgemm(a, b, out, *, M, N, Kdim, lda, ldb, ldc, batch, heads, a_bs, b_bs, c_bs, alpha=1.0, out_f32=0,
          bias=None, bias_mode=0, tile=0):
    """Batched C[z] = alpha * A[z] . B[z]^T with z = (batch, head); strides are (per-batch, per-head) pairs."""
    p = K.IgemmParams()
    p.a0, p.a1, p.c0, p.c1, p.lda0, p.lda1 = ptr(a), 0, Kdim, 0, lda, 0
    p.a_bs_b, p.a_bs_h = a_bs
    p.nimg, p.hin, p.win, p.ho, p.wo = 1, 1, M, 1, M
    p.ks, p.stride, p.pad, p.ups = 1, 1, 0, 0
    p.b, p.ldb = ptr(b), ldb
    p.b_bs_b, p.b_bs_h = b_bs
    p.M, p.N, p.K = M, N, Kdim
    p.bias, p.bias_mode = ptr(bias), bias_mode
    p.alpha = alpha
    p.c, p.ldc = ptr(out), ldc
    p.c_bs_b, p.c_bs_h = c_bs
    p.zcount, p.zh_count = batch * heads, heads
    p.out_f32, p.tile = out_f32, tile
    return K.OP_IGEMM, _keep(p, a, b, out, bias)


def gn_stats(x0, gamma, beta, partial, ss, *, nimg, hw, groups, eps, nparts, x1=None, c0=None, c1=0,
             ld0=None, ld1=None, finalize_only=0, counters=None):
    p = K.GnStatsParams()
    p.counters = ptr(counters)
    p.x0, p.x1 = ptr(x0), ptr(x1)
    p.finalize_only = finalize_only
    p.c0 = x0.shape[-1] if c0 is None else c0
    p.c1 = c1
    p.ld0 = ld0 if ld0 is not None else x0.shape[-1]
    p.ld1 = (ld1 if ld1 is not None else x1.shape[-1]) if x1 is not None else 0
    p.nimg, p.hw, p.groups, p.eps = nimg, hw, groups, eps
    p.gamma, p.beta, p.partial, p.nparts, p.ss = ptr(gamma), ptr(beta), ptr(partial), nparts, ptr(ss)
    return K.OP_GN_STATS, _keep(p, x0, x1, gamma, beta, partial, ss, counters)


def gn_apply(x, y, ss, *, nimg, hw, c, act, ldx=0, ldy=0, ss_ld=0, ss_off=0, y_off=0, x1=None, c1=0, ldx1=0):
    """y[..., y_off:y_off+c] = act(x * scale + shift); ``y_off``/``ldy`` write a channel slice of a wider buffer.  ``x1`` / ``c1``: a second
    source whose channels follow x's (the concatenated input of an up-block resnet in one launch)."""
    p = K.GnApplyParams()
    p.x, p.ss, p.nimg, p.hw, p.c, p.act = ptr(x), ptr(ss), nimg, hw, c, act
    p.y = ptr(y) + y_off * y.element_size()
    p.ldx, p.ldy, p.ss_ld, p.ss_off = ldx, ldy, ss_ld, ss_off
    p.x1, p.c1, p.ldx1 = ptr(x1), (c1 if x1 is not None else 0), ldx1
    return K.OP_GN_APPLY, _keep(p, x, y, ss, x1)


def layernorm(x, y, gamma, beta, *, rows, c, eps=1e-5, ldx=None, ldy=None):
    p = K.LayerNormParams()
    p.x, p.y, p.gamma, p.beta = ptr(x), ptr(y), ptr(gamma), ptr(beta)
    p.rows, p.c, p.ldx, p.ldy, p.eps = rows, c, ldx or c, ldy or c, eps
    return K.OP_LAYERNORM, p


def softmax(s, pout, *, rows, cols, lds, ldp, scale):
    p = K.SoftmaxParams()
    p.s, p.p, p.rows, p.cols, p.lds, p.ldp, p.scale = ptr(s), ptr(pout), rows, cols, lds, ldp, scale
    return K.OP_SOFTMAX, p


def attention(q, k, vt, o, *, batch, heads, d, tq, tk, ldq, ldk, ldvt, ldo, q_bs, k_bs, vt_bs, o_bs, scale, causal=0, ksplit=0, ws=None):
    p = K.AttentionParams()
    p.ksplit, p.ws = ksplit, ptr(ws)
    p.q, p.k, p.vt, p.o = ptr(q), ptr(k), ptr(vt), ptr(o)
    p.batch, p.heads, p.d, p.tq, p.tk = batch, heads, d, tq, tk
    p.ldq, p.ldk, p.ldvt, p.ldo = ldq, ldk, ldvt, ldo
    p.q_bs, p.k_bs, p.vt_bs, p.o_bs, p.scale = q_bs, k_bs, vt_bs, o_bs, scale
    p.causal = causal
    return K.OP_ATTENTION, _keep(p, q, k, vt, o, ws)


def embed(ids, tok, pos, y, *, rows, T, c):
    """y[row] = tok[ids[row]] + pos[row % T]  (CLIP text embeddings; ids int64 on the device)."""
    p = K.EmbedParams()
    p.ids, p.tok, p.pos, p.y, p.rows, p.T, p.c, p.vocab = ptr(ids), ptr(tok), ptr(pos), ptr(y), rows, T, c, tok.shape[0]
    return K.OP_EMBED, _keep(p, ids, tok, pos, y)


def nchw_to_nhwc(x, y, *, n, c, h, w, cpad, mul=1.0, add=0.0, binarize_below=0):
    """x: NCHW float tensor, or a uint8 HWC image batch [n, h, w, c] (then y = x/255*mul + add, or with binarize_below = k:
    y = (x < k)*mul + add -- the sketch script's ``F.to_tensor(img) < 0.5`` is k = 128)."""
    p = K.NchwToNhwcParams()
    p.x, p.y, p.n, p.c, p.h, p.w, p.cpad = ptr(x), ptr(y), n, c, h, w, cpad
    p.src_dtype, p.mul, p.add = (K.U8 if x.dtype == torch.uint8 else DT[x.dtype]), mul, add
    p.binarize_below = int(binarize_below)
    return K.OP_NCHW_TO_NHWC, p


def nhwc_to_nchw(x, y, *, n, c, h, w, ldx, clamp=0, mul=0.0, add=0.0):
    """y: NCHW float tensor, or a uint8 HWC image batch [n, h, w, c] (then y = trunc(clamp01(x*mul+add)*255))."""
    p = K.NhwcToNchwParams()
    p.x, p.y, p.n, p.c, p.h, p.w, p.ldx = ptr(x), ptr(y), n, c, h, w, ldx
    p.dst_dtype, p.clamp = (K.U8 if y.dtype == torch.uint8 else DT[y.dtype]), clamp
    p.mul, p.add = mul, add
    return K.OP_NHWC_TO_NCHW, p


def posterior(moments, eps, u, *, n, hw, lat, ldm, ldu, sf, r=1.0, noise=None, noise_n=0, u_f32=None, moments_f32=0, r_dev=None):
    p = K.PosteriorParams()
    p.r_dev = ptr(r_dev)
    p.moments, p.eps, p.noise, p.u = ptr(moments), ptr(eps), ptr(noise), ptr(u)
    p.n, p.hw, p.lat, p.ldm, p.ldu, p.noise_n, p.sf, p.r = n, hw, lat, ldm, ldu, noise_n, sf, r
    p.u_f32, p.moments_f32 = ptr(u_f32), moments_f32
    return K.OP_POSTERIOR, p


def ddpm_postquant(u, e, y, wpq, bpq, *, n, hw, lat, ldu, lde, ldy, sqrt_abar, sqrt_1m_abar, sf, u_f32=0, e_f32=0):
    p = K.DdpmParams()
    p.u, p.e, p.y, p.wpq, p.bpq = ptr(u), ptr(e), ptr(y), ptr(wpq), ptr(bpq)
    p.n, p.hw, p.lat, p.ldu, p.lde, p.ldy = n, hw, lat, ldu, lde, ldy
    p.sqrt_abar, p.sqrt_1m_abar, p.sf, p.u_f32, p.e_f32 = sqrt_abar, sqrt_1m_abar, sf, u_f32, e_f32
    return K.OP_DDPM_POSTQUANT, p
