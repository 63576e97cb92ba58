"""Per-op parity checks shared by the CPU-emulator tests and the GPU tests.

Each check builds seeded inputs, runs ONE op of the HIP library through the C ABI
(``_capi.Library`` -- the real gfx950 build on a GPU box, the emulator twin on the CPU) and compares with
a plain PyTorch fp32 statement of the same op (the building blocks of oracle/nn.py).
"""
import math

import torch
import torch.nn.functional as F

from img2img_turbo_amd import _capi as K
from img2img_turbo_amd import ops as O

TOL = {torch.float32: 2e-4, torch.bfloat16: 4e-2, torch.float16: 6e-3}


def run_op(lib, opcode, params, dtype, device):
    prog = K.Program()
    prog.add(opcode, O.DT[dtype], params)
    prog.freeze()
    stream = torch.cuda.current_stream().cuda_stream if device != "cpu" else 0
    lib.run(prog, stream)
    if device != "cpu":
        torch.cuda.synchronize()


def pack_conv_weight(w, dtype, cpad=None):
    """OIHW -> [O][KH*KW*Ipad] (k = (ky,kx,ci), ci contiguous)."""
    o, i, kh, kw = w.shape
    ip = cpad or ((i + 7) // 8 * 8)
    wp = torch.zeros(o, kh, kw, ip, dtype=torch.float32)
    wp[..., :i] = w.permute(0, 2, 3, 1)
    return wp.reshape(o, kh * kw * ip).to(dtype).contiguous()


def nhwc(x, dtype, cpad=None):
    n, c, h, w = x.shape
    cp = cpad or ((c + 7) // 8 * 8)
    y = torch.zeros(n, h, w, cp, dtype=torch.float32)
    y[..., :c] = x.permute(0, 2, 3, 1)
    return y.to(dtype).contiguous()


def rel_err(got, ref):
    return float((got.float() - ref.float()).abs().max() / (ref.float().abs().max() + 1e-12))


def gn_scale_shift(x, groups, gamma, beta, eps):
    """fp32 reference of the (scale, shift) pairs i2i_gn_stats must produce. x: NCHW."""
    n, c, h, w = x.shape
    xg = x.reshape(n, groups, -1)
    mean = xg.mean(-1)
    var = xg.var(-1, unbiased=False)
    rstd = torch.rsqrt(var + eps)
    cpg = c // groups
    sc = rstd.repeat_interleave(cpg, 1) * gamma[None]
    sh = beta[None] - mean.repeat_interleave(cpg, 1) * sc
    return torch.stack([sc, sh], -1).contiguous()  # [n][c][2]


def check_conv(lib, device, dtype, *, n=2, cin=16, cout=32, h=10, w=12, ks=3, stride=1, pad=1, ups=0,
               cin2=0, gn=False, act=0, bias=True, res=False, alpha=1.0, asym_pad=False, tile=0, seed=0, groups=4, splitk=0, subpix=False, k2c=0):
    """k2c: channels of a second NHWC tensor at output resolution whose 1x1 convolution is accumulated into the same output
    (i2i_igemm_params.k2_a: the decoder's skip conv folded into the upsampler)."""
    g = torch.Generator().manual_seed(seed)
    ct = cin + cin2
    x = torch.randn(n, ct, h, w, generator=g)
    wt = torch.randn(cout, ct, ks, ks, generator=g) / math.sqrt(ct * ks * ks)
    b = torch.randn(cout, generator=g) * 0.1 if bias else None
    xin = x.to(dtype).float()  # what the kernel sees
    ref_in = xin
    ss = None
    if gn:
        gamma = 1 + 0.1 * torch.randn(ct, generator=g)
        beta = 0.1 * torch.randn(ct, generator=g)
        ss = gn_scale_shift(xin, groups, gamma, beta, 1e-5)
        ref_in = xin * ss[:, :, 0][:, :, None, None] + ss[:, :, 1][:, :, None, None]
        if act:
            ref_in = F.silu(ref_in)
        ref_in = ref_in.to(dtype).float()  # kernel rounds the transformed operand to dtype
    if ups:
        ref_in = F.interpolate(ref_in, scale_factor=2.0, mode="nearest")
    wq = wt.to(dtype).float()
    if asym_pad:
        ref = F.conv2d(F.pad(ref_in, (0, 1, 0, 1)), wq, b, stride=stride)
        kpad = 0
    else:
        ref = F.conv2d(ref_in, wq, b, stride=stride, padding=pad)
        kpad = pad
    ref = ref * alpha if b is None else (ref - b[None, :, None, None]) * alpha + b[None, :, None, None]
    ho, wo = ref.shape[-2:]
    r = None
    if res:
        r = torch.randn(n, cout, ho, wo, generator=g)
        ref = ref + r.to(dtype).float()
    k2 = None
    if k2c:
        sk = torch.randn(n, k2c, ho, wo, generator=g)
        w2 = torch.randn(cout, k2c, generator=g) / math.sqrt(k2c)
        ref = ref + alpha * F.conv2d(sk.to(dtype).float(), w2.to(dtype).float()[:, :, None, None])
        k2 = (nhwc(sk, dtype).to(device), w2.to(dtype).contiguous().to(device), k2c)
    # device tensors
    x0 = nhwc(x[:, :cin], dtype).to(device)
    x1 = nhwc(x[:, cin:], dtype).to(device) if cin2 else None
    # weights: k = (ky,kx,[c0 | c1]) with each source padded to 8
    c0p = x0.shape[-1]
    c1p = x1.shape[-1] if cin2 else 0
    wp = torch.zeros(cout, ks, ks, c0p + c1p)
    wp[..., :cin] = wt[:, :cin].permute(0, 2, 3, 1)
    if cin2:
        wp[..., c0p:c0p + cin2] = wt[:, cin:].permute(0, 2, 3, 1)
    wp = wp.reshape(cout, -1).to(dtype).contiguous().to(device)
    if subpix:     # product packing of the sub-pixel form; the reference above stays F.conv2d on the upsampled input
        from img2img_turbo_amd.packer import subpixel_weights
        assert ups == 1 and not cin2 and c0p == cin
        wp = subpixel_weights(wt).reshape(4 * cout, 4 * cin).to(dtype).contiguous().to(device)
    ssd = None
    if gn:
        ssp = torch.zeros(n, c0p + c1p, 2)
        ssp[:, :cin] = ss[:, :cin]
        if cin2:
            ssp[:, c0p:c0p + cin2] = ss[:, cin:]
        ssd = ssp.contiguous().to(device)
    coutp = (cout + 7) // 8 * 8
    out = torch.full((n, ho, wo, coutp), float("nan"), dtype=dtype, device=device)
    rd = nhwc(r, dtype, coutp).to(device) if res else None
    bd = b.float().to(device) if bias else None
    wsd = torch.full((splitk * n * ho * wo * cout,), float("nan"), device=device) if splitk > 1 else None   # keep alive
    opcode, p = O.conv(x0, wp, out, nimg=n, hin=h, win=w, ho=ho, wo=wo, ks=ks, stride=stride, pad=kpad, ups=ups,
                       x1=x1, c0=c0p, c1=c1p, N=cout, gn_ss=ssd, act=act, bias=bd, alpha=alpha, res=rd, tile=tile,
                       splitk=splitk, ws=wsd, subpix=1 if subpix else 0, k2=k2)
    run_op(lib, opcode, p, dtype, device)
    got = out.cpu().float()[..., :cout].permute(0, 3, 1, 2)
    assert torch.isfinite(got).all(), "non-finite output"
    err = rel_err(got, ref)
    assert err < TOL[dtype], f"conv rel err {err}"
    return err


def check_geglu(lib, device, dtype, *, rows=70, cin=32, cff=64, seed=0, tile=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(rows, cin, generator=g)
    w = torch.randn(2 * cff, cin, generator=g) / math.sqrt(cin)
    b = torch.randn(2 * cff, generator=g) * 0.1
    xq, wq = x.to(dtype).float(), w.to(dtype).float()
    hgl = F.linear(xq, wq, b)
    a, gt = hgl.chunk(2, -1)
    ref = a * F.gelu(gt)
    # interleave rows per 16: [a 16][g 16]
    idx = []
    for j in range(cff // 16):
        idx += list(range(16 * j, 16 * j + 16)) + list(range(cff + 16 * j, cff + 16 * j + 16))
    idx = torch.tensor(idx)
    wp = w[idx].to(dtype).contiguous().to(device)
    bp = b[idx].float().contiguous().to(device)
    xd = x.to(dtype).contiguous().to(device)
    out = torch.full((rows, cff), float("nan"), dtype=dtype, device=device)
    opcode, p = O.conv(xd, wp, out, nimg=1, hin=1, win=rows, ho=1, wo=rows, ks=1, bias=bp, geglu=1, ldc=cff, tile=tile)
    run_op(lib, opcode, p, dtype, device)
    err = rel_err(out.cpu(), ref)
    assert err < TOL[dtype], f"geglu rel err {err}"
    return err


def check_bgemm(lib, device, dtype, *, batch=2, heads=2, M=40, N=24, Kd=64, out_f32=1, seed=0, tile=0):
    g = torch.Generator().manual_seed(seed)
    C = heads * Kd
    a = torch.randn(batch, M, C, generator=g).to(dtype)
    b = torch.randn(batch, N, C, generator=g).to(dtype)
    ref = torch.einsum("bmhk,bnhk->bhmn", a.float().view(batch, M, heads, Kd), b.float().view(batch, N, heads, Kd)) * 0.5
    ad, bd = a.to(device), b.to(device)
    out = torch.full((batch, heads, M, N), float("nan"), dtype=torch.float32 if out_f32 else dtype, device=device)
    opcode, p = O.bgemm(ad, bd, out, M=M, N=N, Kdim=Kd, lda=C, ldb=C, ldc=N, batch=batch, heads=heads,
                        a_bs=(M * C, Kd), b_bs=(N * C, Kd), c_bs=(heads * M * N, M * N), alpha=0.5, out_f32=out_f32, tile=tile)
    run_op(lib, opcode, p, dtype, device)
    err = rel_err(out.cpu(), ref)
    assert err < TOL[dtype], f"bgemm rel err {err}"
    return err


def check_gn_stats(lib, device, dtype, *, n=2, c0=32, c1=0, h=9, w=7, groups=8, nparts=3, eps=1e-5, seed=0, sliced=False):
    """sliced: hand the single-launch kernel its ticket counters (i2i_gn_stats_params.counters, ABI v7): the pixels of an image are
    cut into up to `nparts` slices, the last-arriving workgroup of an (image, group set) finalises.  Checked: same statistics,
    counters back at zero, and a second launch gives the same bits (slices are summed in slice order whoever arrives last)."""
    g = torch.Generator().manual_seed(seed)
    ct = c0 + c1
    x = torch.randn(n, ct, h, w, generator=g) * 1.5 + 0.3
    gamma = 1 + 0.1 * torch.randn(ct, generator=g)
    beta = 0.1 * torch.randn(ct, generator=g)
    xq = x.to(dtype).float()
    ref = gn_scale_shift(xq, groups, gamma, beta, eps)
    x0 = nhwc(x[:, :c0], dtype).to(device)
    x1 = nhwc(x[:, c0:], dtype).to(device) if c1 else None
    partial = torch.zeros(n * nparts * groups * 2, device=device)
    ss = torch.full((n, ct, 2), float("nan"), device=device)
    counters = torch.zeros(n * groups, dtype=torch.int32, device=device) if sliced else None
    gd, bd = gamma.to(device), beta.to(device)          # (named: the descriptor holds raw pointers, and the op runs twice below)
    opcode, p = O.gn_stats(x0, gd, bd, partial, ss, nimg=n, hw=h * w, groups=groups, eps=eps,
                           nparts=nparts, x1=x1, c0=c0, c1=c1, counters=counters)
    run_op(lib, opcode, p, dtype, device)
    err = rel_err(ss.cpu(), ref)
    assert err < 1e-4, f"gn_stats rel err {err}"
    if sliced:
        assert int(counters.abs().sum()) == 0, "the ticket counters must be left at zero"
        first = ss.clone()
        ss.fill_(float("nan"))
        run_op(lib, opcode, p, dtype, device)
        assert torch.equal(ss, first), "sliced statistics are not run-to-run identical"
        assert int(counters.abs().sum()) == 0
    return err


def check_gn_stats_offset(lib, device, dtype, *, n=2, c=32, h=48, w=40, groups=8, nparts=3, mean=100.0, std=0.1, seed=0, finalize_only=False, sliced=False):
    """GroupNorm statistics of a tensor sitting on a large offset (|mean| = 1000 sigma): E[x^2] - mu^2 in fp32 would return
    noise for the variance; the second, shifted pass over the flagged groups must bring x*scale + shift within 1e-3 of
    F.group_norm (which is two-pass).  Channels get different offsets so that groups differ.  finalize_only: the partial sums
    are handed over as a conv epilogue would (one-pass (sum, sum of squares) per part), the tensor rides along."""
    g = torch.Generator().manual_seed(seed)
    base = mean * (1 + 0.05 * torch.arange(groups).float()).repeat_interleave(c // groups)
    x = torch.randn(n, c, h, w, generator=g) * std + base[None, :, None, None]
    gamma = 1 + 0.1 * torch.randn(c, generator=g)
    beta = 0.1 * torch.randn(c, generator=g)
    xq = x.to(dtype).float()
    ref = F.group_norm(xq.double(), groups, gamma.double(), beta.double(), 1e-5).float()
    x0 = nhwc(x, dtype).to(device)
    ss = torch.full((n, c, 2), float("nan"), device=device)
    if finalize_only:
        cpg = c // groups
        xs = xq.reshape(n, groups, cpg, h * w)
        per = -(-(h * w) // nparts)
        parts = torch.zeros(n, nparts, groups, 2)
        for k in range(nparts):
            sl = xs[..., k * per:(k + 1) * per]
            parts[:, k, :, 0] = sl.sum((-1, -2))
            parts[:, k, :, 1] = (sl * sl).sum((-1, -2))
        partial = parts.reshape(-1).contiguous().to(device)
    else:
        partial = torch.zeros(n * nparts * groups * 2, device=device)
    counters = torch.zeros(n * groups, dtype=torch.int32, device=device) if sliced else None
    gd, bd = gamma.to(device), beta.to(device)
    opcode, p = O.gn_stats(x0, gd, bd, partial, ss, nimg=n, hw=h * w, groups=groups, eps=1e-5,
                           nparts=nparts, c0=c, finalize_only=1 if finalize_only else 0, counters=counters)
    run_op(lib, opcode, p, dtype, device)
    sc = ss.cpu()
    y = xq * sc[:, :, 0][:, :, None, None] + sc[:, :, 1][:, :, None, None]
    err = (y - ref).abs().max().item()
    scale = ref.abs().max().item()
    # fp32: 1e-3 of the output scale.  16-bit storage: one rounding step of the dtype the consumer applies / stores the normalised
    # value in (2^-8 bf16, 2^-11 fp16) -- csrc/norm.hip flags a group for the second pass only when the one-pass variance error
    # could exceed that (GnRefine<T>::RATIO)
    tol = {torch.float32: 1e-3, torch.float16: max(1e-3, 2.0 ** -11), torch.bfloat16: 2.0 ** -8}[dtype]
    assert torch.isfinite(y).all() and err < tol * max(scale, 1.0), f"gn_stats with offset: max-abs {err} (outputs up to {scale}, tol {tol})"
    return err


def check_layernorm(lib, device, dtype, *, rows=9, c=320, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(rows, c, generator=g) * 2 + 0.5
    gamma = 1 + 0.1 * torch.randn(c, generator=g)
    beta = 0.1 * torch.randn(c, generator=g)
    ref = F.layer_norm(x.to(dtype).float(), (c,), gamma, beta, 1e-5)
    xd = x.to(dtype).to(device)
    y = torch.full((rows, c), float("nan"), dtype=dtype, device=device)
    opcode, p = O.layernorm(xd, y, gamma.to(device), beta.to(device), rows=rows, c=c)
    run_op(lib, opcode, p, dtype, device)
    err = rel_err(y.cpu(), ref)
    assert err < TOL[dtype], f"layernorm rel err {err}"
    return err


def check_softmax(lib, device, dtype, *, rows=11, cols=77, ldp=80, seed=0):
    g = torch.Generator().manual_seed(seed)
    s = torch.randn(rows, cols, generator=g) * 4
    ref = torch.softmax(s * 0.125, -1)
    sd = s.to(device)
    pout = torch.full((rows, ldp), float("nan"), dtype=dtype, device=device)
    opcode, p = O.softmax(sd, pout, rows=rows, cols=cols, lds=cols, ldp=ldp, scale=0.125)
    run_op(lib, opcode, p, dtype, device)
    got = pout.cpu().float()
    assert (got[:, cols:] == 0).all()
    err = rel_err(got[:, :cols], ref)
    assert err < TOL[dtype], f"softmax rel err {err}"
    return err


def check_gn_apply(lib, device, dtype, *, n=2, c=64, c1=0, h=6, w=7, act=1, seed=0):
    """Standalone GroupNorm apply (+SiLU) from a (scale, shift) table; ``c1``: a second source whose channels follow the first's
    (i2i_gn_apply_params.x1, ABI v10: the concatenated input of an up-block resnet in one launch)."""
    g = torch.Generator().manual_seed(seed)
    ct = c + c1
    x = torch.randn(n, h * w, ct, generator=g).to(dtype)
    ss = torch.stack([1 + 0.2 * torch.randn(n, ct, generator=g), 0.3 * torch.randn(n, ct, generator=g)], -1).contiguous()
    ref = x.float() * ss[:, None, :, 0] + ss[:, None, :, 1]
    if act:
        ref = F.silu(ref)
    x0 = x[..., :c].contiguous().to(device)
    x1 = x[..., c:].contiguous().to(device) if c1 else None
    y = torch.full((n, h * w, ct), float("nan"), dtype=dtype, device=device)
    opcode, p = O.gn_apply(x0, y, ss.to(device), nimg=n, hw=h * w, c=c, act=act, ldy=ct, ss_ld=ct, x1=x1, c1=c1)
    run_op(lib, opcode, p, dtype, device)
    got = y.cpu().float()
    assert torch.isfinite(got).all()
    err = rel_err(got, ref)
    assert err < TOL[dtype], f"gn_apply rel err {err}"
    return err


def check_attention(lib, device, dtype, *, batch=2, heads=2, tq=70, tk=77, d=64, seed=0, spike=False, ksplit=0):
    """ksplit > 1: keys divided among ksplit workgroups per query tile + the merge launch (i2i_attention_params.ksplit, d = 512)."""
    g = torch.Generator().manual_seed(seed)
    C = heads * d
    q = torch.randn(batch, tq, C, generator=g).to(dtype)
    k = torch.randn(batch, tk, C, generator=g).to(dtype)
    v = torch.randn(batch, tk, C, generator=g).to(dtype)
    if spike:  # force a late running-max jump (rescale branch) on one query
        k[0, tk - 3, :d] = q[0, 5, :d] * 3
    qf, kf, vf = (t.float().view(batch, -1, heads, d).transpose(1, 2) for t in (q, k, v))
    ref = torch.softmax(qf @ kf.transpose(-1, -2) / math.sqrt(d), -1) @ vf
    ref = ref.transpose(1, 2).reshape(batch, tq, C)
    epc = 4 if dtype == torch.float32 else 8
    ldvt = (tk + epc - 1) // epc * epc
    vt = torch.full((batch, C, ldvt), float("nan"), dtype=dtype)  # padding deliberately poisoned
    vt[:, :, :tk] = v.transpose(1, 2)
    qd, kd, vtd = q.to(device), k.to(device), vt.to(device)
    o = torch.full((batch, tq, C), float("nan"), dtype=dtype, device=device)
    ws = torch.full((batch * heads * ksplit * tq * (d + 2),), float("nan"), device=device) if ksplit > 1 else None
    opcode, p = O.attention(qd, kd, vtd, o, batch=batch, heads=heads, d=d, tq=tq, tk=tk, ldq=C, ldk=C, ldvt=ldvt, ldo=C,
                            q_bs=tq * C, k_bs=tk * C, vt_bs=C * ldvt, o_bs=tq * C, scale=1.0 / math.sqrt(d), ksplit=ksplit, ws=ws)
    run_op(lib, opcode, p, dtype, device)
    got = o.cpu()
    assert torch.isfinite(got.float()).all()
    err = rel_err(got, ref)
    assert err < TOL[dtype], f"attention rel err {err}"
    return err


def check_boundary(lib, device, dtype, *, n=2, h=6, w=10, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(n, 3, h, w, generator=g)
    y = torch.full((n, h, w, 8), float("nan"), dtype=dtype, device=device)
    opcode, p = O.nchw_to_nhwc(x.to(device), y, n=n, c=3, h=h, w=w, cpad=8, mul=2.0, add=-1.0)
    run_op(lib, opcode, p, dtype, device)
    ref = nhwc(x * 2 - 1, dtype)
    assert torch.equal(y.cpu().float(), ref.float())
    # back
    t = (torch.randn(n, h, w, 8, generator=g) * 1.5).to(dtype)
    out = torch.full((n, 3, h, w), float("nan"), dtype=torch.float32, device=device)
    opcode, p = O.nhwc_to_nchw(t.to(device), out, n=n, c=3, h=h, w=w, ldx=8, clamp=1)
    run_op(lib, opcode, p, dtype, device)
    assert torch.equal(out.cpu(), t.float()[..., :3].permute(0, 3, 1, 2).clamp(-1, 1))
    return 0.0


def check_latent_ops(lib, device, dtype, *, n=2, h=4, w=6, seed=0, r=0.4):
    g = torch.Generator().manual_seed(seed)
    hw, lat = h * w, 4
    moments = torch.randn(n, 2 * lat, h, w, generator=g)
    eps = torch.randn(n, lat, h, w, generator=g)
    noise = torch.randn(1, lat, h, w, generator=g)
    mean, logvar = moments.chunk(2, 1)
    z = (mean + torch.exp(0.5 * logvar.clamp(-30, 20)) * eps) * 0.18215
    u_ref = z * r + noise * (1 - r)
    md = moments.permute(0, 2, 3, 1).contiguous().to(device)  # fp32 NHWC moments
    u = torch.full((n, hw, 8), float("nan"), dtype=dtype, device=device)
    u32 = torch.full((n, hw, lat), float("nan"), device=device)
    opcode, p = O.posterior(md, eps.to(device), u, n=n, hw=hw, lat=lat, ldm=2 * lat, ldu=8, sf=0.18215, r=r,
                            noise=noise.to(device), noise_n=1, u_f32=u32, moments_f32=1)
    run_op(lib, opcode, p, dtype, device)
    ref_nhwc = u_ref.permute(0, 2, 3, 1).reshape(n, hw, lat)
    assert rel_err(u32.cpu(), ref_nhwc) < 1e-5
    assert rel_err(u.cpu()[..., :lat], ref_nhwc) < TOL[dtype]
    assert (u.cpu().float()[..., lat:] == 0).all()
    # ddpm + post_quant
    e = torch.randn(n, hw, lat, generator=g)
    wpq = torch.randn(lat, lat, generator=g) * 0.5
    bpq = torch.randn(lat, generator=g) * 0.1
    sa, s1 = 0.06826489, 0.99766723
    x0 = (ref_nhwc - s1 * e) / sa / 0.18215
    ref = x0 @ wpq.t() + bpq
    y = torch.full((n, hw, 8), float("nan"), dtype=dtype, device=device)
    opcode, p = O.ddpm_postquant(u32, e.to(device), y, wpq.to(device), bpq.to(device), n=n, hw=hw, lat=lat, ldu=lat, lde=lat, ldy=8,
                                 sqrt_abar=sa, sqrt_1m_abar=s1, sf=0.18215, u_f32=1, e_f32=1)
    run_op(lib, opcode, p, dtype, device)
    err = rel_err(y.cpu()[..., :lat], ref)
    assert err < max(TOL[dtype], 1e-4) and (y.cpu().float()[..., lat:] == 0).all()
    return err


def check_conv_gn_part(lib, device, dtype, *, n=2, cin=64, cout=64, h=16, w=16, groups=8, tile=10, res=True, seed=0, subpix=False,
                       ks=3, stride=1, skip_if_declined=False):
    """3x3 conv whose epilogue emits the GroupNorm partial sums of its OUTPUT (gn_part), finished by
    gn_stats(finalize_only): the (scale, shift) pairs must match statistics taken from the stored tensor."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, ks, ks, generator=g) / math.sqrt(cin * ks * ks)
    b = torch.randn(cout, generator=g) * 0.1
    r = torch.randn(n, cout, h // stride, w // stride, generator=g) if res else None
    gamma = 1 + 0.1 * torch.randn(cout, generator=g)
    beta = 0.1 * torch.randn(cout, generator=g)
    x0 = nhwc(x, dtype).to(device)
    wp = pack_conv_weight(wt, dtype).to(device)
    ho, wo = (2 * h, 2 * w) if subpix else (h // stride, w // stride)
    if subpix:      # Upsample2D in sub-pixel form: the partial sums cover the 2h x 2w OUTPUT, four parity workgroups per tile
        from img2img_turbo_amd.packer import subpixel_weights
        wp = subpixel_weights(wt).reshape(4 * cout, 4 * cin).to(dtype).contiguous().to(device)
        r = torch.randn(n, cout, ho, wo, generator=g) if res else None
    out = torch.full((n, ho, wo, cout), float("nan"), dtype=dtype, device=device)
    rd = nhwc(r, dtype).to(device) if res else None
    bdev = b.to(device)      # keep alive: the op only holds raw pointers
    opcode, p = O.conv(x0, wp, out, nimg=n, hin=h, win=w, ho=ho, wo=wo, ks=ks, stride=stride, pad=ks // 2, ups=1 if subpix else 0, N=cout, bias=bdev,
                       res=rd, tile=tile, subpix=1 if subpix else 0)
    parts = lib.igemm_gn_parts(p, O.DT[dtype], groups)
    if parts == 0 and skip_if_declined:
        import pytest
        pytest.skip("this build of the library does not emit GroupNorm partials for this op (compile-time gated feature)")
    assert parts > 0, "kernel declined GroupNorm partials"
    part = torch.full((n * parts * groups * 2,), float("nan"), device=device)
    p.gn_part, p.gn_part_groups = part.data_ptr(), groups
    ss = torch.full((n, cout, 2), float("nan"), device=device)
    gd, bd = gamma.to(device), beta.to(device)
    op2, p2 = O.gn_stats(None, gd, bd, part, ss, nimg=n, hw=ho * wo, groups=groups, eps=1e-5, nparts=parts, c0=cout, ld0=cout, finalize_only=1)
    prog = K.Program()
    prog.add(opcode, O.DT[dtype], p)
    prog.add(op2, O.DT[dtype], p2)
    prog.freeze()
    lib.run(prog, torch.cuda.current_stream().cuda_stream if device != "cpu" else 0)
    if device != "cpu":
        torch.cuda.synchronize()
    stored = out.cpu().float().permute(0, 3, 1, 2)
    ref = gn_scale_shift(stored, groups, gamma, beta, 1e-5)
    err = rel_err(ss.cpu(), ref)
    assert err < 2e-4, f"fused gn stats rel err {err}"
    return err


def check_u8_boundary(lib, device, dtype, *, n=2, h=6, w=10, seed=0):
    """uint8 HWC <-> NHWC with the callers' pre/post-processing folded in (to_tensor / Normalize / x*0.5+0.5 / ToPILImage)."""
    g = torch.Generator().manual_seed(seed)
    img = torch.randint(0, 256, (n, h, w, 3), generator=g, dtype=torch.uint8)
    y = torch.full((n, h, w, 8), float("nan"), dtype=dtype, device=device)
    imgd = img.to(device)
    opcode, p = O.nchw_to_nhwc(imgd, y, n=n, c=3, h=h, w=w, cpad=8, mul=2.0, add=-1.0)
    run_op(lib, opcode, p, dtype, device)
    ref = (img.float() / 255.0) * 2.0 - 1.0
    got = y.cpu().float()
    assert (got[..., 3:] == 0).all()
    assert (got[..., :3] - ref.to(dtype).float()).abs().max() <= (1e-6 if dtype == torch.float32 else 8e-3)
    # the sketch script's binarisation F.to_tensor(img) < 0.5 (src/inference_paired.py:57-58): exact, every byte value
    ramp = torch.arange(256, dtype=torch.uint8).repeat(n * h * w * 3 // 256 + 1)[: n * h * w * 3].reshape(n, h, w, 3).contiguous()
    rampd = ramp.to(device)
    yb = torch.full((n, h, w, 8), float("nan"), dtype=dtype, device=device)
    opcode, p = O.nchw_to_nhwc(rampd, yb, n=n, c=3, h=h, w=w, cpad=8, binarize_below=128)
    run_op(lib, opcode, p, dtype, device)
    assert torch.equal(yb.cpu().float()[..., :3], ((ramp.float() / 255.0) < 0.5).float()) and (yb.cpu().float()[..., 3:] == 0).all()
    # output side: values on an exact 1/255 grid survive the round trip bit for bit in fp32
    x = torch.zeros(n, h, w, 8)
    x[..., :3] = ref
    x = x * 1.0
    xd = x.to(dtype).to(device)
    out = torch.zeros(n, h, w, 3, dtype=torch.uint8, device=device)
    opcode, p = O.nhwc_to_nchw(xd, out, n=n, c=3, h=h, w=w, ldx=8, clamp=1, mul=0.5, add=0.5)
    run_op(lib, opcode, p, dtype, device)
    exp = ((xd.cpu().float()[..., :3].clamp(-1, 1) * 0.5 + 0.5).clamp(0, 1) * 255.0).to(torch.uint8)
    assert (out.cpu().int() - exp.int()).abs().max() <= (0 if dtype == torch.float32 else 1)


def check_ln_gemm(lib, device, dtype, *, rows=200, cin=128, nq=160, nv=0, geglu=False, tile=52, lora_rank=4, r=0.7, seed=0, offset=0.0):
    """LayerNorm folded into the wide GEMM (i2i_igemm_params.ln_cs) together with the device-side merge that prepares its operands
    (i2i_lora_merge_params.kscale ..): out = F.linear(F.layer_norm(x), W + r B.A, b) from the UN-normalised rows, `nv` further
    output columns written transposed (the V^T of a self-attention block), or the GEGLU form.  Reference: plain fp32 torch on the
    rounded inputs, LayerNorm output NOT rounded (the kernel never materialises it)."""
    g = torch.Generator().manual_seed(seed)
    N = nq + nv
    x = (torch.randn(rows, cin, generator=g) * 1.3 + offset + 0.2 * torch.randn(rows, 1, generator=g)).to(dtype)
    W = torch.randn(N, cin, generator=g) / math.sqrt(cin)
    b = torch.randn(N, generator=g) * 0.1
    A = torch.randn(lora_rank, cin, generator=g) / math.sqrt(cin)
    Bm = torch.randn(N, lora_rank, generator=g) * 0.1
    gamma = 1 + 0.2 * torch.randn(cin, generator=g)
    beta = 0.2 * torch.randn(cin, generator=g)
    Wm = W + r * (Bm @ A)
    if geglu:
        half = N // 2
        idx = torch.arange(half).reshape(-1, 16)
        idx = torch.cat([idx, idx + half], 1).reshape(-1)
        Wp, bp, Bp = W[idx], b[idx], Bm[idx]
    else:
        Wp, bp, Bp = W, b, Bm
    dev = lambda t, dt=torch.float32: t.to(dt).contiguous().to(device)
    wd = torch.zeros(N, cin, dtype=dtype, device=device)
    cs = torch.full((N,), float("nan"), device=device)
    bout = torch.full((N,), float("nan"), device=device)
    rg = dev(torch.tensor([r, 1.0]))
    mp = K.LoraMergeParams()
    keep = [dev(Wp), dev(A), dev(Bp), dev(gamma), dev(beta), dev(bp)]
    mp.dst, mp.w0, mp.a, mp.b, mp.N, mp.K, mp.rank, mp.use_gamma, mp.rg = wd.data_ptr(), keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr(), N, cin, lora_rank, 0, rg.data_ptr()
    mp.kscale, mp.kshift, mp.bias0, mp.colsum, mp.bias_out = keep[3].data_ptr(), keep[4].data_ptr(), keep[5].data_ptr(), cs.data_ptr(), bout.data_ptr()
    run_op(lib, K.OP_LORA_MERGE, mp, dtype, device)
    # the merge: stored weights = cvt(W' * gamma), colsum of the STORED values, bias' = b + W'.beta
    wq = (Wm * gamma[None]).to(dtype)
    wref = wq[idx] if geglu else wq
    assert rel_err(wd.cpu(), wref) < (1e-6 if dtype == torch.float32 else 1e-2), "ln-fold merge: weights"
    assert rel_err(cs.cpu(), wd.cpu().float().sum(1)) < 1e-5, "ln-fold merge: column sums"
    bref = b + Wm @ beta
    assert rel_err(bout.cpu(), bref[idx] if geglu else bref) < 1e-5, "ln-fold merge: bias"
    # the GEMM
    xd = x.to(device)
    n_out = N // 2 if geglu else nq
    out = torch.full((rows, n_out), float("nan"), dtype=dtype, device=device)
    out2 = torch.full((max(nv, 1), rows), float("nan"), dtype=dtype, device=device)
    opcode, p = O.conv(xd, wd, out, nimg=1, hin=1, win=rows, ho=1, wo=rows, ks=1, c0=cin, lda0=cin, N=N, bias=bout, ldc=n_out, geglu=int(geglu), tile=tile)
    p.ln_cs, p.ln_eps = cs.data_ptr(), 1e-5
    if nv:
        p.n_trans, p.c2, p.ldc2 = nq, out2.data_ptr(), rows
    assert lib.igemm_route(p, O.DT[dtype]) == "gemm_w32_kernel"
    run_op(lib, opcode, p, dtype, device)
    y = F.layer_norm(x.float(), (cin,), gamma, beta, 1e-5)
    full = y @ wq.float().T / gamma.new_ones(1) * 1.0     # LN(x) . (W' gamma)^T / gamma is NOT what runs: restate exactly below
    full = ((x.float() - x.float().mean(1, keepdim=True)) * torch.rsqrt(x.float().var(1, unbiased=False, keepdim=True) + 1e-5)) @ wq.float().T + bref[None]
    if geglu:
        ref = full[:, :half] * F.gelu(full[:, half:])
        err = rel_err(out.cpu(), ref)
    else:
        err = rel_err(out.cpu(), full[:, :nq])
        if nv:
            err = max(err, rel_err(out2.cpu(), full[:, nq:].T))
    assert err < TOL[dtype], f"ln gemm rel err {err}"
    return err
