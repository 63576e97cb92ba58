"""Plan files: save a planned forward so that a C / C++ host can run it without Python (include/i2i_turbo.h ``i2i_plan_*``,
csrc/plan_file.hip has the file layout).

    model = Pix2Pix_Turbo(weights=..., device="cuda", dtype=torch.bfloat16)
    plan = model.get_plan(8, 512, 512)
    export_plan(plan, "pix2pix_bs8_512.i2iplan")          # once, where Python and the checkpoints are

    /* C host */  i2i_plan_load(path, &p); i2i_plan_write(p, "x", ...); "ctx"; "eps"; i2i_plan_run(p, stream); i2i_plan_read(p, "out", ...);

What is saved: the op program (``plan.prog``: every launch with its tile / route decisions already taken), one buffer record per
torch storage an op pointer refers to, and a relocation table (op, pointer field) -> (buffer, offset).  Buffers of the plan's
recycling pool and its boundary buffers are *scratch* (size only, zero-filled at load); everything else -- packed weights with the LoRA
merged at the current scale, norm parameters, device scalars, ticket counters -- is saved with its contents.  The boundary buffers get
names: "x", "ctx", "eps", "noise" (stochastic plans), "out".

This replaces the reference's ``model(...)`` call (src/pix2pix_turbo.py:186-219, src/cyclegan_turbo.py:241-254) for hosts that are not
Python; the planner itself (route queries, tile choices, buffer recycling) stays in plan.py and runs once, at export.
"""
import bisect
import ctypes as C
import struct

import torch

from . import _capi as K

MAGIC = b"I2IPLAN1"


def _pointer_fields():
    """{opcode: [(byte offset inside i2i_op, field name)]} for every pointer-typed member of that opcode's parameter struct."""
    out = {}
    u_off = K.Op.u.offset
    for opcode, member in K._FIELD_OF.items():
        st = dict(K._OpUnion._fields_)[member]
        out[opcode] = [(u_off + getattr(st, name).offset, name) for name, ct in st._fields_ if ct is K.vp]
    return out


def _storage_key(t):
    s = t.untyped_storage()
    return s.data_ptr(), s.nbytes()


def export_plan(plan, path, extra_tensors=()):
    """Write ``plan`` (a ForwardPlan) to ``path``.  Returns a small summary dict.  The plan must not be replaying while this runs
    (the buffers' contents are read back through torch).

    The packed weights are SHARED by every plan of a model and hold whatever LoRA / skip scale ``r`` was merged last (another plan's
    ``r``, or 1.0 after a deterministic plan ran): ``plan._prepare()`` re-merges them at THIS plan's ``r`` before anything is read
    back, and refuses a plan whose buffers were released (plan-cache eviction)."""
    plan._prepare()
    named = {"x": plan.x_in, "ctx": plan.ctx, "eps": plan.eps, "out": plan.out}
    if getattr(plan, "noise", None) is not None:
        named["noise"] = plan.noise
    # (the GroupNorm scratch -- partial sums, (scale, shift) tables, ticket counters -- is patched into the ops after they were recorded:
    # plan._finish_gn_scratch; produced inside the program, zero at rest)
    gn_scratch = [t for t in (getattr(plan, n, None) for n in ("gn_partial", "gn_ss", "gn_counters")) if t is not None]
    return export_program(plan.prog, path, named, scratch=list(plan.pool.all) + gn_scratch, holders=[plan] + list(extra_tensors), device=plan.device)


def export_text_plan(encoder, batch, path):
    """The native CLIP text tower (text_encoder.ClipTextEncoder) for ``batch`` prompts as a plan file: "ids" (int64
    [batch * 77] token ids, padded / truncated as the reference's tokenizer call does, src/pix2pix_turbo.py:190-193) -> "ctx" ([batch * 77]
    [hidden] last hidden states in the encoder's dtype: what the generator plans take as their "ctx")."""
    from .text_encoder import _Plan
    tp = encoder._plans.get(batch)
    if tp is None:
        tp = encoder._plans[batch] = _Plan(encoder, batch)
    return export_program(tp.prog, path, {"ids": tp.ids, "ctx": tp.out}, scratch=list(tp._keep), holders=[tp, encoder.w], device=encoder.device)


def export_program(prog, path, named, scratch=(), holders=(), device="cpu"):
    """The file writer behind both: ``prog`` (a frozen _capi.Program), ``named`` boundary tensors, ``scratch`` tensors whose contents need
    not be saved (zero-filled at load, like the boundary buffers), ``holders``: objects / tensors that own anything else the ops point at."""
    assert prog.array is not None, "the program is not frozen"
    named = dict(named)
    scratch_keys = {_storage_key(t) for t in list(scratch) + list(named.values())}
    tensors = list(named.values()) + list(scratch)

    def harvest(obj, depth):                     # any other tensor the plan (or its packers) holds -- device scalars such as the LoRA /
        if isinstance(obj, torch.Tensor):        # skip scale r: saved with contents; only storages an op points into end up in the file
            tensors.append(obj)
        elif isinstance(obj, (list, tuple)):
            for t in obj:
                harvest(t, depth)
        elif isinstance(obj, dict):
            for t in obj.values():
                harvest(t, depth)
        elif depth > 0 and hasattr(obj, "__dict__") and type(obj).__module__.startswith(__package__):
            for t in vars(obj).values():
                harvest(t, depth - 1)
    for hld in holders:
        harvest(hld, 2)
    for _, _, params, _ in prog.ops:
        tensors += [t for t in getattr(params, "_keep", ()) if isinstance(t, torch.Tensor)]
    storages = {}                                # (ptr, nbytes) -> a tensor that owns it
    for t in tensors:
        storages.setdefault(_storage_key(t), t)
    keys = sorted(storages)                      # by address
    starts = [k[0] for k in keys]

    def locate(addr, what):
        i = bisect.bisect_right(starts, addr) - 1
        if i < 0 or addr >= keys[i][0] + max(keys[i][1], 1):
            raise K.I2IError("export_plan: %s points at 0x%x, which no tensor pinned to the program owns" % (what, addr))
        return i, addr - keys[i][0]

    # ---- relocations
    fields = _pointer_fields()
    relocs, used = [], set()
    for oi in range(prog.n):
        op = prog.array[oi]
        base = C.addressof(op)
        for off, fname in fields[op.opcode]:
            v = C.c_void_p.from_address(base + off).value
            if not v:
                continue
            b, o = locate(v, "op %d (%s) field %s" % (oi, prog.labels[oi], fname))
            relocs.append((oi, off, b, o))
            used.add(b)
    io = []
    for name, t in named.items():
        b, o = locate(t.data_ptr(), "boundary buffer " + name)
        io.append((name, b, o, t.numel() * t.element_size()))
        used.add(b)
    # ---- compact the buffer table to the storages that are referenced
    remap, bufs = {}, []
    for b in sorted(used):
        remap[b] = len(bufs)
        k = keys[b]
        bufs.append((k[1], 0 if k in scratch_keys else 1, storages[k]))
    if str(device) != "cpu":
        torch.cuda.synchronize()
    with open(path, "wb") as f:
        f.write(MAGIC + struct.pack("<6I", K.ABI_VERSION, C.sizeof(K.Op), prog.n, len(bufs), len(relocs), len(io)))
        for nbytes, kind, _ in bufs:
            f.write(struct.pack("<QII", nbytes, kind, 0))
        for name, b, o, n in io:
            f.write(struct.pack("<24sIIQQ", name.encode(), remap[b], 0, o, n))
        for oi, off, b, o in relocs:
            f.write(struct.pack("<IIIIQ", oi, off, remap[b], 0, o))
        raw = bytearray(C.string_at(C.addressof(prog.array), prog.n * C.sizeof(K.Op)))
        for oi, off, _, _ in relocs:             # pointer fields carry no meaning in the file
            p0 = oi * C.sizeof(K.Op) + off
            raw[p0:p0 + 8] = b"\0" * 8
        f.write(bytes(raw))
        data_bytes = 0
        for nbytes, kind, t in bufs:
            if kind != 1:
                continue
            s = t.untyped_storage()
            flat = torch.empty(0, dtype=torch.uint8, device=t.device).set_(s, 0, (s.nbytes(),), (1,))
            host = flat.cpu().numpy()
            f.write(host.tobytes())
            data_bytes += nbytes
    return {"ops": prog.n, "buffers": len(bufs), "relocations": len(relocs), "data_bytes": data_bytes,
            "scratch_bytes": sum(b[0] for b in bufs if b[1] == 0), "io": {n: sz for n, _, _, sz in io}}


def main(argv=None):
    """python -m img2img_turbo_amd.plan_file --out pix2pix_bs8_512.i2iplan [--model pix2pix|cyclegan] [--batch 8 --size 512 --dtype bf16]
    [--stochastic --gamma 0.4] [--direction a2b] [--u8] (--base-dir <sd-turbo snapshot> --pretrained-path <lora .pkl> | --synthetic)"""
    import argparse
    ap = argparse.ArgumentParser(description="export a planned forward to a plan file for C / C++ hosts (include/i2i_turbo.h i2i_plan_*)")
    ap.add_argument("--out", required=True)
    ap.add_argument("--model", default="pix2pix", choices=["pix2pix", "cyclegan"])
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, nargs="+", default=[512], help="H [W]")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--stochastic", action="store_true")
    ap.add_argument("--gamma", type=float, default=None, help="LoRA / skip scale r the weights are merged at (stochastic plans; default 0.4, "
                    "the reference's default: src/inference_paired.py:22)")
    ap.add_argument("--sketch", action="store_true", help="with --u8: 'x' holds raw uint8 sketches, binarised in the boundary kernel as the sketch "
                    "script does (F.to_tensor(img) < 0.5, src/inference_paired.py:56-58); implied by --stochastic --u8")
    ap.add_argument("--direction", default="a2b", choices=["a2b", "b2a"])
    ap.add_argument("--u8", action="store_true", help="uint8 NHWC boundary (to_tensor / Normalize / ToPILImage inside the boundary kernels)")
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--base-dir", default=None, help="local stabilityai/sd-turbo snapshot directory")
    ap.add_argument("--pretrained-path", default=None, help="the reference's LoRA checkpoint (.pkl)")
    ap.add_argument("--synthetic", action="store_true", help="random-init weights of the SD-Turbo architecture (benchmarks, tests)")
    ap.add_argument("--arch", default="sd-turbo", choices=["sd-turbo", "tiny"], help="with --synthetic: the test architecture instead of SD-Turbo's")
    ap.add_argument("--lib", default=None, help="another build of the kernel library (tests: the CPU emulator)")
    a = ap.parse_args(argv)
    dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[a.dtype]
    H, W = a.size[0], a.size[-1]
    kw = dict(device=a.device, dtype=dt)
    if a.lib:
        kw["lib"] = K.Library(a.lib)
    if a.synthetic:
        from . import arch
        from .synth import make_cyclegan_weights, make_pix2pix_weights
        ua, va = (arch.TINY_UNET, arch.TINY_VAE) if a.arch == "tiny" else (arch.SD_TURBO_UNET, arch.SD_TURBO_VAE)
        kw["weights"] = make_cyclegan_weights(ua, va) if a.model == "cyclegan" else make_pix2pix_weights(ua, va, seed=1234, sketch=a.stochastic)
    else:
        if not (a.base_dir and a.pretrained_path):
            ap.error("give --base-dir and --pretrained-path (or --synthetic)")
        kw.update(base_dir=a.base_dir, pretrained_path=a.pretrained_path)
    if a.model == "cyclegan":
        from .cyclegan_turbo import CycleGAN_Turbo
        model = CycleGAN_Turbo(**kw)
        u8 = (2.0, -1.0) if a.u8 else None          # CycleGAN callers normalise to [-1, 1] (src/inference_unpaired.py:42-44)
        plan = model.get_plan(a.batch, H, W, direction=a.direction, u8_io=u8)
    else:
        from .pix2pix_turbo import Pix2Pix_Turbo
        model = Pix2Pix_Turbo(**kw)
        # uint8 boundary: edge maps go through to_tensor (x / 255); the sketch model's input is the binarised sketch (threshold 128 =
        # `F.to_tensor(img) < 0.5`), which is what Pix2Pix_Turbo.forward_u8(..., sketch=True) feeds -- a stochastic plan is the sketch model
        u8 = ((1.0, 0.0, 128) if (a.sketch or a.stochastic) else (1.0, 0.0)) if a.u8 else None
        gamma = a.gamma if a.gamma is not None else (0.4 if a.stochastic else 1.0)
        plan = model.get_plan(a.batch, H, W, stochastic=a.stochastic, r=gamma, u8_io=u8)
    info = export_plan(plan, a.out)                 # (re-merges the weights at this plan's r first: plan._prepare)
    print("wrote %s: %d ops, %d buffers, %.2f GB of weights, %.2f GB of scratch at load; boundary buffers %s"
          % (a.out, info["ops"], info["buffers"], info["data_bytes"] / 1e9, info["scratch_bytes"] / 1e9, info["io"]))


if __name__ == "__main__":
    main()
