"""Build libi2i_turbo.so (gfx950) in-tree with hipcc.  No GPU needed: hipcc cross-compiles.

    python img2img-turbo_amd/csrc/build.py [--force] [-j N]

One object per .hip source (parallel), then one shared library next to this file.  The .so is
git-ignored but travels to the GPU box with the gpurun snapshot (objects do not: .gpurunignore).

Staleness is decided by CONTENT, not by mtime (a snapshot copy does not keep mtimes): every object and the library
carry a sidecar ``<file>.srchash`` = sha256 of (the source, the shared headers, the flags).  ``build()`` prints one
line per object saying whether it was compiled or reused, and one for the link, so a log shows what a call really did.
"""
import argparse
import concurrent.futures as cf
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
# "<file>" or "<file>#<n>": conv3x3_w32.hip is compiled as four translation units, one (dtype, tile) family of instantiations each
# (-DW32_PART=n; the whole file in one unit is 4 minutes of hipcc), in parallel with everything else
SOURCES = ["igemm.hip", "conv3x3.hip", "conv3x3_w32.hip#0", "conv3x3_w32.hip#1", "conv3x3_w32.hip#2", "conv3x3_w32.hip#3", "gemm_dma.hip", "gemm_w32.hip#0", "gemm_w32.hip#1",
           "conv_narrow.hip", "norm.hip", "elementwise.hip", "lora_merge.hip", "resize.hip", "attention.hip", "capi.hip", "calib.hip", "plan_file.hip", "runtime_hip.hip"]


def _split(src):
    """'file.hip#n' -> (file.hip, ['-DW32_PART=n'], 'file_pn.o'); 'file.hip' -> (file.hip, [], 'file.o')."""
    if "#" in src:
        f, n = src.split("#")
        return f, ["-DW32_PART=" + n], f.replace(".hip", "_p%s.o" % n)
    return src, [], src.replace(".hip", ".o")
HEADERS = ["i2i_dev.h", "launch.h", os.path.join("..", "..", "include", "i2i_turbo.h")]
LIB = os.path.join(HERE, "libi2i_turbo.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]
# per-source flags.  conv3x3_w32 / gemm_w32: VALU work runs in the MFMA shadow at one wave per SIMD, where packed-f32
# VALU (v_pk_mul_f32 / v_pk_add_f32, what SLP vectorisation makes of adjacent scalar ops) costs more than two scalar ops
# (MI355X_MICROARCH.md "price of one filler beside MFMAs")
EXTRA_FLAGS = {"conv3x3_w32.hip": ["-fno-slp-vectorize"], "gemm_w32.hip": ["-fno-slp-vectorize"]}


def _digest(paths, extra=()):
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    for e in extra:
        h.update(str(e).encode())
    return h.hexdigest()


def _recorded(path):
    try:
        with open(path + ".srchash") as f:
            return f.read().strip()
    except OSError:
        return None


def _record(path, digest):
    with open(path + ".srchash", "w") as f:
        f.write(digest + "\n")


def source_digest(src, defs=()):
    """What an object depends on: its source, the shared headers, the flags."""
    f, part, _ = _split(src)
    return _digest([os.path.join(HERE, f)] + [os.path.join(HERE, h) for h in HEADERS], FLAGS + EXTRA_FLAGS.get(f, []) + part + list(defs))


def library_digest(defs=()):
    files = sorted({_split(s)[0] for s in SOURCES})
    return _digest([os.path.join(HERE, f) for f in files] + [os.path.join(HERE, h) for h in HEADERS],
                   FLAGS + [EXTRA_FLAGS.get(_split(s)[0], []) + _split(s)[1] for s in SOURCES] + list(defs))


def _compile(src, force, bdir="build", defs=()):
    f, part, oname = _split(src)
    obj = os.path.join(HERE, bdir, oname)
    want = source_digest(src, defs)
    if not force and os.path.exists(obj) and _recorded(obj) == want:
        return obj, "reused"
    cmd = ["hipcc"] + FLAGS + EXTRA_FLAGS.get(f, []) + part + list(defs) + ["-c", os.path.join(HERE, f), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    _record(obj, want)
    return obj, "compiled"


def build(force=False, jobs=None, tag=None, defs=(), verbose=True):
    """tag/defs: an EXPERIMENT build next to the product one (libi2i_turbo_<tag>.so, objects in build_<tag>/), e.g.
    ``build.py --tag trace --defs=-DI2I_TRACE=1`` (the conv kernels' cycle tracer); load it with I2I_LIB=<path> (img2img_turbo_amd._capi)."""
    bdir = "build" + ("_" + tag if tag else "")
    lib = LIB if not tag else os.path.join(HERE, "libi2i_turbo_%s.so" % tag)
    say = (lambda *a: print("[build]", *a, flush=True)) if verbose else (lambda *a: None)
    want = library_digest(defs)
    if not force and os.path.exists(lib) and _recorded(lib) == want:
        say("%s: reused (library matches the sources: %s)" % (os.path.basename(lib), want[:16]))
        return lib
    os.makedirs(os.path.join(HERE, bdir), exist_ok=True)
    with cf.ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 4)) as ex:
        res = list(ex.map(lambda s: _compile(s, force, bdir, defs), SOURCES))
    for s, (_, how) in zip(SOURCES, res):
        say("%-18s %s" % (s, how))
    cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + [o for o, _ in res]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    _record(lib, want)
    say("%s: linked (%s)" % (os.path.basename(lib), want[:16]))
    return lib


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("-j", type=int, default=None)
    ap.add_argument("--tag", default=None, help="experiment build: libi2i_turbo_<tag>.so")
    ap.add_argument("--defs", default="", help="extra hipcc flags of the experiment build, e.g. -DI2I_TRACE=1")
    a = ap.parse_args()
    print(build(a.force, a.j, a.tag, a.defs.split()))
