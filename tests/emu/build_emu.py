"""Build the CPU emulator twin of libi2i_turbo.so from the SAME kernel sources (test infrastructure).

    python tests/emu/build_emu.py

Host clang compiles img2img-turbo_amd/csrc/*.hip as plain C++ with tests/emu/hip_emu.h force-included.
Output: tests/emu/build/libi2i_turbo_emu.so (git-ignored).  Never loaded by the product.
"""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "img2img-turbo_amd", "csrc")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
# "<file>#<n>": conv3x3_w32.hip as four translation units (-DW32_PART=n, one family of instantiations each: csrc/build.py) -- the
# whole file in one unit is 20 minutes of host clang
SOURCES = ["igemm.hip", "conv3x3.hip", "conv3x3_w32.hip#0", "conv3x3_w32.hip#1", "conv3x3_w32.hip#2", "conv3x3_w32.hip#3", "gemm_dma.hip", "gemm_w32.hip#0", "gemm_w32.hip#1",
           "conv_narrow.hip", "norm.hip", "elementwise.hip", "lora_merge.hip", "resize.hip", "attention.hip", "capi.hip", "calib.hip", "plan_file.hip"]
OUT = os.path.join(HERE, "build", "libi2i_turbo_emu.so")
FLAGS = ["-std=c++17", "-O2", "-fPIC", "-ffp-contract=off", "-I", HERE, "-I", CSRC, "-include", os.path.join(HERE, "hip_emu.h"),
         "-Wno-unused-function", "-Wno-unknown-attributes"]


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def _cc(src_path, obj, lang_cxx, extra=()):
    deps = [src_path, os.path.join(HERE, "hip_emu.h"), os.path.join(CSRC, "i2i_dev.h"), os.path.join(CSRC, "launch.h"),
            os.path.join(ROOT, "include", "i2i_turbo.h")]
    if _stale(obj, deps):
        cmd = [CLANG] + FLAGS + list(extra) + (["-x", "c++"] if lang_cxx else []) + ["-c", src_path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("emu compile failed for %s:\n%s" % (src_path, r.stderr))
    return obj


def build(tag=None, extra=()):
    """tag/extra: a second emulator library with additional defines (the LOGIC of an experiment build's source variants)."""
    bdir = os.path.join(HERE, "build" + ("_" + tag if tag else ""))
    out = OUT if not tag else os.path.join(bdir, "libi2i_turbo_emu_%s.so" % tag)
    os.makedirs(bdir, exist_ok=True)
    jobs = []
    for s in SOURCES:
        f, _, n = s.partition("#")
        part = ("-DW32_PART=" + n,) if n else ()
        jobs.append((os.path.join(CSRC, f), os.path.join(bdir, f.replace(".hip", (".p%s" % n if n else "") + ".emu.o")), True, tuple(extra) + part))
    jobs += [(os.path.join(HERE, "hip_emu.cpp"), os.path.join(bdir, "hip_emu.o"), False, extra),
             (os.path.join(HERE, "runtime_emu.cpp"), os.path.join(bdir, "runtime_emu.o"), False, extra)]
    with cf.ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(lambda j: _cc(*j), jobs))
    if _stale(out, objs):
        r = subprocess.run([CLANG, "-shared", "-fPIC", "-o", out] + objs, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("emu link failed:\n" + r.stderr)
    return out


if __name__ == "__main__":
    print(build())
