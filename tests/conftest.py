import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")
    config.addinivalue_line("markers", "slow: multi-second CPU test")


_EMU_BUILDS = {}


def _emu_builds():
    """The emulator library, compiled on first use (a cold build is minutes of host clang; the objects are cached in
    tests/emu/build/, so this only matters after a kernel source changed)."""
    if not _EMU_BUILDS:
        sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
        import build_emu
        # I2I_EMU_TAG: a second object directory (tests/emu/build_<tag>/), so a run against edited sources does not rebuild the
        # library under a test session that is still using the first one
        _EMU_BUILDS["main"] = build_emu.build(tag=os.environ.get("I2I_EMU_TAG") or None)
    return _EMU_BUILDS


@pytest.fixture(scope="session")
def emu_lib():
    """CPU emulator twin of the HIP library (tests/emu/): same kernel sources, host clang."""
    from img2img_turbo_amd import _capi
    lib = _capi.Library(_emu_builds()["main"])
    assert lib.backend == "emu"
    return lib


@pytest.fixture(scope="session")
def gpu_lib():
    """The product library on a real GPU; fails loudly (no fallback) if the HIP build is missing."""
    import torch
    from img2img_turbo_amd import _capi
    assert torch.cuda.is_available(), "gpu-marked test without a GPU"
    # host threads for the CPU oracle / the packers: the GPU boxes have hundreds of hardware threads, on which torch's default
    # (one OpenMP thread each) is far slower than a few dozen for these convolution shapes (bench.py: 16 threads ~10 s per
    # 512x512 oracle forward).  I2I_TEST_THREADS=0 keeps torch's default.
    nthr = int(os.environ.get("I2I_TEST_THREADS", "32"))
    if nthr > 0:
        torch.set_num_threads(min(nthr, os.cpu_count() or nthr))
    lib = _capi.default_library()
    assert lib.backend == "gfx950"
    # Parity is "unpinned" (DESIGN.md section 0, row c) only because the reference's dependencies are missing: say on every GPU session
    # whether this box has them -- the day one does, tests/golden/make_reference_golden.py generates the reference-made fixtures
    # that tests/test_oracle_kats.py::test_oracle_matches_reference_golden consumes.
    have = {}
    for mod in ("diffusers", "peft", "torchvision"):
        try:
            __import__(mod)
            have[mod] = True
        except Exception:
            have[mod] = False
    print("[reference deps on this box] " + ", ".join("%s: %s" % (k, "importable" if v else "absent") for k, v in have.items())
          + ("  -> run tests/golden/make_reference_golden.py here to pin the oracle" if all(have.values()) else ""))
    # the -m gpu suite certifies the PRODUCT library: an I2I_LIB override (measurement builds) must not ride along silently
    assert os.path.realpath(lib.path) == os.path.realpath(_capi.DEFAULT_LIB), "I2I_LIB points the GPU tests at %s" % lib.path
    return lib
