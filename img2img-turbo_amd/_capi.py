"""ctypes binding of include/i2i_turbo.h (the C ABI of the gfx950 kernels).

The product loads exactly one library: the hipcc-built ``csrc/libi2i_turbo.so`` that sits in-tree.
There is NO fallback: if the library is missing or does not export the ABI, import-time use raises.
(Tests may pass an explicit path to a differently built twin, e.g. the CPU emulator under
tests/emu/; the product never does.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "csrc", "libi2i_turbo.so")

F32, BF16, F16, U8 = 0, 1, 2, 3
OP_IGEMM, OP_GN_STATS, OP_LAYERNORM, OP_SOFTMAX = 1, 2, 3, 4
OP_NCHW_TO_NHWC, OP_NHWC_TO_NCHW, OP_POSTERIOR, OP_DDPM_POSTQUANT, OP_ATTENTION, OP_GN_APPLY, OP_EMBED = 5, 6, 7, 8, 9, 10, 11
OP_LORA_MERGE = 12
OP_RESIZE_U8 = 13
OP_NOP = 14

i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p
ABI_VERSION = 10         # include/i2i_turbo.h I2I_ABI_VERSION this binding was written for


class IgemmParams(C.Structure):
    _fields_ = [("a0", vp), ("a1", vp), ("c0", i32), ("c1", i32), ("lda0", i32), ("lda1", i32),
                ("a_bs_b", i64), ("a_bs_h", i64), ("nimg", i32), ("hin", i32), ("win", i32),
                ("ho", i32), ("wo", i32), ("ks", i32), ("stride", i32), ("pad", i32), ("ups", i32),
                ("b", vp), ("ldb", i32), ("b_bs_b", i64), ("b_bs_h", i64),
                ("M", i32), ("N", i32), ("K", i32), ("gn_ss", vp), ("act", i32),
                ("bias", vp), ("bias_mode", i32), ("alpha", f32),
                ("res", vp), ("ldr", i32), ("r_bs_b", i64), ("r_bs_h", i64),
                ("c", vp), ("ldc", i32), ("c_bs_b", i64), ("c_bs_h", i64),
                ("zcount", i32), ("zh_count", i32), ("geglu", i32), ("out_f32", i32), ("tile", i32),
                ("splitk", i32), ("ws", vp), ("gn_part", vp), ("gn_part_groups", i32), ("subpix", i32), ("act_out", i32), ("up_h", i32), ("up_w", i32),
                ("k2_a", vp), ("k2_b", vp), ("k2_c", i32), ("k2_lda", i32), ("k2_ldb", i32),
                ("ln_cs", vp), ("ln_eps", f32), ("n_trans", i32), ("c2", vp), ("ldc2", i32)]


class GnStatsParams(C.Structure):
    _fields_ = [("x0", vp), ("x1", vp), ("c0", i32), ("c1", i32), ("ld0", i32), ("ld1", i32),
                ("nimg", i32), ("hw", i32), ("groups", i32), ("eps", f32),
                ("gamma", vp), ("beta", vp), ("partial", vp), ("nparts", i32), ("ss", vp), ("finalize_only", i32), ("counters", vp)]


class GnApplyParams(C.Structure):
    _fields_ = [("x", vp), ("y", vp), ("ss", vp), ("nimg", i32), ("hw", i32), ("c", i32), ("act", i32),
                ("ldx", i32), ("ldy", i32), ("ss_ld", i32), ("ss_off", i32), ("x1", vp), ("c1", i32), ("ldx1", i32)]


class LayerNormParams(C.Structure):
    _fields_ = [("x", vp), ("y", vp), ("gamma", vp), ("beta", vp), ("rows", i32), ("c", i32),
                ("ldx", i32), ("ldy", i32), ("eps", f32)]


class SoftmaxParams(C.Structure):
    _fields_ = [("s", vp), ("p", vp), ("rows", i64), ("cols", i32), ("lds", i32), ("ldp", i32), ("scale", f32)]


class AttentionParams(C.Structure):
    _fields_ = [("q", vp), ("k", vp), ("vt", vp), ("o", vp),
                ("batch", i32), ("heads", i32), ("d", i32), ("tq", i32), ("tk", i32),
                ("ldq", i32), ("ldk", i32), ("ldvt", i32), ("ldo", i32),
                ("q_bs", i64), ("k_bs", i64), ("vt_bs", i64), ("o_bs", i64), ("scale", f32), ("causal", i32),
                ("ksplit", i32), ("ws", vp)]


class EmbedParams(C.Structure):
    _fields_ = [("ids", vp), ("tok", vp), ("pos", vp), ("y", vp), ("rows", i32), ("T", i32), ("c", i32), ("vocab", i32)]


class NchwToNhwcParams(C.Structure):
    _fields_ = [("x", vp), ("y", vp), ("n", i32), ("c", i32), ("h", i32), ("w", i32), ("cpad", i32),
                ("src_dtype", i32), ("mul", f32), ("add", f32), ("binarize_below", i32)]


class NhwcToNchwParams(C.Structure):
    _fields_ = [("x", vp), ("y", vp), ("n", i32), ("c", i32), ("h", i32), ("w", i32), ("ldx", i32),
                ("dst_dtype", i32), ("clamp", i32), ("mul", f32), ("add", f32)]


class PosteriorParams(C.Structure):
    _fields_ = [("moments", vp), ("eps", vp), ("noise", vp), ("u", vp),
                ("n", i32), ("hw", i32), ("lat", i32), ("ldm", i32), ("ldu", i32), ("noise_n", i32),
                ("sf", f32), ("r", f32), ("r_dev", vp), ("u_f32", vp), ("moments_f32", i32)]


class DdpmParams(C.Structure):
    _fields_ = [("u", vp), ("e", vp), ("y", vp), ("wpq", vp), ("bpq", vp),
                ("n", i32), ("hw", i32), ("lat", i32), ("ldu", i32), ("lde", i32), ("ldy", i32),
                ("sqrt_abar", f32), ("sqrt_1m_abar", f32), ("sf", f32), ("u_f32", i32), ("e_f32", i32)]


class LoraMergeParams(C.Structure):
    _fields_ = [("dst", vp), ("w0", vp), ("a", vp), ("b", vp), ("N", i32), ("K", i32), ("rank", i32), ("use_gamma", i32), ("rg", vp),
                ("kscale", vp), ("kshift", vp), ("bias0", vp), ("colsum", vp), ("bias_out", vp)]


class ResizeU8Params(C.Structure):
    _fields_ = [("src", vp), ("dst", vp), ("n", i32), ("hin", i32), ("win", i32), ("c", i32), ("axis", i32), ("nout", i32), ("ksize", i32),
                ("bounds", vp), ("coeffs", vp)]


class NopParams(C.Structure):
    _fields_ = [("unused", i32)]


class _OpUnion(C.Union):
    _fields_ = [("igemm", IgemmParams), ("gn_stats", GnStatsParams), ("gn_apply", GnApplyParams),
                ("layernorm", LayerNormParams), ("softmax", SoftmaxParams), ("attention", AttentionParams),
                ("to_nhwc", NchwToNhwcParams), ("to_nchw", NhwcToNchwParams), ("embed", EmbedParams),
                ("posterior", PosteriorParams), ("ddpm", DdpmParams), ("lora_merge", LoraMergeParams), ("resize_u8", ResizeU8Params),
                ("nop", NopParams)]


class Op(C.Structure):
    _fields_ = [("opcode", i32), ("dtype", i32), ("u", _OpUnion)]


_FIELD_OF = {OP_IGEMM: "igemm", OP_GN_STATS: "gn_stats", OP_GN_APPLY: "gn_apply", OP_LAYERNORM: "layernorm",
             OP_SOFTMAX: "softmax", OP_ATTENTION: "attention", OP_NCHW_TO_NHWC: "to_nhwc",
             OP_NHWC_TO_NCHW: "to_nchw", OP_POSTERIOR: "posterior", OP_DDPM_POSTQUANT: "ddpm", OP_EMBED: "embed",
             OP_LORA_MERGE: "lora_merge", OP_RESIZE_U8: "resize_u8", OP_NOP: "nop"}

EXPORTS = ["i2i_abi_version", "i2i_backend", "i2i_last_error", "i2i_sizeof_op", "i2i_igemm", "i2i_igemm_gn_parts", "i2i_igemm_route", "i2i_gn_stats",
           "i2i_gn_apply", "i2i_nop", "i2i_calib_mfma", "i2i_calib_stream", "i2i_layernorm", "i2i_softmax", "i2i_attention", "i2i_nchw_to_nhwc",
           "i2i_nhwc_to_nchw", "i2i_posterior", "i2i_ddpm_postquant", "i2i_embed", "i2i_lora_merge", "i2i_resize_u8", "i2i_run", "i2i_run_timed",
           "i2i_graph_create", "i2i_graph_launch", "i2i_graph_destroy",
           "i2i_plan_load", "i2i_plan_io", "i2i_plan_write", "i2i_plan_read", "i2i_plan_ops", "i2i_plan_run", "i2i_plan_destroy"]


class I2IError(RuntimeError):
    pass


def _preload_torch_hip_runtime():
    """PyTorch-ROCm ships its own libamdhip64 (same SONAME as /opt/rocm's).  Streams, events and graphs are only
    meaningful inside ONE HIP runtime instance, so make sure torch's copy is the one already mapped when the
    kernel library's DT_NEEDED entry is resolved (the loader then binds to it by SONAME)."""
    import torch
    hip = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(hip):
        C.CDLL(hip, mode=C.RTLD_GLOBAL)


def _assert_single_hip_runtime():
    paths = set()
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    paths.add(os.path.realpath(line.split()[-1]))
    except OSError:
        return
    if len(paths) > 1:
        raise I2IError("two HIP runtimes are mapped in this process (%s): import torch before loading the kernel "
                       "library so both share one runtime" % sorted(paths))


class Library:
    """One loaded libi2i_turbo.so.  ``path`` defaults to the in-tree product build."""

    def __init__(self, path=None):
        path = path or os.environ.get("I2I_LIB") or DEFAULT_LIB      # I2I_LIB: an experiment build (csrc/build.py --tag)
        if not os.path.exists(path):
            raise I2IError(
                "HIP kernel library not found at %s -- build it with `python __graft_entry__.py` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % path)
        self.path = path
        _preload_torch_hip_runtime()
        self.lib = C.CDLL(path)
        _assert_single_hip_runtime()
        missing = [s for s in EXPORTS if not hasattr(self.lib, s)]
        if missing:
            raise I2IError("library %s lacks ABI symbols: %s" % (path, missing))
        L = self.lib
        L.i2i_backend.restype = C.c_char_p
        L.i2i_last_error.restype = C.c_char_p
        L.i2i_sizeof_op.restype = C.c_size_t
        for name in ("i2i_igemm", "i2i_gn_stats", "i2i_gn_apply", "i2i_layernorm", "i2i_softmax", "i2i_attention",
                     "i2i_nchw_to_nhwc", "i2i_nhwc_to_nchw", "i2i_posterior", "i2i_ddpm_postquant", "i2i_embed", "i2i_lora_merge", "i2i_resize_u8"):
            getattr(L, name).argtypes = [vp, C.c_int, vp]
            getattr(L, name).restype = C.c_int
        L.i2i_nop.argtypes = [vp]
        L.i2i_calib_mfma.argtypes = [C.c_int, C.c_int, vp, vp, vp]
        L.i2i_calib_stream.argtypes = [vp, vp, C.c_size_t, vp]
        L.i2i_igemm_gn_parts.argtypes = [vp, C.c_int, C.c_int]
        L.i2i_igemm_gn_parts.restype = C.c_int
        L.i2i_igemm_route.argtypes = [vp, C.c_int]
        L.i2i_igemm_route.restype = C.c_char_p
        L.i2i_run.argtypes = [vp, C.c_int, vp]
        L.i2i_run_timed.argtypes = [vp, C.c_int, vp, vp]
        L.i2i_graph_create.argtypes = [vp, C.c_int, C.POINTER(vp)]
        L.i2i_graph_launch.argtypes = [vp, vp]
        L.i2i_graph_destroy.argtypes = [vp]
        L.i2i_plan_load.argtypes = [C.c_char_p, C.POINTER(vp)]
        L.i2i_plan_io.argtypes = [vp, C.c_char_p, C.POINTER(vp), C.POINTER(C.c_size_t)]
        L.i2i_plan_write.argtypes = [vp, C.c_char_p, vp, C.c_size_t]
        L.i2i_plan_read.argtypes = [vp, C.c_char_p, vp, C.c_size_t]
        L.i2i_plan_ops.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int)]
        L.i2i_plan_run.argtypes = [vp, vp]
        L.i2i_plan_destroy.argtypes = [vp]
        if L.i2i_abi_version() != ABI_VERSION:
            raise I2IError("ABI version mismatch")
        if L.i2i_sizeof_op() != C.sizeof(Op):
            raise I2IError("i2i_op layout mismatch: C %d vs ctypes %d" % (L.i2i_sizeof_op(), C.sizeof(Op)))
        self.backend = L.i2i_backend().decode()

    def check(self, rc):
        if rc != 0:
            raise I2IError("i2i error %d: %s" % (rc, self.lib.i2i_last_error().decode()))

    # ---- programs -------------------------------------------------------------------------------
    def igemm_gn_parts(self, params, dtype_code, groups):
        """Partial-sum slots per image the igemm op would write through gn_part (0 = its kernel cannot)."""
        return int(self.lib.i2i_igemm_gn_parts(C.addressof(params), dtype_code, groups))

    def igemm_route(self, params, dtype_code):
        """Kernel family i2i_igemm() will run this op on (reporting only)."""
        return self.lib.i2i_igemm_route(C.addressof(params), dtype_code).decode()

    def run(self, prog, stream=0):
        self.check(self.lib.i2i_run(C.addressof(prog.array), prog.n, vp(stream)))

    def run_timed(self, prog, stream=0):
        ms = (C.c_float * prog.n)()
        self.check(self.lib.i2i_run_timed(C.addressof(prog.array), prog.n, vp(stream), C.addressof(ms)))
        return list(ms)

    def graph_create(self, prog):
        g = vp()
        self.check(self.lib.i2i_graph_create(C.addressof(prog.array), prog.n, C.byref(g)))
        return g

    def graph_launch(self, g, stream=0):
        self.check(self.lib.i2i_graph_launch(g, vp(stream)))

    def graph_destroy(self, g):
        self.lib.i2i_graph_destroy(g)

    # ---- plan files (include/i2i_turbo.h, csrc/plan_file.hip; written by plan_file.export_plan) ----
    def plan_load(self, path):
        h = vp()
        self.check(self.lib.i2i_plan_load(str(path).encode(), C.byref(h)))
        return h

    def plan_io(self, h, name):
        ptr, n = vp(), C.c_size_t()
        self.check(self.lib.i2i_plan_io(h, name.encode(), C.byref(ptr), C.byref(n)))
        return ptr.value, n.value

    def plan_write(self, h, name, host_tensor):
        t = host_tensor.contiguous()
        assert t.device.type == "cpu"
        self.check(self.lib.i2i_plan_write(h, name.encode(), vp(t.data_ptr()), t.numel() * t.element_size()))

    def plan_read(self, h, name, host_tensor):
        assert host_tensor.device.type == "cpu" and host_tensor.is_contiguous()
        self.check(self.lib.i2i_plan_read(h, name.encode(), vp(host_tensor.data_ptr()), host_tensor.numel() * host_tensor.element_size()))
        return host_tensor

    def plan_run(self, h, stream=0):
        self.check(self.lib.i2i_plan_run(h, vp(stream)))

    def plan_destroy(self, h):
        self.lib.i2i_plan_destroy(h)


class Program:
    """A flat, immutable-once-frozen array of ops (what i2i_run consumes)."""

    def __init__(self):
        self.ops = []      # (opcode, dtype, params struct, label)
        self.array = None
        self.n = 0
        self.labels = []
        self.keep = []     # python objects (tensors) that own the memory the ops point at

    def add(self, opcode, dtype, params, label=""):
        assert self.array is None, "program is frozen"
        self.ops.append((opcode, dtype, params, label))

    def freeze(self):
        self.n = len(self.ops)
        self.array = (Op * max(self.n, 1))()
        for i, (opcode, dtype, params, label) in enumerate(self.ops):
            self.array[i].opcode = opcode
            self.array[i].dtype = dtype
            setattr(self.array[i].u, _FIELD_OF[opcode], params)
        self.labels = [o[3] for o in self.ops]
        return self


_DEFAULT = None


def default_library():
    """The product library (loaded once).  Raises if the HIP build is missing."""
    global _DEFAULT
    if _DEFAULT is None:
        _DEFAULT = Library()
    return _DEFAULT
