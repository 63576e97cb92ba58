"""Forward planner: (weights, B, H, W, dtype, mode) -> a flat op program over static NHWC buffers.

Not a walk over a diffusers module tree: the generator (src/pix2pix_turbo.py:197-203 /
src/cyclegan_turbo.py:203-206 with the patched VAE forwards src/model.py:14-54) is flattened ahead of
time into ~700 kernel launches on pre-allocated buffers, every norm / activation / residual / skip / concat /
upsample folded into a neighbouring contraction, constants at t=999 folded at pack time.  The program is
executed by one C call (``i2i_run``) or replayed as a hipGraph (``i2i_graph_launch``).

Python only allocates tensors and writes op descriptors here; it never computes.
"""
import math
import os
from dataclasses import dataclass
from typing import List, Optional

import torch

from . import _capi as K
from . import ops as O
from .packer import Packer


@dataclass
class Act:
    """An NHWC activation [n, h, w, c] living in ``t`` (channel stride 1, pixel stride = c)."""
    t: torch.Tensor
    n: int
    h: int
    w: int
    c: int
    producer: object = None     # IgemmParams of the op that wrote this tensor (GroupNorm partial-sum fusion)

    @property
    def hw(self):
        return self.h * self.w


class Pool:
    """Size-keyed recycling of device buffers: the program is static and runs in order on one stream,
    so a buffer can be handed out again as soon as the op list no longer reads it."""

    def __init__(self, device):
        self.device = device
        self.free_lists = {}
        self.all = []
        self.bytes = 0
        self.no_reuse = False

    def get(self, numel, dtype):
        key = (numel, dtype)
        fl = self.free_lists.get(key)
        if fl:
            return fl.pop()
        t = torch.empty(numel, dtype=dtype, device=self.device)
        self.all.append(t)
        self.bytes += numel * t.element_size()
        return t

    def put(self, t):
        if self.no_reuse:      # debugging aid: keep every intermediate intact after the run
            return
        self.free_lists.setdefault((t.numel(), t.dtype), []).append(t)


def one_step_scheduler_constants(t=999, n=1000, beta_start=0.00085, beta_end=0.012):
    """make_1step_sched (src/model.py:7-11): scaled_linear betas, fp32 cumprod; (sqrt(abar_t), sqrt(1-abar_t))."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2
    ac = torch.cumprod(1.0 - betas, dim=0)[t]
    return float(ac ** 0.5), float((1 - ac) ** 0.5)


class ForwardPlan:
    """One planned forward for fixed (B, H, W, dtype, stochastic?, direction).  The LoRA scale / skip gamma ``r`` is NOT part
    of the plan: it lives in the packers' device scalars (Packer.set_scale), so one plan (and its hipGraph) serves every r."""

    def __init__(self, lib, weights, B, H, W, dtype, device, *, stochastic=False, r=1.0, direction="a2b",
                 ctx_batch=1, fuse_gn=True, flash=True, out_dtype=None, packers=None, debug=False, dma_small=True, fuse_gn_stats=True, subpix=True, halo_min_tiles=160, u8_io=None, fuse_vae_attention=True,
                 unet_dtype=None):
        assert H % 8 == 0 and W % 8 == 0, "input must be a multiple of 8 (src/inference_paired.py:38-41)"
        # H, W multiples of 8 suffice (src/inference_paired.py:38-41): latent sizes that are not multiples of 8 make the
        # UNet levels odd (70 -> 35 -> 18 -> 9), handled like diffusers' forward_upsample_size path (explicit sizes).
        self.lib, self.w8s, self.B, self.H, self.W = lib, weights, B, H, W
        self.dtype, self.device = dtype, device
        self.dt = O.DT[dtype]
        # Per-network precision: `dtype` is the VAE's (activations, packed weights), `unet_dtype` the UNet's (default: the same).  The two
        # meet in fp32 (posterior moments / latents in, eps-prediction out), so e.g. a bf16 VAE (whose real activations overflow fp16)
        # can run beside an fp16 UNet (3 more mantissa bits where the 1-step scheduler multiplies the error by 14.6).
        self.vae_dtype, self.unet_dtype = dtype, (unet_dtype or dtype)
        assert (self.unet_dtype == torch.float32) == (dtype == torch.float32), "the exact-f32 mode covers both networks"

        self.stochastic, self.r = stochastic, (r if stochastic else 1.0)
        self.fuse_gn, self.flash = fuse_gn, flash
        self.fuse_gn_stats = fuse_gn_stats and not debug   # conv epilogues emit the next GroupNorm's partial sums
        self.fuse_vae_attention = fuse_vae_attention   # d = 512 flash kernel for the VAE mid-block attention (16-bit types)
        self.halo_min_tiles = halo_min_tiles   # fewer halo-conv tiles than this: LDS-DMA igemm + split-K instead
        self.subpix = subpix         # Upsample2D convs in sub-pixel form (4 parity 2x2 convs on the source plane)
        self.dma_small = dma_small   # non-halo GN convs: materialise GN and use the LDS-DMA igemm (+ split-K)
        self.ua, self.va = weights.unet_arch, weights.vae_arch
        vae_sd = weights.vae if (direction == "a2b" or weights.vae_b2a is None) else weights.vae_b2a
        if packers is None:
            packers = (Packer(weights.unet, weights.unet_scaling, self.unet_dtype, device, lib, self.r, self.r),
                       Packer(vae_sd, weights.vae_scaling, dtype, device, lib, self.r, self.r))
        self.pu, self.pv = packers
        self.pool = Pool(device)
        self.pool.no_reuse = debug
        self.taps = {}          # label -> Act of that op's output (meaningful with debug=True)
        self.prog = K.Program()
        self.op_flops = []      # algorithmic FLOPs per op (2*MAC), parallel to prog.ops
        self.op_flops_exec = [] # FLOPs the matrix pipe executes for the op (sub-pixel upsamplers: 4/9 of the 3x3 part), parallel to prog.ops
        self.op_kernel = []     # HIP kernel each op resolves to (reporting only), parallel to prog.ops
        self.op_bytes = []      # algorithmic HBM bytes per op for the HBM-bound kernels (1 read [+ 1 write] per element), parallel to prog.ops
        self.flops = 0
        self.gn_partial = None
        self.gn_ss = None
        self._gn_part_elems = 0
        self._gn_ss_elems = 0
        self._gn_cnt_elems = 0
        self._pending_gn = []
        self._keep = []         # tensors referenced by the program that are not owned by the pool
        self._ckv = None        # merged cross-attention K / V^T of the UNet (see _cross_kv)
        self.cross_kv_merged = os.environ.get("I2I_CROSS_KV_MERGED", "1") != "0"
        self.fuse_skip = os.environ.get("I2I_FUSE_SKIP", "1") != "0"       # decoder skip convs folded into the upsamplers (A/B hook)
        self.fuse_shortcut = os.environ.get("I2I_FUSE_SHORTCUT", "1") != "0"   # resnet conv_shortcut folded into conv2 (A/B hook)
        # UNet small-plane 3x3 convolutions on the wide GEMM with K slices (A/B hook), bit mask: 1 = the stride-2 downsamplers,
        # 2 = the 16 x 16 planes the halo conv would take at batch 8 (<= 2048 rows, >= 1024 channels), 4 = the planes no halo tile
        # fits or fills (8 x 8; 16 x 16 / 32 x 32 at small batch) that were split-K launches of the LDS-DMA igemm, 8 = every other 3x3
        # conv of the UNet (32 x 32 / 64 x 64 planes at batch 8: whole 160-column tiles instead of 2.5 / 5 halo channel tiles); 0 = none
        self.w32_splitk = int(os.environ.get("I2I_W32_SPLITK", "15"))
        # (rows / workgroups below which such a conv stays on the LDS-DMA igemm; the emulator tests lower them to reach the route
        # with a tiny model)
        self.w32_splitk_min_rows = int(os.environ.get("I2I_W32_SPLITK_MIN_ROWS", "512"))
        self.w32_splitk_min_wgs = int(os.environ.get("I2I_W32_SPLITK_MIN_WGS", "96"))
        self.w32_splitk_longk = int(os.environ.get("I2I_W32_SPLITK_LONGK", "32"))
        self.small_tile_rows = int(os.environ.get("I2I_SMALL_TILE_ROWS", "4096"))   # see _small_tile (A/B hooks)
        self.small_tile_k = int(os.environ.get("I2I_SMALL_TILE_K", "640"))
        self.small_tile_min_tiles = int(os.environ.get("I2I_SMALL_TILE_MIN_TILES", "128"))     # (the emulator tests lower it to reach the route)
        self.att_ksplit = os.environ.get("I2I_ATT_KSPLIT", "1") != "0"         # key-split VAE mid-block attention at small batch (A/B hook)
        self.att_ksplit64 = int(os.environ.get("I2I_ATT_KSPLIT64", "4"))      # key splits of the d = 64 flash kernel on small grids (A/B hook)
        self.att_ksplit64_min_tk = int(os.environ.get("I2I_ATT_KSPLIT64_MIN_TK", "2048"))     # (the emulator tests lower it to reach the route)
        self.att_q_log2 = os.environ.get("I2I_ATT_Q_LOG2", "1") != "0"         # scale * log2(e) folded into to_q for the flash kernel (A/B hook)
        self.vt_one_launch = os.environ.get("I2I_VT_ONE_LAUNCH", "1") != "0"   # self-attention V^T of all images in one wide-GEMM launch (A/B hook)
        # round 6: LayerNorm folded into the GEMM behind it (norm1 -> to_q | to_k | to_v^T in ONE launch, norm2 -> to_q, norm3 -> GEGLU on the
        # small planes: csrc/gemm_w32.hip LNF; 16-bit types).  A/B hook.
        self.ln_fold = os.environ.get("I2I_LN_FOLD", "1") != "0" and self.unet_dtype != torch.float32
        self.ln_fold_geglu_max_rows = int(os.environ.get("I2I_LN_FOLD_GEGLU_ROWS", "1024"))
        lat = self.va.latent_channels
        h8, w8 = H // 8, W // 8
        self.out_dtype = out_dtype or dtype
        # ---- static boundary buffers (graph-stable addresses) ----
        # u8_io = (mul, add) of the input normalisation: the boundary becomes uint8 HWC image batches on both sides
        # (row f1: F.to_tensor / Normalize / x*0.5+0.5 / ToPILImage of the callers run inside the boundary kernels)
        self.u8_io = u8_io
        self.ctx_batch = ctx_batch
        self.x_in = (torch.zeros(B, H, W, 3, dtype=torch.uint8, device=device) if u8_io else
                     torch.zeros(B, 3, H, W, dtype=torch.float32, device=device))
        self.eps = torch.zeros(B, lat, h8, w8, dtype=torch.float32, device=device)
        self.noise = torch.zeros(B, lat, h8, w8, dtype=torch.float32, device=device) if stochastic else None
        self.ctx = torch.zeros(ctx_batch, 77, self.ua.cross_attention_dim, dtype=self.unet_dtype, device=device)
        self.out = (torch.zeros(B, H, W, 3, dtype=torch.uint8, device=device) if u8_io else
                    torch.zeros(B, 3, H, W, dtype=self.out_dtype, device=device))
        self._build()
        self._finish_gn_scratch()
        self.prog.freeze()
        self.graph = None

    # ------------------------------------------------------------------ helpers
    def _as(self, dtype):
        """Context: record the ops inside with ``dtype`` as the element type (the UNet section of a mixed-precision plan)."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            old = (self.dtype, self.dt)
            self.dtype, self.dt = dtype, O.DT[dtype]
            try:
                yield
            finally:
                self.dtype, self.dt = old
        return cm()

    def _add(self, op, label, flops=0, kernel=None, flops_exec=None, nbytes=0):
        """Record one launch.  ``kernel``: which HIP kernel the C dispatcher will pick for an igemm op (i2i_igemm_route),
        used only for reporting (bench.py groups timings by kernel).  ``flops``: algorithmic FLOPs of the reference op(s)
        the launch replaces; ``flops_exec``: what the matrix pipe executes when that differs (sub-pixel upsamplers)."""
        self.prog.add(op[0], self.dt, op[1], label)
        self.op_flops.append(flops)
        self.op_flops_exec.append(flops if flops_exec is None else flops_exec)
        self.op_bytes.append(nbytes)       # algorithmic HBM bytes of the HBM-bound launches (norms, layout / latent ops); 0 = not priced
        self.op_kernel.append(kernel or {K.OP_GN_STATS: "gn_stats", K.OP_LAYERNORM: "layernorm", K.OP_SOFTMAX: "softmax",
                                        K.OP_ATTENTION: "attention_kernel", K.OP_GN_APPLY: "gn_apply",
                                        K.OP_IGEMM: "igemm_dma_kernel"}.get(op[0], "boundary/elementwise"))

    def _reroute(self, params, kernel):
        """Reporting only: the op that owns ``params`` changed its kernel after it was recorded."""
        for i, (_opc, _dt, p, _label) in enumerate(self.prog.ops):
            if p is params:
                self.op_kernel[i] = kernel

    @property
    def esz(self):
        return 4 if self.dtype == torch.float32 else 2

    def new(self, n, h, w, c, dtype=None):
        t = self.pool.get(n * h * w * c, dtype or self.dtype)
        return Act(t, n, h, w, c)

    def free(self, a):
        self.pool.put(a.t if isinstance(a, Act) else a)

    def _gn_scratch(self, nimg, ct, nparts, groups):
        self._gn_part_elems = max(self._gn_part_elems, nimg * nparts * groups * 2)
        self._gn_cnt_elems = max(self._gn_cnt_elems, nimg * groups)
        self._gn_ss_elems = max(self._gn_ss_elems, nimg * ct * 2)

    def _finish_gn_scratch(self):
        self.gn_partial = torch.zeros(max(self._gn_part_elems, 1), dtype=torch.float32, device=self.device)
        self.gn_ss = torch.zeros(max(self._gn_ss_elems, 1), dtype=torch.float32, device=self.device)
        # ticket counters of the sliced single-launch statistics kernel (csrc/norm.hip: zero before the first launch, every launch
        # leaves them zero); I2I_GN_SLICED=0 keeps one workgroup per (image, group set) (A/B hook)
        self.gn_counters = torch.zeros(max(self._gn_cnt_elems, 1), dtype=torch.int32, device=self.device)
        sliced = os.environ.get("I2I_GN_SLICED", "1") != "0"
        for p, which in self._pending_gn:   # patch pointers now that the scratch exists
            if which == "stats":
                p.partial, p.ss = self.gn_partial.data_ptr(), self.gn_ss.data_ptr()
                p.counters = self.gn_counters.data_ptr() if sliced else 0
            elif which == "stats_ss":
                p.ss = self.gn_ss.data_ptr()
            elif which == "igemm":
                p.gn_ss = self.gn_ss.data_ptr()
            else:
                p.ss = self.gn_ss.data_ptr()

    def gn_stats(self, pk, norm_name, x: Act, groups, eps, x1: Optional[Act] = None, label=""):
        gamma, beta = pk.norm(norm_name)
        ct = x.c + (x1.c if x1 else 0)
        # Fused statistics: when the tensor was written by a conv whose epilogue can emit GroupNorm partial sums
        # (i2i_igemm_gn_parts > 0), the streaming re-read of the tensor disappears; only the tiny finalize runs.
        if self.fuse_gn_stats and x1 is None and x.producer is not None and not x.producer.gn_part:
            parts = self.lib.igemm_gn_parts(x.producer, self.dt, groups)
            if parts == 0 and x.producer.tile == 0 and self.lib.igemm_route(x.producer, self.dt) == "gemm_w32_kernel":
                # the wide GEMM has no statistics epilogue: when the LDS-DMA igemm can emit the partial sums, keeping the
                # producer there is cheaper than a streaming gn_stats pass over its output
                x.producer.tile = 20
                parts = self.lib.igemm_gn_parts(x.producer, self.dt, groups)
                if parts == 0:
                    x.producer.tile = 0
                else:
                    self._reroute(x.producer, "igemm_dma_kernel")
            if parts > 0:
                # The slab below is sized for the route seen NOW.  tile == 0 re-resolves the route at every launch (the wide GEMM's
                # auto rule reads I2I_GEMM_W32 per call) and another kernel would write a different number of slots: name the
                # route in the op itself, so that an eager run()/run_timed() after an environment change cannot diverge from it.
                if x.producer.tile == 0:
                    route = self.lib.igemm_route(x.producer, self.dt)
                    pin = {"conv3x3_w32_kernel": 40, "conv3x3_w32_kernel<SUBPIX>": 40, "conv3x3_halo_kernel": 10,
                           "conv3x3_halo_kernel<SUBPIX>": 10, "igemm_dma_kernel": 20, "conv_narrow_kernel": 60}.get(route, 0)
                    if route == "gemm_w32_kernel":      # statistics come from its 128-column tiles only: 256 rows (53) or 128 rows (54)
                        pin = 53 if parts * 256 == x.hw else 54
                    if pin:
                        x.producer.tile = pin
                        if self.lib.igemm_route(x.producer, self.dt) != route or self.lib.igemm_gn_parts(x.producer, self.dt, groups) != parts:
                            x.producer.tile = 0         # (cannot name it: keep the auto route; graph replays are unaffected)
                # dedicated slab (not pooled): it is written by an op recorded EARLIER than this point, so a pooled
                # buffer could have been lent to an op in between
                part = torch.empty(x.n * parts * groups * 2, dtype=torch.float32, device=self.device)
                self._keep.append(part)
                x.producer.gn_part, x.producer.gn_part_groups = part.data_ptr(), groups
                self._gn_scratch(x.n, ct, 1, groups)
                # (the tensor still rides along: groups whose one-pass variance is cancellation-prone are re-read, csrc/norm.hip)
                op = O.gn_stats(x.t, gamma, beta, part, None, nimg=x.n, hw=x.hw, groups=groups, eps=eps, nparts=parts,
                                c0=x.c, ld0=x.c, finalize_only=1)
                self._pending_gn.append((op[1], "stats_ss"))
                self._add(op, (label or norm_name) + ".finalize", nbytes=x.n * parts * groups * 8 + x.n * ct * 8)
                return
        nparts = int(min(256, max(1, (x.hw * ct) // 65536)))
        self._gn_scratch(x.n, ct, nparts, groups)
        op = O.gn_stats(x.t, gamma, beta, None, None, nimg=x.n, hw=x.hw, groups=groups, eps=eps, nparts=nparts,
                        x1=x1.t if x1 else None, c0=x.c, c1=x1.c if x1 else 0, ld0=x.c, ld1=x1.c if x1 else 0)
        self._pending_gn.append((op[1], "stats"))
        self._add(op, label or norm_name, nbytes=x.n * x.hw * ct * self.esz + x.n * ct * 8)

    def gn_probe_ws(self, numel):
        """A 16-byte aligned fp32 slab for a route QUERY (never launched): taken from the pool and handed straight back."""
        t = self.pool.get(numel, torch.float32)
        self.pool.put(t)
        return t

    def _splitk(self, M, N, Kd):
        """Split-K factor + fp32 slab for the weight-streaming shapes (few rows, K in the thousands); mirrors the
        tile choice of launch_dma_t (csrc/gemm_dma.hip).  The slab goes back to the pool right after the op is
        recorded: the program runs in order on one stream, so later ops may reuse it."""
        bk = 32 if self.dtype == torch.float32 else 64
        if Kd % bk:
            return 0, None
        bm = 64 if M <= 2048 else 128
        bn = 128 if N > 64 else 64
        tiles = -(-M // bm) * -(-N // bn)
        sk = min(512 // max(tiles, 1), (Kd // bk) // 4, 16)
        if sk < 2:
            return 0, None
        return sk, self.pool.get(sk * M * N, torch.float32)

    def _small_tile(self, M, N, Kd):
        """A linear / 1x1 conv of few rows and a SHORT K (the UNet's 320- and 640-wide projections at batch 1: 1024 / 4096 token rows) on
        64 x 32 tiles of the LDS-DMA igemm (tile 26) when those alone fill the chip: one launch with the epilogue in it, instead of
        K slices + the reduce launch, or of ~100 large tiles.  Only up to ~10 K steps: the engine's ring is three stages deep, so an
        un-sliced K loop costs one memory latency per three steps (measured at batch 1, profiles/r6g_*: 256 rows x K = 1280 un-sliced
        17.0 us against 15.5 us sliced + reduced, K = 5120 45 against 24; 1024 rows x K = 640 13.4 against 16.0).
        I2I_SMALL_TILE_ROWS / _K = row / K limits (0 rows = off)."""
        if self.dtype == torch.float32 or M > self.small_tile_rows or Kd % 64 or N % 8 or -(-M // 64) * -(-N // 32) < self.small_tile_min_tiles:
            return 0
        # (longer K loops stay K-sliced: the same tile behind an EIGHT-stage ring, 96 KiB, ran 256 rows x K = 1280 in 18.6 us against 15.5
        # sliced + reduced and 17.0 on the 3-stage ring, +0.35 ms per batch-1 step: profiles/r6i_ab_bs1_small_tile_8_stage_ring_negative.log --
        # a lone workgroup per CU serialises wait -> barrier -> issue -> read -> MFMA per step whatever the ring holds)
        # (K up to 1280 where the tile gives >= 512 workgroups: -0.02 +- 0.014 ms at batch 1 but +0.26 +- 0.05 ms at batch 8, where it takes
        # the 2048-row x 1280 x 1280 projections from the wide GEMM: profiles/r6m_*, r6o_ab_bs8_small_tile_rules.log)
        return 26 if Kd <= self.small_tile_k else 0

    @staticmethod
    def _w32_splitk_cfg(M, N, Kd, min_wgs=96, longk=32):
        """(wide-GEMM tile id, K slices) for a small-plane 3x3 convolution; (0, 0) = leave it on the LDS-DMA igemm.
        Measured at batch 8 (profiles/r4h_bench_ops_splitk_w32.log vs _dma.log): 512 rows -> 128 x 128 tiles x 6 slices
        (1280 -> 1280 @ 8 x 8: 0.050 -> 0.034 ms), 2048 rows -> 256 x 160 x 4 when that gives 64 tiles (1280 -> 1280 @ 16 x 16:
        0.157 -> 0.081 ms) else 128 x 160 x 4 (640 @ 32 x 32 stride 2: 0.055 -> 0.035 ms), 8192 rows -> 128 x 128, one slice
        (320 @ 64 x 64 stride 2: 0.067 -> 0.033 ms); against the halo conv (r4k_*): 32768 rows x 320 -> 256 x 160, one slice (0.091 ->
        0.065 ms, 960 -> 320: 0.240 -> 0.162), 8192 rows x 640 -> 128 x 160 (0.090 -> 0.072, 1280 -> 640: 0.175 -> 0.125)."""
        if Kd % 64 or N % 8 or N < 128:
            return 0, 0
        stages = Kd // 64
        if M >= 1024 and N % 160 == 0:
            # 256 x 160 tiles when they alone fill the chip (or, with 4 slices, a 2048-row op), 128 x 160 otherwise
            cfg = 51 if -(-M // 256) * (N // 160) >= (192 if M >= 4096 else 64) else 52
        else:
            cfg = 54
        bm, bn = {51: (256, 160), 52: (128, 160), 54: (128, 128)}[cfg]
        tiles = -(-M // bm) * -(-N // bn)
        sk = 1 if tiles >= 128 else max(1, min(256 // tiles, stages // 8))      # (half a round of tiles beats slicing a short K)
        if 128 <= tiles < 256 and stages >= longk:
            # ... but not a long one: the VAE's 64 x 64-plane 512 -> 512 convolutions of a batch-1 forward (128 tiles x 72 stages) in two
            # slices: 50 -> 39 us each, -0.14 +- 0.01 ms per batch-1 step (profiles/r6l_ab_bs1_w32_longk_two_slices.log; I2I_W32_SPLITK_LONGK)
            sk = 2          # (3 / 4 slices measured worse: +0.26 / +0.18 ms, profiles/r6m_ab_bs1_*.log)
        if tiles * sk < min_wgs:
            return 0, 0
        return cfg, sk

    def upsample_conv(self, pk, name, x: Act, label, size=None, k2=None) -> Act:
        """Upsample2D: nearest-2x + 3x3 conv.  Sub-pixel form (4/9 of the MACs, csrc/conv3x3.hip SUBPIX) whenever the
        halo kernel takes it -- slab-aligned channels, source plane of at least one 8x16 tile; else the index-map gather."""
        bk = 32 if self.dtype == torch.float32 else 64
        if size is not None and tuple(size) == (2 * x.h, 2 * x.w):
            size = None
        if size is None and self.subpix and x.c % bk == 0 and x.h >= 8 and x.w >= 16:
            return self.conv(pk.conv_subpixel(name), x, ups=1, label=label, k2=k2)
        return self.conv(pk.conv(name), x, ups=1, up_size=size, label=label)   # explicit size: F.interpolate(size=...)

    def conv(self, pw, x: Act, *, ks=None, stride=1, pad=None, ups=0, up_size=None, asym=False, x1: Optional[Act] = None, gn=False, act=0,
             res: Optional[Act] = None, alpha=1.0, out: Optional[Act] = None, geglu=0, out_f32=0, cout_pad=None,
             label="", k2=None) -> Act:
        """One implicit-GEMM launch.  ``gn``: apply the pending GroupNorm scale/shift (+act) to the A operand.
        ``k2`` = dict(x=Act at output resolution, w=packed 1x1 weights, label, [bias], [fallback]): a second contraction folded
        into this launch when the kernel it routes to takes one (i2i_igemm_params.k2_a); the returned Act then has ``k2_fused``
        set; otherwise ``fallback()`` (if given) records the separate launch and its output becomes this launch's residual."""
        # gn: False, or (packer, norm name, groups, eps) of the GroupNorm in front of this conv: its statistics launch is recorded here
        gn_spec, gn = (gn, True) if isinstance(gn, tuple) else (None, bool(gn))
        ks = ks or pw["ks"]
        pad = (ks // 2 if not asym else 0) if pad is None else pad
        hin, win = x.h, x.w
        hu, wu = up_size if up_size else (hin << ups, win << ups)
        if asym:           # F.pad(0,1,0,1) + stride-2 p0 (VAE Downsample2D)
            ho, wo = (hu + 1 - ks) // stride + 1, (wu + 1 - ks) // stride + 1
        else:
            ho, wo = (hu + 2 * pad - ks) // stride + 1, (wu + 2 * pad - ks) // stride + 1
        n_out = pw["n"] // 2 if geglu else pw["n"]
        cp = cout_pad or ((n_out + 7) // 8 * 8)
        if out is None:
            out = self.new(x.n, ho, wo, cp, torch.float32 if out_f32 else None)
        c1 = x1.c if x1 else 0
        subpix = 1 if pw.get("subpix") else 0
        assert pw["w"].shape[1] == (4 if subpix else ks * ks) * (x.c + c1), (label, pw["w"].shape, ks, x.c, c1)
        x_in0, x_in1 = x, x1
        c0_eff, c1_eff = x.c, c1
        epc = 4 if self.dtype == torch.float32 else 8
        bk = 8 * epc
        # mirrors conv3x3_halo_eligible (csrc/conv3x3.hip): those convolutions apply GN+SiLU while staging the halo
        halo = (ks == 3 and stride == 1 and pad == 1 and not asym and not geglu and x.c % bk == 0 and c1 % bk == 0
                and wo >= 16 and ho >= 8)
        # ... but a halo launch that cannot fill the chip (few 8x16x128 tiles: the 16x16 / 32x32 UNet planes at
        # small batch) is a weight-streaming problem: the LDS-DMA igemm with split-K takes it (tile 20 forces it)
        force_tile, grp = 0, 0      # grp: which group of the I2I_W32_SPLITK mask took the op away from the halo conv (0 = none)
        if halo and not pw.get("subpix"):
            halo_tiles = x.n * -(-ho // 8) * -(-wo // 16) * -(-pw["n"] // 128)
            wide_ok = (self.dtype != torch.float32 and self.dma_small and pw["n"] % 160 == 0 and not ups and (gn or x1 is None)
                       and x.n * hin * win * (x.c + c1) * 2 < (1 << 32))          # (what gemm_w32_eligible will ask of the materialised operand)
            if halo_tiles < self.halo_min_tiles:
                halo, force_tile, grp = False, 20, 4
            elif (self.w32_splitk & 2) and wide_ok and x.n * ho * wo <= 2048 and pw["n"] >= 1024:
                # 16 x 16 planes at batch 8 (2048 rows x K = 5760 .. 23040): the wide GEMM with 4 K slices runs them at 740 - 920
                # TFLOP/s against ~500 of the halo conv's 160 tiles (profiles/r4h_*, r4i_per_op_*); GroupNorm + SiLU then come
                # materialised, like below
                halo, force_tile, grp = False, 20, 2
            elif (self.w32_splitk & 8) and wide_ok:
                # the UNet's 32 x 32 / 64 x 64 planes (320 / 640 output channels: 2.5 / 5 halo channel tiles, 650 - 750 TFLOP/s) on
                # 256 x 160 / 128 x 160 wide-GEMM tiles: 930 - 1115 TFLOP/s before the extra gn_apply pass (profiles/r4k_*)
                halo, force_tile, grp = False, 20, 8
        M, N, Kd = x.n * ho * wo, pw["n"], ks * ks * (x.c + c1)
        if grp in (2, 8):
            # These two groups leave a conv the halo kernel would take (with GroupNorm fused into its staging) for a wide-GEMM
            # route that needs the norm materialised: ask the C dispatcher about exactly that op BEFORE recording the extra
            # gn_apply pass, and keep the halo decision when it says no (wide_ok restates only part of gemm_w32_eligible).
            cfg_, sk_ = self._w32_splitk_cfg(M, N, Kd, self.w32_splitk_min_wgs, self.w32_splitk_longk)
            ok_ = bool(cfg_) and M >= self.w32_splitk_min_rows and x.c % epc == 0 and c1 % epc == 0
            if ok_:
                ct_ = x.c + c1
                probe = O.conv(x.t, pw["w"], out.t, nimg=x.n, hin=hin, win=win, ho=ho, wo=wo, ks=ks, stride=stride, pad=pad, ups=ups,
                               c0=ct_, c1=0, lda0=ct_, lda1=0, N=pw["n"], bias=pw["b"], alpha=alpha, res=res.t if res else None,
                               ldr=res.c if res else None, ldc=out.c, splitk=sk_ if sk_ > 1 else 0, ws=self.gn_probe_ws(sk_ * M * N) if sk_ > 1 else None,
                               tile=cfg_)
                ok_ = self.lib.igemm_route(probe[1], self.dt) == "gemm_w32_kernel"
            if not ok_:
                halo, force_tile, grp = True, 0, 0
        fused = gn and self.fuse_gn and (halo or not self.dma_small)
        if gn_spec is not None:
            self.gn_stats(gn_spec[0], gn_spec[1], x, gn_spec[2], gn_spec[3], x1)
        if gn and not fused:
            # materialise act(GN(x)) (both concat sources into ONE buffer): the LDS-DMA igemm / wide GEMM that take the UNet's
            # planes / 1x1 projections have no operand prologue, and the extra pass is over a few MB at most
            ct = x.c + c1
            y = self.new(x.n, x.h, x.w, ct)
            # (ABI v10: both sources of a concatenated input in ONE launch -- it was one per source: -0.06 +- 0.05 ms per batch-8 step,
            # -0.04 +- 0.015 ms at batch 1, profiles/r6k_ab_*_gn_apply_one.log)
            op = O.gn_apply(x.t, y.t, None, nimg=x.n, hw=x.hw, c=x.c, act=act, ldy=ct, ss_ld=ct, x1=x1.t if x1 else None, c1=c1)
            self._pending_gn.append((op[1], "apply"))
            self._add(op, label + ".gn_apply", nbytes=2 * x.n * x.hw * ct * self.esz)
            x_in0, x_in1, c0_eff, c1_eff = y, None, ct, 0
        if (ks == 1 and stride == 1 and not ups and not (halo or fused or geglu or out_f32) and force_tile in (0, 20)
                and self._small_tile(M, N, Kd)):
            tile_was, force_tile = force_tile, self._small_tile(M, N, Kd)
        splitk, ws = (0, None) if (halo or fused or geglu or force_tile == 26) else self._splitk(M, N, Kd)
        # (round 6, measured negative: the 256-row weight-streaming 3x3 convs of a batch-1 forward on ONE 256-row tile per column block
        # (tile 25) instead of four 64-row tiles that each stream the same weight slab: 50 -> 52.6 / 33 -> 36.7 us, +0.09 ms per step;
        # more than two slices for the 128-tile long-K wide-GEMM convs: +0.18 ... +0.26 ms: profiles/r6m_ab_bs1_*.log)
        mk = lambda tile_, splitk_, ws_: O.conv(
            x_in0.t, pw["w"], out.t, nimg=x.n, hin=hin, win=win, ho=ho, wo=wo, ks=ks, stride=stride, pad=pad, ups=ups,
            x1=x_in1.t if x_in1 else None, c0=c0_eff, c1=c1_eff, lda0=c0_eff, lda1=c1_eff, N=pw["n"],
            gn_ss=None, act=act if fused else 0, bias=pw["b"], alpha=alpha,
            res=res.t if res else None, ldr=res.c if res else None, ldc=out.c, geglu=geglu, out_f32=out_f32,
            splitk=splitk_, ws=ws_, subpix=subpix, tile=tile_, up_size=up_size)
        op = None
        want = ((self.w32_splitk & 1) and M <= 16384) if stride == 2 else (self.w32_splitk & (grp or 4))      # (VAE downsamplers: tile 0 finds the wide GEMM by itself)
        if (want and ks == 3 and stride in (1, 2) and not (halo or fused or geglu or ups or out_f32) and self.dtype != torch.float32
                and x_in1 is None and M >= self.w32_splitk_min_rows):
            # UNet small-plane / stride-2 3x3 convolutions on the wide GEMM (csrc/gemm_w32.hip: im2col gather + K slices): the tile
            # and slice count that put ~256 workgroups on the chip with >= 8 stages each (sweep: profiles/r4h_bench_ops_splitk_*)
            cfg, sk = self._w32_splitk_cfg(M, N, Kd, self.w32_splitk_min_wgs, self.w32_splitk_longk)
            if cfg:
                ws2 = self.pool.get(sk * M * N, torch.float32) if sk > 1 else None
                cand = mk(cfg, sk if sk > 1 else 0, ws2)
                if self.lib.igemm_route(cand[1], self.dt) == "gemm_w32_kernel":
                    if ws is not None:
                        self.pool.put(ws)
                    op, ws = cand, ws2
                else:
                    if grp in (2, 8):       # (the probe above asked the same question: unreachable unless the two drift apart)
                        raise K.I2IError("plan: %s was taken from the halo conv for a wide-GEMM route the dispatcher refuses (%s)"
                                         % (label, self.lib.igemm_route(cand[1], self.dt)))
                    if ws2 is not None:
                        self.pool.put(ws2)
        if op is None:
            op = mk(force_tile, splitk, ws)
            if force_tile == 26 and self.lib.igemm_route(op[1], self.dt) != "igemm_dma_kernel":     # (not taken by the engine: the old decision)
                force_tile = tile_was
                splitk, ws = self._splitk(M, N, Kd)
                op = mk(force_tile, splitk, ws)
        if ws is not None:
            self.pool.put(ws)      # the program runs in order on one stream: later ops may reuse the slab
        out.producer = op[1]
        if fused:
            self._pending_gn.append((op[1], "igemm"))
            op[1].gn_ss = 1  # placeholder, patched in _finish_gn_scratch
        fl = 2 * x.n * ho * wo * pw["n"] * ks * ks * (x.c + c1)
        fl_k2 = 0
        kname = self.lib.igemm_route(op[1], self.dt)     # which kernel the C dispatcher picks
        out.k2_fused = False
        if k2 is not None:
            # k2 = dict(x=Act at output resolution, w=packed 1x1 weights, label=..., [bias=fp32 bias to use when fused],
            #           [fallback=callable -> residual Act when the launch cannot take the contraction])
            k2x, k2w = k2["x"], k2["w"]
            p_ = op[1]
            ok = (kname in ("conv3x3_w32_kernel<SUBPIX>", "conv3x3_w32_kernel") and res is None and k2x.c % 64 == 0 and
                  (k2x.n, k2x.h, k2x.w) == (x.n, ho, wo) and k2w["n"] == pw["n"] and k2w["w"].shape[1] == k2x.c and
                  (k2w["b"] is None or k2.get("bias") is not None))
            if ok:
                p_.k2_a, p_.k2_b, p_.k2_c, p_.k2_lda, p_.k2_ldb = k2x.t.data_ptr(), k2w["w"].data_ptr(), k2x.c, k2x.c, k2w["w"].shape[1]
                keep = (k2x.t, k2w["w"])
                if k2.get("bias") is not None:          # the folded conv's bias rides in the launch's bias vector (Packer.bias_sum)
                    p_.bias, p_.bias_mode = k2["bias"].data_ptr(), 1
                    keep += (k2["bias"],)
                p_._keep = tuple(p_._keep) + keep
                fl_k2 = 2 * x.n * ho * wo * pw["n"] * k2x.c
                label = label + " + " + k2["label"]
                out.k2_fused = True
            elif k2.get("fallback") is not None:        # the separate launch after all: its output is this launch's residual
                r_ = k2["fallback"]()
                p_.res, p_.ldr = r_.t.data_ptr(), r_.c
                p_._keep = tuple(p_._keep) + (r_.t,)
                out.k2_residual = r_
        if fused and kname == "igemm_kernel":
            kname = "igemm_kernel (register-staged, GN prologue)"
        # the sub-pixel form executes 4/9 of the upsample + 3x3 MACs; a folded 1x1 contraction (k2) is executed in full
        fl_exec = (fl * 4 // 9 if "<SUBPIX>" in kname else fl) + fl_k2
        fl += fl_k2
        if kname.startswith("conv3x3_"):
            self.halo_flops_real = getattr(self, "halo_flops_real", 0) + fl_exec
        # the narrow-end convs (conv_narrow.hip) are HBM streams: priced in bytes as well (one read of the input, one write of the output)
        nb = (x.n * hin * win * (x.c + c1) + x.n * ho * wo * out.c) * self.esz if kname.startswith("conv_narrow") else 0
        self._add(op, label, fl, kernel=kname, flops_exec=fl_exec, nbytes=nb)
        self.taps[label] = out
        if gn and not fused:
            self.free(x_in0)
        self.flops += fl
        return out

    def linear(self, pw, x2d, rows, cin, *, out=None, res=None, geglu=0, label="", out_cols=None):
        """Token-major linear: x2d is a flat tensor viewed as [rows][cin]."""
        n_out = pw["n"] // 2 if geglu else pw["n"]
        out_cols = out_cols or n_out
        if out is None:
            out = self.pool.get(rows * out_cols, self.dtype)
        mk = lambda tile_, splitk_, ws_: O.conv(x2d, pw["w"], out, nimg=1, hin=1, win=rows, ho=1, wo=rows, ks=1, c0=cin, lda0=cin, N=pw["n"], bias=pw["b"],
                                                res=res, ldr=n_out if res is not None else None, ldc=out_cols, geglu=geglu, splitk=splitk_, ws=ws_, tile=tile_)
        # (round 6, measured negative: the K-sliced small linears -- a split-K launch + its reduce launch -- as ONE un-sliced wide-GEMM launch
        # with a handful of tiles: +0.06 ... +0.32 ms at batch 1 for row limits 64 ... 4096, neutral at batch 8; profiles/r6d_ab_bs1_unsplit.log.
        # What does pay is the un-sliced launch on tiles small enough to fill the chip: _small_tile)
        op = None
        st = 0 if geglu else self._small_tile(rows, pw["n"], cin)
        if st:
            op = mk(st, 0, None)
            if self.lib.igemm_route(op[1], self.dt) != "igemm_dma_kernel":
                op = None
        if op is None:
            splitk, ws = (0, None) if geglu else self._splitk(rows, pw["n"], cin)
            op = mk(0, splitk, ws)
            if ws is not None:
                self.pool.put(ws)
        self._add(op, label, 2 * rows * pw["n"] * cin, kernel=self.lib.igemm_route(op[1], self.dt))
        self.flops += 2 * rows * pw["n"] * cin
        return out

    def ln_gemm(self, pw, x2d, rows, cin, *, out=None, out_cols=None, geglu=0, label="", n_trans=0, out2=None, ldc2=0):
        """LayerNorm + linear in one wide-GEMM launch (csrc/gemm_w32.hip LNF): ``x2d`` holds the UN-normalised rows, ``pw`` comes from
        Packer.ln_linear (LayerNorm weight folded into W, colsum / bias' beside it).  With ``n_trans`` the output columns from there on
        are written transposed to ``out2`` (row pitch ``ldc2``: the self-attention V^T).  Returns the output tensor, or None -- with
        nothing recorded -- when the dispatcher would not run this op on the wide GEMM (the caller then keeps the LayerNorm launch)."""
        n_out = pw["n"] // 2 if geglu else pw["n"]
        cols = out_cols or (n_trans if n_trans else n_out)
        own = out is None
        if own:
            out = self.pool.get(rows * cols, self.dtype)
        op = O.conv(x2d, pw["w"], out, nimg=1, hin=1, win=rows, ho=1, wo=rows, ks=1, c0=cin, lda0=cin, N=pw["n"], bias=pw["b"], ldc=cols,
                    geglu=geglu, tile=50)
        q = op[1]
        q.ln_cs, q.ln_eps = pw["cs"].data_ptr(), 1e-5
        if n_trans:
            q.n_trans, q.c2, q.ldc2 = n_trans, out2.data_ptr(), ldc2
        q._keep = tuple(q._keep) + (pw["cs"],) + ((out2,) if out2 is not None else ())
        if self.lib.igemm_route(q, self.dt) != "gemm_w32_kernel":
            if own:
                self.pool.put(out)
            return None
        self._add(op, label, 2 * rows * pw["n"] * cin, kernel="gemm_w32_kernel")
        self.flops += 2 * rows * pw["n"] * cin
        return out

    # ------------------------------------------------------------------ blocks
    def resnet(self, pk, prefix, x: Act, cout, groups, eps, x1: Optional[Act] = None, arch=None) -> Act:
        split = x.c if x1 else None
        n1, n2 = (pk, prefix + ".norm1", groups, eps), (pk, prefix + ".norm2", groups, eps)
        h = self.conv(pk.resnet_conv1(prefix, arch, split), x, x1=x1, gn=n1, act=1, label=prefix + ".conv1")
        if pk.has(prefix + ".conv_shortcut"):
            # output = conv_shortcut(input) + conv2(...) (diffusers ResnetBlock2D): the 1x1 shortcut rides as a second contraction
            # of the conv2 launch when the wide-tile conv takes it (no launch of its own, no write + re-read of its output);
            # otherwise it runs first and conv2 adds its output as the residual.
            sc_conv = lambda: self.conv(pk.conv(prefix + ".conv_shortcut", split=split), x, x1=x1, label=prefix + ".conv_shortcut")
            if self.fuse_shortcut and x1 is None and x.c % 64 == 0:
                k2 = dict(x=x, w=pk.conv(prefix + ".conv_shortcut"), label=prefix + ".conv_shortcut",
                          bias=pk.bias_sum(prefix + ".conv2", prefix + ".conv_shortcut"), fallback=sc_conv)
                out = self.conv(pk.conv(prefix + ".conv2"), h, gn=n2, act=1, label=prefix + ".conv2", k2=k2)
                if getattr(out, "k2_residual", None) is not None:
                    self.free(out.k2_residual)
            else:
                sc = sc_conv()
                out = self.conv(pk.conv(prefix + ".conv2"), h, gn=n2, act=1, res=sc, label=prefix + ".conv2")
                self.free(sc)
        else:
            assert x1 is None
            out = self.conv(pk.conv(prefix + ".conv2"), h, gn=n2, act=1, res=x, label=prefix + ".conv2")
        self.free(h)
        return out

    def vae_attention(self, pk, prefix, x: Act, groups, eps) -> Act:
        """VAE mid attention (1 head of width C): GN -> q,k | v^T -> scores -> softmax -> p.v -> out + x.
        Unfused (scores materialised): head dim 512 does not fit the d=64 flash kernel."""
        B, T, C = x.n, x.hw, x.c
        self.gn_stats(pk, prefix + ".group_norm", x, groups, eps)
        xn = self.new(x.n, x.h, x.w, C)
        op = O.gn_apply(x.t, xn.t, None, nimg=B, hw=T, c=C, act=0)
        self._pending_gn.append((op[1], "apply"))
        self._add(op, prefix + ".gn_apply", nbytes=2 * B * T * C * self.esz)
        qk = self.linear(pk.stacked_linear([prefix + ".to_q", prefix + ".to_k"]), xn.t, B * T, C, label=prefix + ".to_qk")
        wv = pk.conv(prefix + ".to_v")
        Tp = (T + 7) // 8 * 8                      # row pitch of V^T / scores / probabilities (T itself when the plane is /8-aligned)
        vt = self.pool.get(B * C * Tp, self.dtype)
        self._add(O.bgemm(wv["w"], xn.t, vt, M=C, N=T, Kdim=C, lda=C, ldb=C, ldc=Tp, batch=B, heads=1, a_bs=(0, 0),
                          b_bs=(T * C, 0), c_bs=(C * Tp, 0), bias=wv["b"], bias_mode=2), prefix + ".to_v^T", 2 * B * T * C * C)
        self.flops += 2 * B * T * C * C
        self.free(xn)
        if C == 512 and self.dtype != torch.float32 and T >= 8 and self.fuse_vae_attention:
            # one head of width 512: the wide-head flash kernel (attention.hip), scores never leave the CU
            o = self.pool.get(B * T * C, self.dtype)
            # too few query tiles to fill the chip (batch 1: 32 workgroups of 128 queries, all on one XCD): split the KEYS over
            # ksplit workgroups per query tile and merge (csrc/attention.hip SPLIT); every split keeps >= 8 key tiles of 32
            nqt, ktiles = B * -(-T // 128), -(-T // 32)
            ksplit = 1
            while self.att_ksplit and nqt * ksplit * 2 <= 256 and ktiles % (ksplit * 2) == 0 and ktiles // (ksplit * 2) >= 8 and ksplit < 8:
                ksplit *= 2
            ws = self.pool.get(B * ksplit * T * (C + 2), torch.float32) if ksplit > 1 else None
            self._add(O.attention(qk, qk[C:], vt, o, batch=B, heads=1, d=C, tq=T, tk=T, ldq=2 * C, ldk=2 * C, ldvt=Tp, ldo=C,
                                  q_bs=T * 2 * C, k_bs=T * 2 * C, vt_bs=C * Tp, o_bs=T * C, scale=1.0 / math.sqrt(C),
                                  ksplit=ksplit if ksplit > 1 else 0, ws=ws), prefix + ".sdpa",
                      4 * B * T * T * C, kernel="attention_wide_kernel")
            if ws is not None:
                self.pool.put(ws)
            self.flops += 4 * B * T * T * C
            self.pool.put(qk)
            self.pool.put(vt)
            out = self.new(x.n, x.h, x.w, C)
            self.linear(pk.conv(prefix + ".to_out.0"), o, B * T, C, out=out.t, res=x.t, label=prefix + ".to_out")
            self.pool.put(o)
            return out
        s = self.pool.get(B * T * Tp, torch.float32)
        self._add(O.bgemm(qk, qk[C:], s, M=T, N=T, Kdim=C, lda=2 * C, ldb=2 * C, ldc=Tp, batch=B, heads=1,
                          a_bs=(T * 2 * C, 0), b_bs=(T * 2 * C, 0), c_bs=(T * Tp, 0), out_f32=1), prefix + ".qk^T")
        p = self.pool.get(B * T * Tp, self.dtype)
        self._add(O.softmax(s, p, rows=B * T, cols=T, lds=Tp, ldp=Tp, scale=1.0 / math.sqrt(C)), prefix + ".softmax")
        self.pool.put(s)
        self.pool.put(qk)
        o = self.pool.get(B * T * C, self.dtype)
        # pad columns of P are exact zeros; the matching V^T columns only have to be finite (zeroed once below)
        self._add(O.bgemm(p, vt, o, M=T, N=C, Kdim=Tp, lda=Tp, ldb=Tp, ldc=C, batch=B, heads=1, a_bs=(T * Tp, 0), b_bs=(C * Tp, 0),
                          c_bs=(T * C, 0)), prefix + ".pv")
        if Tp != T:
            self._zero_init.append(vt)
        self.flops += 4 * B * T * T * C
        self.pool.put(p)
        self.pool.put(vt)
        out = self.new(x.n, x.h, x.w, C)
        self.linear(pk.conv(prefix + ".to_out.0"), o, B * T, C, out=out.t, res=x.t, label=prefix + ".to_out")
        self.pool.put(o)
        return out

    def _cross_kv(self, pk, ctx, tk):
        """Every cross-attention K and V^T projection of the UNet in TWO launches (diffusers Attention.to_k / to_v of the 16
        attn2 modules, reference call src/pix2pix_turbo.py:199): they all read the same text states, so their weights are
        stacked along the output rows (packer.stacked_linear: LoRA re-merge per row block as before) and module i reads its
        column slice of K_all [ctx_batch*tk][sumC] / its row slice of V^T_all [vb][sumC][ldvt].  32 launch-latency-bound
        77-row GEMMs (16 to 25 us each) become two.  I2I_CROSS_KV_MERGED=0 keeps one pair per module (A/B)."""
        if self._ckv is None:
            names = sorted({k[:k.index(".to_k.")] for k in pk.sd if ".attn2.to_k." in k})
            widths = [pk.base(n + ".to_k")[0].shape[0] for n in names]
            offs, tot = {}, 0
            for n, c in zip(names, widths):
                offs[n], tot = tot, tot + c
            cd = self.ua.cross_attention_dim
            epc = 4 if self.dtype == torch.float32 else 8
            ldvt = (tk + epc - 1) // epc * epc
            vb = self.ctx_batch
            kall = self.linear(pk.stacked_linear([n + ".to_k" for n in names]), ctx, self.ctx_batch * tk, cd, label="attn2.to_k (all modules)")
            wv = pk.stacked_linear([n + ".to_v" for n in names])
            assert wv["b"] is None
            vt = self.pool.get(vb * tot * ldvt, self.dtype)
            self._add(O.bgemm(wv["w"], ctx, vt, M=tot, N=tk, Kdim=cd, lda=cd, ldb=cd, ldc=ldvt, batch=vb, heads=1,
                              a_bs=(0, 0), b_bs=((tk * cd if vb > 1 else 0), 0), c_bs=(tot * ldvt, 0)), "attn2.to_v^T (all modules)", 2 * vb * tk * tot * cd)
            self.flops += 2 * vb * tk * tot * cd
            self._zero_init.append(vt)       # (pad columns of V^T only have to be finite)
            self._ckv = dict(k=kall, vt=vt, off=offs, tot=tot, ldvt=ldvt)      # never returned to the pool: read until the last block
        return self._ckv

    def attention_block(self, pk, p, xn, x_res, B, T, C, heads, ctx=None, tk=None, ln=None):
        """attn1 (self) or attn2 (cross, ``ctx`` = text states).  Returns x_res + to_out(attn).
        ``ln`` = name of the LayerNorm in front of the block: ``xn`` is then the UN-normalised tensor and the norm rides in the
        projection launch (self-attention: to_q | to_k | to_v^T in one launch; cross-attention: to_q); returns None with nothing
        recorded when that launch is not available (the caller runs the LayerNorm and calls again without ``ln``)."""
        d = C // heads
        scale = 1.0 / math.sqrt(d)
        # the LDS-DMA flash kernel works in log2 units: scale * log2(e) is folded into to_q (W, bias and lora_B in fp32 before the one
        # rounding of the packed weights), so q leaves its projection ready for v_exp_f32 and the kernel is told scale = ln 2
        # (softmax(ln2 * (c q).k) with c = scale * log2 e is the same function); csrc/attention.hip attention_dma_kernel
        qs = None
        if self.flash and d == 64 and self.dtype != torch.float32 and self.att_q_log2:
            qs, scale = scale * math.log2(math.e), math.log(2.0)
        vt_fused = None
        if ctx is None:
            tk = T
            if ln is not None:
                if not (self.flash and d == 64 and tk % 8 == 0 and pk.base(p + ".to_v")[1] is None):
                    return None
                nt = B * T
                qk, vt_fused = self.pool.get(B * T * 2 * C, self.dtype), self.pool.get(C * nt, self.dtype)
                if self.ln_gemm(pk.ln_linear([p + ".to_q", p + ".to_k", p + ".to_v"], ln, scale0=qs), xn, B * T, C, out=qk, out_cols=2 * C,
                                label=p + ".norm+to_qkv^T", n_trans=2 * C, out2=vt_fused, ldc2=nt) is None:
                    self.pool.put(qk)
                    self.pool.put(vt_fused)
                    return None
            else:
                qk = self.linear(pk.stacked_linear([p + ".to_q", p + ".to_k"], scale0=qs), xn, B * T, C, label=p + ".to_qk")
            q, k, ldq, ldk, q_bs, k_bs = qk, qk[C:], 2 * C, 2 * C, T * 2 * C, T * 2 * C
            kv_src, kv_cin, kv_bs = xn, C, T * C
        else:
            cd = self.ua.cross_attention_dim
            if ln is not None:
                q = self.ln_gemm(pk.ln_linear([p + ".to_q"], ln, scale0=qs), xn, B * T, C, label=p + ".norm+to_q")
                if q is None:
                    return None
            else:
                q = self.linear(pk.conv(p + ".to_q", out_scale=qs), xn, B * T, C, label=p + ".to_q")
            ldq, q_bs = C, T * C
            qk = None
            merged = self.cross_kv_merged and p.endswith(".attn2")
            if merged:
                ck = self._cross_kv(pk, ctx, tk)
                k, ldk = ck["k"][ck["off"][p]:], ck["tot"]
                k_bs = tk * ck["tot"] if self.ctx_batch > 1 else 0
            else:
                k = self.linear(pk.conv(p + ".to_k"), ctx, self.ctx_batch * tk, cd, label=p + ".to_k")
                ldk = C
                k_bs = tk * C if self.ctx_batch > 1 else 0
            kv_src, kv_cin, kv_bs = ctx, cd, (tk * cd if self.ctx_batch > 1 else 0)
        epc = 4 if self.dtype == torch.float32 else 8
        ldvt = (tk + epc - 1) // epc * epc
        vb = B if (ctx is None or self.ctx_batch > 1) else 1
        if vt_fused is not None:
            vt, ldvt, vt_stride = vt_fused, B * T, tk      # V^T_all [C][B*T] written by the projection launch: image b = column block b*T
        elif ctx is not None and merged:
            vt, vt_stride = ck["vt"][ck["off"][p] * ldvt:], ck["tot"] * ldvt        # rows off .. off + C of every image's V^T_all
        else:
            wv = pk.conv(p + ".to_v")
            one = None
            if ctx is None and self.vt_one_launch and self.flash and d == 64 and wv["b"] is None and tk % 8 == 0:
                # V^T of ALL images as one plain GEMM, V^T_all [C][B*T] = W_v [C][K] . X [B*T][K]^T (the weight matrix is the row
                # operand): image b's V^T is the column block b*T .. of every row, which the attention kernel reads through its
                # strides (ldvt = B*T, batch stride T).  One chip-filling launch on the wide GEMM instead of B weight-row-bound
                # z-slices of the LDS-DMA igemm.
                nt = vb * tk
                vt = self.pool.get(C * nt, self.dtype)
                one = O.conv(wv["w"], kv_src, vt, nimg=1, hin=1, win=C, ho=1, wo=C, ks=1, c0=kv_cin, lda0=kv_cin, N=nt, ldb=kv_cin, ldc=nt,
                             tile=53 if C % 256 == 0 else 54)
                if self.lib.igemm_route(one[1], self.dt) != "gemm_w32_kernel":
                    self.pool.put(vt)
                    one = None
            if one is not None:
                ldvt, vt_stride = nt, tk
                self._add(one, p + ".to_v^T", 2 * vb * tk * C * kv_cin, kernel="gemm_w32_kernel")
            else:
                vt, vt_stride = self.pool.get(vb * C * ldvt, self.dtype), C * ldvt
                self._add(O.bgemm(wv["w"], kv_src, vt, M=C, N=tk, Kdim=kv_cin, lda=kv_cin, ldb=kv_cin, ldc=ldvt, batch=vb, heads=1,
                                  a_bs=(0, 0), b_bs=(kv_bs, 0), c_bs=(C * ldvt, 0)), p + ".to_v^T", 2 * vb * tk * C * kv_cin)
            self.flops += 2 * vb * tk * C * kv_cin
        o = self.pool.get(B * T * C, self.dtype)
        if self.flash and d == 64:
            # few workgroups x many SERIAL key tiles (batch 1, T = 4096: 320 workgroups of 64 tiles at ~1 us each): keys divided among
            # `ks` workgroups per query tile + the merge launch (csrc/attention.hip SPLIT; I2I_ATT_KSPLIT64 = splits, 0 = off)
            ks, ws = 0, None
            if (self.att_ksplit64 > 1 and self.dtype != torch.float32 and tk >= self.att_ksplit64_min_tk and B * heads * -(-T // 64) <= 512
                    and C % 8 == 0 and (T * C) % 8 == 0):
                ks = self.att_ksplit64
                ws = self.pool.get(B * heads * ks * T * (d + 2), torch.float32)
            self._add(O.attention(q, k, vt, o, batch=B, heads=heads, d=d, tq=T, tk=tk, ldq=ldq, ldk=ldk, ldvt=ldvt, ldo=C,
                                  q_bs=q_bs, k_bs=k_bs, vt_bs=(vt_stride if vb > 1 else 0), o_bs=T * C, scale=scale, ksplit=ks, ws=ws), p + ".sdpa",
                      4 * B * heads * T * tk * d, kernel="attention_dma_kernel" if self.dtype != torch.float32 else "attention_kernel")
            if ws is not None:
                self.pool.put(ws)
        else:
            ldp = ldvt
            s = self.pool.get(B * heads * T * tk, torch.float32)
            self._add(O.bgemm(q, k, s, M=T, N=tk, Kdim=d, lda=ldq, ldb=ldk, ldc=tk, batch=B, heads=heads, a_bs=(q_bs, d),
                              b_bs=(k_bs, d), c_bs=(heads * T * tk, T * tk), out_f32=1), p + ".qk^T")
            pm = self.pool.get(B * heads * T * ldp, self.dtype)
            self._add(O.softmax(s, pm, rows=B * heads * T, cols=tk, lds=tk, ldp=ldp, scale=scale), p + ".softmax")
            self.pool.put(s)
            # V^T pad columns are multiplied by exact zeros of P; keep them finite
            self._add(O.bgemm(pm, vt, o, M=T, N=d, Kdim=ldp, lda=ldp, ldb=ldvt, ldc=C, batch=B, heads=heads,
                              a_bs=(heads * T * ldp, T * ldp), b_bs=((vt_stride if vb > 1 else 0), d * ldvt), c_bs=(T * C, d)), p + ".pv")
            self.pool.put(pm)
            if not (ctx is not None and merged):
                self._zero_init.append(vt)
        self.flops += 4 * B * heads * T * tk * d
        if qk is not None:
            self.pool.put(qk)
        else:
            self.pool.put(q)
        if ctx is None or not merged:
            if qk is None:
                self.pool.put(k)
            self.pool.put(vt)
        out = self.linear(pk.conv(p + ".to_out.0"), o, B * T, C, res=x_res, label=p + ".to_out")
        self.pool.put(o)
        return out

    def transformer(self, pk, p, x: Act, heads, groups) -> Act:
        """Transformer2DModel (1 BasicTransformerBlock, linear projections): GN(eps 1e-6) -> proj_in -> [attn1, attn2, GEGLU FF] -> proj_out + x."""
        B, T, C = x.n, x.hw, x.c
        rows = B * T
        h = self.conv(pk.conv(p + ".proj_in"), x, ks=1, gn=(pk, p + ".norm", groups, 1e-6), act=0, label=p + ".proj_in").t   # tokens [rows][C]
        t = p + ".transformer_blocks.0"

        def ln(name, src):
            g, b = pk.norm(name)
            y = self.pool.get(rows * C, self.dtype)
            self._add(O.layernorm(src, y, g, b, rows=rows, c=C, eps=1e-5), name, nbytes=2 * rows * C * self.esz)
            return y

        # LayerNorm folded into the projection behind it where the wide GEMM takes the launch (self.ln_fold), else its own launch
        h2 = self.attention_block(pk, t + ".attn1", h, h, B, T, C, heads, ln=t + ".norm1") if self.ln_fold else None
        if h2 is None:
            y = ln(t + ".norm1", h)
            h2 = self.attention_block(pk, t + ".attn1", y, h, B, T, C, heads)
            self.pool.put(y)
        self.pool.put(h)
        h3 = self.attention_block(pk, t + ".attn2", h2, h2, B, T, C, heads, ctx=self.ctx, tk=77, ln=t + ".norm2") if self.ln_fold else None
        if h3 is None:
            y = ln(t + ".norm2", h2)
            h3 = self.attention_block(pk, t + ".attn2", y, h2, B, T, C, heads, ctx=self.ctx, tk=77)
            self.pool.put(y)
        self.pool.put(h2)
        # (norm3 -> GEGLU only where the launch is latency bound: at batch 8 the folded epilogue costs the GEGLU tiles their second
        # workgroup per CU -- 197 + 80 registers -- and more than the LayerNorm launch it saves: profiles/r6b_*)
        ff = (self.ln_gemm(pk.ln_linear([t + ".ff.net.0.proj"], t + ".norm3", geglu=True), h3, rows, C, geglu=1, label=t + ".norm3+ff.geglu")
              if (self.ln_fold and rows <= self.ln_fold_geglu_max_rows) else None)
        if ff is None:
            y = ln(t + ".norm3", h3)
            ff = self.linear(pk.geglu_linear(t + ".ff.net.0.proj"), y, rows, C, geglu=1, label=t + ".ff.geglu")
            self.pool.put(y)
        h4 = self.linear(pk.conv(t + ".ff.net.2"), ff, rows, 4 * C, res=h3, label=t + ".ff.net.2")
        self.pool.put(ff)
        self.pool.put(h3)
        out = self.new(x.n, x.h, x.w, C)
        self.linear(pk.conv(p + ".proj_out"), h4, rows, C, out=out.t, res=x.t, label=p + ".proj_out")
        self.pool.put(h4)
        return out

    # ------------------------------------------------------------------ the three networks
    def _vae_encoder(self, x: Act):
        pk, a = self.pv, self.va
        g, eps = a.norm_num_groups, a.eps
        boc = a.block_out_channels
        h = self.conv(pk.conv("encoder.conv_in"), x, label="encoder.conv_in")
        skips = []
        for i, c in enumerate(boc):
            skips.append(h)                                   # src/model.py:18-20 (kept alive for the decoder)
            cur = h
            for j in range(a.layers_per_block):
                nxt = self.resnet(pk, f"encoder.down_blocks.{i}.resnets.{j}", cur, c, g, eps)
                if cur is not h:
                    self.free(cur)
                cur = nxt
            if i < len(boc) - 1:
                nxt = self.conv(pk.conv(f"encoder.down_blocks.{i}.downsamplers.0.conv"), cur, stride=2, asym=True,
                                label=f"encoder.down_blocks.{i}.downsamplers.0.conv")
                self.free(cur)
                cur = nxt
            h = cur
        m = self.resnet(pk, "encoder.mid_block.resnets.0", h, boc[-1], g, eps)
        self.free(h)                                          # down3 output is not a skip
        m2 = self.vae_attention(pk, "encoder.mid_block.attentions.0", m, g, eps)
        self.free(m)
        m3 = self.resnet(pk, "encoder.mid_block.resnets.1", m2, boc[-1], g, eps)
        self.free(m2)
        moments = self.conv(pk.encoder_out(), m3, gn=(pk, "encoder.conv_norm_out", g, eps), act=1, out_f32=1, label="encoder.conv_out+quant_conv")
        self.free(m3)
        return moments, skips

    def _unet(self, u: Act) -> Act:
        pk, a = self.pu, self.ua
        g, eps = a.norm_num_groups, a.norm_eps
        boc, heads = a.block_out_channels, a.num_heads
        nb = len(boc)
        conv_in = pk.twin_conv_in() if self.w8s.is_twin_conv else pk.conv("conv_in")
        h = self.conv(conv_in, u, label="conv_in")
        res = [h]
        for i, c in enumerate(boc):
            for j in range(a.layers_per_block):
                h2 = self.resnet(pk, f"down_blocks.{i}.resnets.{j}", h, c, g, eps, arch=a)
                if i < nb - 1:
                    h3 = self.transformer(pk, f"down_blocks.{i}.attentions.{j}", h2, heads[i], g)
                    self.free(h2)
                    h2 = h3
                h = h2
                res.append(h)
            if i < nb - 1:
                h = self.conv(pk.conv(f"down_blocks.{i}.downsamplers.0.conv"), h, stride=2, label=f"down_blocks.{i}.downsamplers.0.conv")
                res.append(h)
        m = self.resnet(pk, "mid_block.resnets.0", h, boc[-1], g, eps, arch=a)
        m2 = self.transformer(pk, "mid_block.attentions.0", m, heads[-1], g)
        self.free(m)
        h = self.resnet(pk, "mid_block.resnets.1", m2, boc[-1], g, eps, arch=a)
        self.free(m2)
        rheads = list(reversed(heads))
        rboc = list(reversed(boc))
        for i, c in enumerate(rboc):
            for j in range(a.layers_per_block + 1):
                skip = res.pop()
                h2 = self.resnet(pk, f"up_blocks.{i}.resnets.{j}", h, c, g, eps, x1=skip, arch=a)   # cat([hidden, skip])
                self.free(h)
                self.free(skip)
                if i > 0:
                    h3 = self.transformer(pk, f"up_blocks.{i}.attentions.{j}", h2, rheads[i], g)
                    self.free(h2)
                    h2 = h3
                h = h2
            if i < nb - 1:
                nxt = res[-1]      # the next level's skip fixes the output size (UNet2DConditionModel.forward: upsample_size)
                h2 = self.upsample_conv(pk, f"up_blocks.{i}.upsamplers.0.conv", h, f"up_blocks.{i}.upsamplers.0.conv", size=(nxt.h, nxt.w))
                self.free(h)
                h = h2
        assert not res
        e = self.conv(pk.conv("conv_out"), h, gn=(pk, "conv_norm_out", g, eps), act=1, out_f32=1, label="conv_out")     # eps-prediction kept fp32
        self.free(h)
        return e

    def _vae_decoder(self, z: Act, skips: List[Act]) -> Act:
        pk, a = self.pv, self.va
        g, eps = a.norm_num_groups, a.eps
        rboc = list(reversed(a.block_out_channels))
        h = self.conv(pk.conv("decoder.conv_in"), z, label="decoder.conv_in")
        m = self.resnet(pk, "decoder.mid_block.resnets.0", h, rboc[0], g, eps)
        self.free(h)
        m2 = self.vae_attention(pk, "decoder.mid_block.attentions.0", m, g, eps)
        self.free(m)
        h = self.resnet(pk, "decoder.mid_block.resnets.1", m2, rboc[0], g, eps)
        self.free(m2)
        skip_done = False      # the skip conv of this block already rode in the upsampler that produced h
        for i, c in enumerate(rboc):
            sk = skips[::-1][i]
            # sample = sample + skip_conv_i(skip * gamma)   (src/model.py:41-43), in place
            # (gamma is folded into the skip-conv weights by the device-side merge: Packer.conv(gamma=True))
            if not skip_done:
                self.conv(pk.conv(f"decoder.skip_conv_{i + 1}", gamma=True), sk, ks=1, res=h, out=h, label=f"decoder.skip_conv_{i + 1}")
            self.free(sk)
            for j in range(a.layers_per_block + 1):
                h2 = self.resnet(pk, f"decoder.up_blocks.{i}.resnets.{j}", h, c, g, eps)
                self.free(h)
                h = h2
            if i < len(rboc) - 1:
                # the NEXT block's `sample + skip_conv(skip * gamma)` as a second contraction of this upsampler (when the wide-tile
                # sub-pixel kernel takes the launch): the 1x1 skip conv was an HBM-bound read-modify-write of the whole stream
                k2 = dict(x=skips[::-1][i + 1], w=pk.conv(f"decoder.skip_conv_{i + 2}", gamma=True), label=f"decoder.skip_conv_{i + 2}") if self.fuse_skip else None
                h2 = self.upsample_conv(pk, f"decoder.up_blocks.{i}.upsamplers.0.conv", h, f"decoder.up_blocks.{i}.upsamplers.0.conv", k2=k2)
                skip_done = bool(getattr(h2, "k2_fused", False))
                self.free(h)
                h = h2
        y = self.conv(pk.conv("decoder.conv_out"), h, gn=(pk, "decoder.conv_norm_out", g, eps), act=1, label="decoder.conv_out")
        self.free(h)
        return y

    def _build(self):
        B, H, W = self.B, self.H, self.W
        lat = self.va.latent_channels
        h8, w8 = H // 8, W // 8
        self._zero_init = []
        x = self.new(B, H, W, 8)
        mul, add, thr = (tuple(self.u8_io) + (0,))[:3] if self.u8_io else (1.0, 0.0, 0)      # (mul, add[, binarize_below])
        self._add(O.nchw_to_nhwc(self.x_in, x.t, n=B, c=3, h=H, w=W, cpad=8, mul=mul, add=add, binarize_below=thr), "input.to_nhwc",
                  nbytes=B * H * W * (3 * self.x_in.element_size() + 8 * self.esz))
        moments, skips = self._vae_encoder(x)
        # x (= conv_in input) is not a skip; skips[0] is conv_in's output
        self.free(x)
        self.u32 = torch.zeros(B * h8 * w8 * lat, dtype=torch.float32, device=self.device)
        sf = self.va.scaling_factor
        with self._as(self.unet_dtype):          # fp32 moments in, the UNet's element type out; eps-prediction comes back in fp32
            u = self.new(B, h8, w8, 8)
            self._add(O.posterior(moments.t, self.eps, u.t, n=B, hw=h8 * w8, lat=lat, ldm=moments.c, ldu=8, sf=sf, r=self.r,
                                  r_dev=self.pv.rg if self.stochastic else None,
                                  noise=self.noise, noise_n=B, u_f32=self.u32, moments_f32=1), "posterior_sample")
            self.free(moments)
            e = self._unet(u)
            self.free(u)
        sa, s1 = one_step_scheduler_constants(self.ua.timestep)
        wpq, bpq = self.pv.small_f32("post_quant_conv")
        z = self.new(B, h8, w8, 8)
        self._add(O.ddpm_postquant(self.u32, e.t, z.t, wpq, bpq, n=B, hw=h8 * w8, lat=lat, ldu=lat, lde=e.c, ldy=8,
                                   sqrt_abar=sa, sqrt_1m_abar=s1, sf=sf, u_f32=1, e_f32=1), "ddpm_step+post_quant")
        self.free(e)
        y = self._vae_decoder(z, skips)
        self.free(z)
        self._add(O.nhwc_to_nchw(y.t, self.out, n=B, c=3, h=H, w=W, ldx=y.c, clamp=1, mul=0.5, add=0.5), "output.from_nhwc",
                  nbytes=B * H * W * (y.c * self.esz + 3 * self.out.element_size()))
        for t in self._zero_init:
            t.zero_()

    # ------------------------------------------------------------------ execution
    def _on_device(self):
        """Make the plan's device current for the launches (kernels run where their pointers live even when the caller's
        current device is another GPU); a no-op context for the CPU emulator."""
        import contextlib
        return torch.cuda.device(self.device) if torch.device(self.device).type == "cuda" else contextlib.nullcontext()

    def stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream if torch.device(self.device).type == "cuda" else 0

    # The LoRA scale / skip gamma / TwinConv fold `r` is DEVICE STATE of the model's packed weights, shared by every plan
    # (packer.py).  A plan therefore records the r it was built for (`self.r`: the stochastic r, 1.0 for deterministic plans)
    # and re-applies it through `before_run` (set by the owning model: Pix2Pix_Turbo.set_lora_scale) every time it executes:
    # holding a stochastic and a deterministic plan, or two r values, and alternating run()/replay() is safe.  One model is
    # single-stream with respect to r: the re-merge rewrites the weights in place on the current stream.
    before_run = None
    released = False

    def _prepare(self):
        if self.released:
            raise RuntimeError("ForwardPlan was released (plan-cache eviction / dtype change): its buffers are gone; get a new plan")
        if self.before_run is not None:
            self.before_run(self)

    def run(self):
        self._prepare()
        with self._on_device():
            self.lib.run(self.prog, self.stream())

    def capture(self):
        if self.released:
            raise RuntimeError("ForwardPlan was released: cannot capture a graph over freed buffers")
        if self.graph is None:
            with self._on_device():
                self.graph = self.lib.graph_create(self.prog)
        return self.graph

    def replay(self):
        self._prepare()
        with self._on_device():
            self.lib.graph_launch(self.capture(), self.stream())

    def run_timed(self):
        self._prepare()
        with self._on_device():
            return self.lib.run_timed(self.prog, self.stream())

    def release(self):
        """Destroy the captured hipGraph and drop the activation pool (plan-cache eviction).  The program's descriptors
        still hold raw pointers into the freed tensors: the plan refuses to run afterwards."""
        if self.graph is not None:
            self.lib.graph_destroy(self.graph)
            self.graph = None
        self.pool.all.clear()
        self.pool.free_lists.clear()
        self._keep.clear()
        self.released = True

