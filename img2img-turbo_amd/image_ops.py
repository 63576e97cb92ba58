"""Device-side image pre/post-processing of the callers (row f1): LANCZOS resize of uint8 HWC image batches, bit-identical to
Pillow, which the reference applies on the host before and after the generator:

  src/inference_paired.py:38-41    input_image.resize((w - w % 8, h - h % 8), Image.LANCZOS)
  src/inference_unpaired.py:40,53  transforms.Resize((512, 512), LANCZOS) ... output_pil.resize(input size, Image.LANCZOS)

The tap windows and 22-bit fixed-point weights are computed here in double precision exactly as Pillow's
``precompute_coeffs`` / ``normalize_coeffs_8bpc`` do (src/libImaging/Resample.c); the two integer passes run in
csrc/resize.hip through ``i2i_resize_u8``.  Pure plumbing otherwise: no pixel arithmetic happens in Python.
"""
import ctypes as C
import functools
import math

import numpy as np
import torch

from . import _capi

PRECISION_BITS = 32 - 8 - 2


def _sinc(x):
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def _lanczos(x):
    if -3.0 <= x < 3.0:
        return _sinc(x) * _sinc(x / 3)
    return 0.0


@functools.lru_cache(maxsize=64)
def lanczos_coeffs(in_size, out_size):
    """Pillow's per-output-coordinate tap window and weights for one axis: (ksize, bounds int32 [out, 2], weights int32 [out, ksize])."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 3.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.float64)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        ww = 0.0
        for x in range(xmax):
            w = _lanczos((x + xmin - center + 0.5) * ss)
            kk[xx, x] = w
            ww += w
        if ww != 0.0:
            for x in range(xmax):
                kk[xx, x] /= ww
        bounds[xx] = (xmin, xmax)
    ik = np.where(kk < 0, (-0.5 + kk * (1 << PRECISION_BITS)).astype(np.int64),
                  (0.5 + kk * (1 << PRECISION_BITS)).astype(np.int64)).astype(np.int32)
    return ksize, bounds, ik


import collections
import contextlib

_DEV_TABLES = collections.OrderedDict()      # (in, out, device) -> coefficient tables on the device; small LRU: a server fed
_DEV_TABLES_MAX = 64                         # arbitrary input sizes must not grow device memory without bound


def _tables(in_size, out_size, device):
    key = (in_size, out_size, str(device))
    if key in _DEV_TABLES:
        _DEV_TABLES.move_to_end(key)
    else:
        ksize, b, k = lanczos_coeffs(in_size, out_size)
        _DEV_TABLES[key] = (ksize, torch.from_numpy(b.copy()).to(device), torch.from_numpy(k.copy()).to(device))
        while len(_DEV_TABLES) > _DEV_TABLES_MAX:
            _DEV_TABLES.popitem(last=False)
    return _DEV_TABLES[key]


def lanczos_resize_u8(images, size, lib=None):
    """images: uint8 [N, H, W, C] (C <= 4) on the device -> uint8 [N, size[1], size[0], C]; ``size`` = (width, height) as PIL takes it.
    Bit-identical to ``Image.resize(size, Image.LANCZOS)`` per image.  Asynchronous on the current stream."""
    assert images.dtype == torch.uint8 and images.dim() == 4 and images.shape[-1] <= 4
    lib = lib or _capi.default_library()
    n, h, w, c = images.shape
    ow, oh = int(size[0]), int(size[1])
    cur = images.contiguous()
    dev = cur.device
    # the GPU library takes device tensors only, the CPU emulator host tensors only: no silent mismatch
    assert (lib.backend == "emu") == (dev.type == "cpu"), "library backend %s cannot take a tensor on %s" % (lib.backend, dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream) if dev.type == "cuda" else None
    # the launches go to the GPU that owns the images even when another one is current
    with (torch.cuda.device(dev) if dev.type == "cuda" else contextlib.nullcontext()):
        for axis, n_in, n_out in ((1, w, ow), (0, h, oh)):          # Pillow's order: horizontal pass, then vertical pass
            if n_in == n_out:
                continue
            ksize, bounds, coeffs = _tables(n_in, n_out, dev)
            hin, win = cur.shape[1], cur.shape[2]
            out = torch.empty((n, hin, n_out, c) if axis == 1 else (n, n_out, win, c), dtype=torch.uint8, device=dev)
            p = _capi.ResizeU8Params()
            p.src, p.dst, p.n, p.hin, p.win, p.c = cur.data_ptr(), out.data_ptr(), n, hin, win, c
            p.axis, p.nout, p.ksize, p.bounds, p.coeffs = axis, n_out, ksize, bounds.data_ptr(), coeffs.data_ptr()
            lib.check(lib.lib.i2i_resize_u8(C.addressof(p), 0, stream))
            cur = out
    return cur


def resize_to_multiple_of_8(images, lib=None):
    """src/inference_paired.py:38-41 on the device: LANCZOS resize to (w - w % 8, h - h % 8)."""
    h, w = images.shape[1], images.shape[2]
    return lanczos_resize_u8(images, (w - w % 8, h - h % 8), lib)


def apply_image_prep(images, image_prep, lib=None):
    """The inference-time options of the reference's ``build_transform`` (src/my_utils/training_utils.py:184-215, used at
    src/inference_unpaired.py:40) on a uint8 HWC batch on the device:
      "resize_512x512" / "resize_512", "resize_256x256" / "resize_256"  -> Resize((S, S), LANCZOS)
      "resized_crop_512"  -> Resize(512, LANCZOS) (shorter side to 512, torchvision's size rule) then CenterCrop(512)
      "no_resize"         -> identity
    The random training augmentation ("resize_286_randomcrop_256x256_hflip") is out of scope (training)."""
    h, w = images.shape[1], images.shape[2]
    if image_prep == "no_resize":
        return images
    if image_prep in ("resize_256", "resize_256x256"):
        return lanczos_resize_u8(images, (256, 256), lib)
    if image_prep in ("resize_512", "resize_512x512"):
        return lanczos_resize_u8(images, (512, 512), lib)
    if image_prep == "resized_crop_512":
        size = 512
        short, long = (w, h) if w <= h else (h, w)
        new_long = int(size * long / short)                      # torchvision _compute_resized_output_size
        ow, oh = (size, new_long) if w <= h else (new_long, size)
        r = lanczos_resize_u8(images, (ow, oh), lib)
        top, left = int(round((oh - size) / 2.0)), int(round((ow - size) / 2.0))      # torchvision center_crop
        return r[:, top:top + size, left:left + size].contiguous()
    raise ValueError("image_prep %r is a training-time augmentation or unknown (supported: resize_256[x256], resize_512[x512], "
                     "resized_crop_512, no_resize)" % (image_prep,))
