"""PIL's LANCZOS resampling of 8-bit RGB images, restated in numpy (test infrastructure; see oracle/__init__.py).

The reference resizes on the host with Pillow before / after the generator:
  src/inference_paired.py:38-41    input_image.resize((w - w % 8, h - h % 8), Image.LANCZOS)
  src/inference_unpaired.py:40,53  transforms.Resize(..., interpolation=LANCZOS) / output_pil.resize(input size, Image.LANCZOS)
  src/my_utils/training_utils.py:184-215 (build_transform: "resize_512x512", "resize_256", ...)
Pillow is a third-party dependency of the reference that IS installed here (12.2), so this restatement is PINNED:
tests/test_oracle_kats.py compares it bit for bit with ``Image.resize(..., Image.LANCZOS)``.

Algorithm (Pillow src/libImaging/Resample.c): separable, horizontal pass then vertical pass on uint8 data;
per output coordinate xx: center = (xx + 0.5) * scale, support = 3 * max(scale, 1), taps xmin..xmin+xmax-1 with weights
lanczos((x + xmin - center + 0.5) / max(scale, 1)) normalised to sum 1 in double, then rounded to 22-bit fixed point
(``normalize_coeffs_8bpc``); a pixel = clip8((2^21 + sum_x in[x] * k[x]) >> 22) in 32-bit integer arithmetic.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _sinc(x):
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def lanczos(x):
    if -3.0 <= x < 3.0:
        return _sinc(x) * _sinc(x / 3)
    return 0.0


def precompute_coeffs(in_size, out_size):
    """(ksize, bounds int32 [out, 2] = (first tap, tap count), fixed-point weights int32 [out, ksize])."""
    in0, in1 = 0.0, float(in_size)
    filterscale = scale = (in1 - in0) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 3.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.float64)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
        ww = 0.0
        ss = 1.0 / filterscale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        for x in range(xmax):
            w = lanczos((x + xmin - center + 0.5) * ss)
            kk[xx, x] = w
            ww += w
        if ww != 0.0:
            for x in range(xmax):
                kk[xx, x] /= ww
        bounds[xx] = (xmin, xmax)
    ik = np.where(kk < 0, (-0.5 + kk * (1 << PRECISION_BITS)).astype(np.int64),
                  (0.5 + kk * (1 << PRECISION_BITS)).astype(np.int64)).astype(np.int32)
    return ksize, bounds, ik


def _resample_axis(img, bounds, ik, axis):
    n_out = len(bounds)
    shape = (img.shape[0], n_out, img.shape[2]) if axis == 1 else (n_out, img.shape[1], img.shape[2])
    out = np.empty(shape, np.uint8)
    for o, (lo, cnt) in enumerate(bounds):
        acc = np.full(shape[:axis] + shape[axis + 1:], 1 << (PRECISION_BITS - 1), np.int64)
        for t in range(cnt):
            acc += np.take(img, lo + t, axis=axis).astype(np.int64) * int(ik[o, t])
        v = np.clip(acc >> PRECISION_BITS, 0, 255)
        if axis == 1:
            out[:, o, :] = v
        else:
            out[o] = v
    return out


def lanczos_resize_u8(img, out_w, out_h):
    """img uint8 [H, W, C] -> uint8 [out_h, out_w, C], bit-identical to ``Image.fromarray(img).resize((out_w, out_h), Image.LANCZOS)``."""
    h, w = img.shape[:2]
    cur = img
    if out_w != w:
        _, b, k = precompute_coeffs(w, out_w)
        cur = _resample_axis(cur, b, k, 1)
    if out_h != h:
        _, b, k = precompute_coeffs(h, out_h)
        cur = _resample_axis(cur, b, k, 0)
    return cur
