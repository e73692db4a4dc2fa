"""Coefficient tables of Pillow's 8-bit LANCZOS resample, restated (host side of csrc/resize.hip).

Follows Pillow `src/libImaging/Resample.c`: `precompute_coeffs` (support 3 x max(scale, 1), bounds by truncating
`center -/+ support + 0.5`, weights normalised to sum 1) and `normalize_coeffs_8bpc` (22-bit fixed point, round half away from
zero).  The reference reaches it through `image.resize((1024, 1024), Image.LANCZOS)` (`src/pipeline.py:251`).  Pure Python / libm
doubles, as Pillow's C: the tables -- and therefore the device result -- are bit-exact with Pillow (tests/test_cabi_cpu.py)."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
LANCZOS_SUPPORT = 3.0


def _sinc(x):
    if x == 0.0:
        return 1.0
    x *= math.pi
    return math.sin(x) / x


def _lanczos(x):
    return _sinc(x) * _sinc(x / 3) if -3.0 <= x < 3.0 else 0.0


def coefficients(in_size, out_size):
    """-> (kk int32 [out_size, ksize], bounds int32 [out_size, 2] = (first input index, tap count), ksize)."""
    scale = float(np.float32(in_size) - np.float32(0)) / out_size          # Pillow's box is float32
    filterscale = max(scale, 1.0)
    support = LANCZOS_SUPPORT * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)                         # int(): truncation, as the C cast
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_lanczos((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x, v in enumerate(w):
            k = v / ww if ww != 0.0 else v
            kk[xx, x] = int(-0.5 + k * (1 << PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return kk, bounds, ksize


def resample_numpy(rgb, out_h, out_w):
    """The two passes in numpy with the tables above (CPU checker of the tables; the product path is the HIP kernel)."""
    a = np.asarray(rgb, dtype=np.uint8)
    h, w, _ = a.shape

    def one_pass(img, axis_len, out_len):
        kk, bounds, _ = coefficients(axis_len, out_len)
        out = np.empty((img.shape[0], out_len, 3), dtype=np.uint8)
        for o in range(out_len):
            x0, n = bounds[o]
            acc = (img[:, x0:x0 + n, :].astype(np.int64) * kk[o, :n, None].astype(np.int64)).sum(axis=1) + (1 << (PRECISION_BITS - 1))
            out[:, o, :] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
        return out

    if w != out_w:
        a = one_pass(a, w, out_w)
    if h != out_h:
        a = one_pass(a.transpose(1, 0, 2), h, out_h).transpose(1, 0, 2)
    return np.ascontiguousarray(a)
