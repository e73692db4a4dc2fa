"""ORACLE -- TEST INFRASTRUCTURE ONLY.

numpy restatement of what /root/reference/src/pipeline.py:196-210 (FastEditor.preprocess_image) computes with
OpenCV: ``cv2.cvtColor(RGB2GRAY)`` then ``cv2.Canny(gray, low, high)`` (defaults apertureSize=3,
L2gradient=False), stacked to 3 channels.

OpenCV (requirements.txt:16 `opencv-python>=4.8.0`) is absent from /root/reference and from this image, so this
follows the published algorithm of OpenCV 4.x ``modules/imgproc/src/canny.cpp`` and ``color_rgb.simd.hpp``:
**parity unpinned** (no cv2 here to mint golden vectors, and the reference ships no edge-map fixture).

Facts restated (the 15-bit gray coefficients are the OpenCV 4.x `RGB2Gray<uchar>` ones,
RY15/GY15/BY15 = 9798/19235/3735 with a 15-bit descale, which supersede the 14-bit 4899/9617/1868 set):
  * Sobel 3x3 dx, dy in int16 with BORDER_REPLICATE
  * magnitude |dx| + |dy| (L1), rows/cols outside the image count as 0
  * candidates: mag > low; direction test in fixed point, TG22 = round(tan(22.5deg) * 2^15)
      horizontal  if |dy|*2^15 <  |dx|*TG22              keep if m >  left  and m >= right
      vertical    if |dy|*2^15 >  |dx|*(TG22 + 2*2^15)   keep if m >  up    and m >= down
      diagonal    otherwise, s = sign(dx ^ dy)            keep if m >  (up, j-s) and m > (down, j+s)
  * hysteresis: 8-connected growth from pixels with mag > high through candidates; output 0 / 255
"""
import numpy as np
from scipy import ndimage

CANNY_SHIFT = 15
TG22 = int(0.4142135623730950488016887242097 * (1 << CANNY_SHIFT) + 0.5)


def rgb_to_gray(rgb):
    r = rgb[..., 0].astype(np.int64)
    g = rgb[..., 1].astype(np.int64)
    b = rgb[..., 2].astype(np.int64)
    return ((r * 9798 + g * 19235 + b * 3735 + (1 << 14)) >> 15).astype(np.uint8)


def sobel3(gray):
    p = np.pad(gray.astype(np.int32), 1, mode="edge")
    tl, tc, tr = p[:-2, :-2], p[:-2, 1:-1], p[:-2, 2:]
    ml, mr = p[1:-1, :-2], p[1:-1, 2:]
    bl, bc, br = p[2:, :-2], p[2:, 1:-1], p[2:, 2:]
    dx = (tr + 2 * mr + br) - (tl + 2 * ml + bl)
    dy = (bl + 2 * bc + br) - (tl + 2 * tc + tr)
    return dx, dy


def canny(gray, low, high):
    if low > high:
        low, high = high, low
    low, high = int(np.floor(low)), int(np.floor(high))
    dx, dy = sobel3(gray)
    mag = np.abs(dx) + np.abs(dy)
    m = np.pad(mag, 1, mode="constant")           # zero border
    c = m[1:-1, 1:-1]
    left, right = m[1:-1, :-2], m[1:-1, 2:]
    up, down = m[:-2, 1:-1], m[2:, 1:-1]
    ul, ur = m[:-2, :-2], m[:-2, 2:]
    dl, dr = m[2:, :-2], m[2:, 2:]
    x = np.abs(dx).astype(np.int64)
    y = np.abs(dy).astype(np.int64) << CANNY_SHIFT
    tg22x = x * TG22
    tg67x = tg22x + (x << (CANNY_SHIFT + 1))
    horiz = y < tg22x
    vert = (~horiz) & (y > tg67x)
    diag = ~(horiz | vert)
    neg = (dx ^ dy) < 0                           # s = -1: compare (up, j+1) and (down, j-1)
    keep_h = (c > left) & (c >= right)
    keep_v = (c > up) & (c >= down)
    keep_d = np.where(neg, (c > ur) & (c > dl), (c > ul) & (c > dr))
    cand = (c > low) & ((horiz & keep_h) | (vert & keep_v) | (diag & keep_d))
    strong = cand & (c > high)
    lab, n = ndimage.label(cand, structure=np.ones((3, 3), dtype=bool))
    if n == 0:
        return np.zeros_like(gray)
    has_strong = np.zeros(n + 1, dtype=bool)
    has_strong[np.unique(lab[strong])] = True
    has_strong[0] = False
    return (has_strong[lab].astype(np.uint8)) * 255


def canny_rgb(rgb_u8, low=100, high=200):
    """(H,W,3) uint8 -> (H,W,3) uint8 edge map, as preprocess_image returns it."""
    gray = rgb_to_gray(rgb_u8) if rgb_u8.ndim == 3 else rgb_u8
    e = canny(gray, low, high)
    return np.stack([e, e, e], axis=2)
