"""ORACLE -- TEST INFRASTRUCTURE ONLY.

Acceptance metrics restated from /root/reference/src/metrics.py:215-239 (SSIM), :291-347 (PSNR, MSE): both images
LANCZOS-resized to 512x512, RGB in [0,1].  The reference delegates to torchmetrics (absent here), whose defaults
are restated: SSIM = 11x11 Gaussian window, sigma 1.5, k1 0.01, k2 0.03, data_range 1.0, reflect padding with the
padded border cropped before the mean.  **parity unpinned** against torchmetrics itself.
"""
import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image


def _prep(img, size=(512, 512)):
    if not isinstance(img, Image.Image):
        img = Image.fromarray(np.asarray(img))
    if size is not None and img.size != size:
        img = img.resize(size, Image.LANCZOS)
    a = np.asarray(img).astype(np.float32) / 255.0
    return torch.from_numpy(a).permute(2, 0, 1)[None]


def _gauss(k=11, sigma=1.5):
    d = torch.arange((1 - k) / 2, (1 + k) / 2, 1.0, dtype=torch.float32)
    g = torch.exp(-(d / sigma) ** 2 / 2)
    g = (g / g.sum())[None]
    return (g.T @ g)


def ssim(img1, img2, size=(512, 512), data_range=1.0, k1=0.01, k2=0.03):
    x, y = _prep(img1, size), _prep(img2, size)
    c = x.shape[1]
    win = _gauss().expand(c, 1, 11, 11)
    pad = 5
    xp, yp = F.pad(x, (pad,) * 4, mode="reflect"), F.pad(y, (pad,) * 4, mode="reflect")
    stack = torch.cat([xp, yp, xp * xp, yp * yp, xp * yp])
    out = F.conv2d(stack, win, groups=c)
    mu_x, mu_y, e_xx, e_yy, e_xy = out.split(1)
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    sxx, syy, sxy = e_xx - mu_x ** 2, e_yy - mu_y ** 2, e_xy - mu_x * mu_y
    m = ((2 * mu_x * mu_y + c1) * (2 * sxy + c2)) / ((mu_x ** 2 + mu_y ** 2 + c1) * (sxx + syy + c2))
    return m[..., pad:-pad, pad:-pad].mean().item()


def mse(img1, img2, size=(512, 512)):
    return ((_prep(img1, size) - _prep(img2, size)) ** 2).mean().item()


def psnr(img1, img2, size=(512, 512)):
    m = mse(img1, img2, size)
    return float("inf") if m == 0 else 10.0 * np.log10(1.0 / m)
