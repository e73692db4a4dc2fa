#!/usr/bin/env python3
"""DVFS check (guide rule 25): the same kernel on zero-filled vs random operands; a large gap means the chip is holding its
clock down under MFMA load (power), not that the kernel has idle cycles left."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fie_amd  # noqa: E402,F401
from fie_amd import hip  # noqa: E402
from tools.microbench import timeit  # noqa: E402

ctx = hip.context(0)
for name, (b, h, w, cin, cout) in {"vae 256^2 512->512": (1, 256, 256, 512, 512), "unet 64^2 1280->640": (2, 64, 64, 1280, 640)}.items():
    for fill in ("random", "zero"):
        x = torch.randn(b, h, w, cin, device="cuda", dtype=torch.float16) if fill == "random" else torch.zeros(b, h, w, cin, device="cuda", dtype=torch.float16)
        wt = torch.randn(cout, cin, 3, 3, device="cuda", dtype=torch.float16) * (9 * cin) ** -0.5
        if fill == "zero":
            wt.zero_()
        wp = ctx.pack_conv3x3(wt)
        dt = timeit(lambda: ctx.conv3x3(x, wp, cout), iters=20)
        print(f"{name:22s} {fill:6s}: {dt * 1e6:8.1f} us {2 * b * h * w * 9 * cin * cout / dt / 1e12:7.1f} TF", flush=True)
m = n = k = 4096
for fill in ("random", "zero"):
    a = torch.randn(m, k, device="cuda", dtype=torch.float16) if fill == "random" else torch.zeros(m, k, device="cuda", dtype=torch.float16)
    w = torch.randn(n, k, device="cuda", dtype=torch.float16) * 0.02 if fill == "random" else torch.zeros(n, k, device="cuda", dtype=torch.float16)
    wp = ctx.pack_linear(w)
    out = torch.empty(m, n, device="cuda", dtype=torch.float16)
    dt = timeit(lambda: ctx.gemm(a, wp, n, out=out), iters=20)
    print(f"gemm 4096^3            {fill:6s}: {dt * 1e6:8.1f} us {2 * m * n * k / dt / 1e12:7.1f} TF", flush=True)
