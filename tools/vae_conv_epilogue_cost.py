#!/usr/bin/env python3
"""What the GroupNorm-sums epilogue costs on the VAE's largest convs (1024^2 x 128 -> 128, 512^2 x 256 -> 256): the same launch with / without gn_groups, graph-timed."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fie_amd  # noqa: E402,F401
from fie_amd import hip  # noqa: E402
from bench import _graph_ms  # noqa: E402

ctx = hip.context(0)
for hw, c in ((1024, 128), (512, 256), (256, 512)):
    x = torch.randn(1, hw, hw, c, device="cuda", dtype=torch.float16)
    w = ctx.pack_conv3x3(torch.randn(c, c, 3, 3, device="cuda", dtype=torch.float16) * (9 * c) ** -0.5)
    bias = torch.randn(c, device="cuda", dtype=torch.float16)
    res = torch.randn(1, hw, hw, c, device="cuda", dtype=torch.float16)
    for label, kw in (("plain", {}), ("bias", dict(bias=bias)), ("bias+gn", dict(bias=bias, gn_groups=32)), ("bias+res+gn", dict(bias=bias, residual=res, gn_groups=32))):
        def many():
            for _ in range(5):
                y = ctx.conv3x3(x, w, c, **kw)
            return y
        many()
        ms = _graph_ms(many)[0]
        print(f"conv {hw}^2 x {c}: {label:12s} {ms * 1e3 / 5:7.1f} us  ({hip.last_gemm_kernel(ctx)})", flush=True)
