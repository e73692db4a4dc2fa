#!/usr/bin/env python3
"""GroupNorm (+SiLU) launch times at the VAE's and the UNet's shapes (20 launches per hipGraph replay).  FIE_LIB_PATH selects another build for an A/B.
usage: tools/gn_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fie_amd  # noqa: E402,F401
from fie_amd import hip  # noqa: E402
from bench import _graph_ms  # noqa: E402

ctx = hip.context(0)
for b, hw, c in [(1, 1024, 128), (1, 512, 256), (1, 256, 512), (1, 128, 512), (2, 128, 320), (2, 64, 640), (2, 32, 1280)]:
    x = torch.randn(b, hw, hw, c, device="cuda", dtype=torch.float16)
    g, bb = torch.randn(c, device="cuda", dtype=torch.float16), torch.randn(c, device="cuda", dtype=torch.float16)
    ref = torch.nn.functional.silu(torch.nn.functional.group_norm(x[:, :64].permute(0, 3, 1, 2).float(), 32, g.float(), bb.float(), 1e-5)) if hw <= 64 else None
    n = 20

    def many():
        for _ in range(n):
            y = ctx.groupnorm(x, g, bb, 32, 1e-5, True)
        return y
    ms, y, _ = _graph_ms(many)
    us = ms * 1e3 / n
    print(f"groupnorm+silu B={b} {hw}x{hw} C={c}: {us:7.1f} us per call (all passes)  {4.0 * x.numel() / us / 1e3:6.0f} GB/s algorithmic", flush=True)
