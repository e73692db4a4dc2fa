#!/usr/bin/env python3
"""LayerNorm launch time at the UNet's shapes (100 back-to-back launches in one hipGraph).  FIE_LIB_PATH selects another build for an A/B.  usage: tools/ln_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fie_amd  # noqa: E402,F401
from fie_amd import hip  # noqa: E402
from bench import _graph_ms  # noqa: E402

ctx = hip.context(0)
for rows, c in [(2048, 1280), (8192, 640), (154, 1280), (154, 768)]:
    x = torch.randn(rows, c, device="cuda", dtype=torch.float16)
    g, b = torch.randn(c, device="cuda", dtype=torch.float16), torch.randn(c, device="cuda", dtype=torch.float16)
    ref = torch.nn.functional.layer_norm(x.float(), (c,), g.float(), b.float(), 1e-5)
    err = (ctx.layernorm(x, g, b).float() - ref).abs().max().item()
    y = torch.empty_like(x)

    def many():
        for _ in range(100):
            ctx.layernorm(x, g, b, out=y)
    dt = _graph_ms(many)[0] * 1e-3 / 100
    print(f"layernorm rows={rows} C={c}: {dt * 1e6:5.2f} us  {4.0 * rows * c / dt / 1e9:6.0f} GB/s  max err {err:.2e}", flush=True)
