#!/usr/bin/env python3
"""Is the v3 GEMM main loop bound by bytes in flight?  Same kernel, occupancy lowered by padding its dynamic LDS.
usage: tools/occupancy_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fie_amd  # noqa: E402,F401
from fie_amd import hip  # noqa: E402
from tools.microbench import timeit  # noqa: E402

ctx = hip.context(0)
for m, n, k in [(4096, 4096, 4096), (2048, 10240, 1280), (8192, 5120, 640), (16384, 10240, 1280)]:
    a = torch.randn(m, k, device="cuda", dtype=torch.float16)
    w = ctx.pack_linear(torch.randn(n, k, device="cuda", dtype=torch.float16) * k ** -0.5)
    out = torch.empty(m, n, device="cuda", dtype=torch.float16)
    for code, pads in [(43, (0, 24 * 1024, 64 * 1024)), (42, (0, 32 * 1024)), (62, (0,)), (61, (0,))]:
        res = []
        for pad in pads:
            hip.lib().fie_debug_force_tile(code)
            hip.lib().fie_debug_extra_lds(pad)
            dt = timeit(lambda: ctx.gemm(a, w, n, out=out))
            res.append(f"+{pad // 1024:3d} KiB: {dt * 1e6:7.1f} us {2 * m * n * k / dt / 1e12:6.1f} TF")
        print(f"M={m} N={n} K={k} tile {code}: " + "   ".join(res), flush=True)
hip.lib().fie_debug_extra_lds(0)
hip.lib().fie_debug_force_tile(0)
