#!/usr/bin/env python3
"""Does an exact-fit grid (one tile per CU) pay for the N = 1280 projections?  Times tile codes on M = 2048 with N chosen so that the grid is
exactly 256 tiles, next to N = 1280 (cold weights, rotating copies).  usage: tools/fit_probe.py"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fie_amd  # noqa: E402,F401
from fie_amd import hip  # noqa: E402
from tools.cold_weights import time_rot  # noqa: E402

ctx = hip.context(0)
for k in (1280, 5120):
    for code, ns in ((42, (1280,)), (47, (1280,)), (48, (1280,))):
        for n in ns:
            m = 2048
            copies = max(2, int(600e6 / (n * k * 2)) + 1)
            a = torch.randn(m, k, device="cuda", dtype=torch.float16)
            res = torch.randn(m, n, device="cuda", dtype=torch.float16)
            ws = [ctx.pack_linear(torch.randn(n, k, device="cuda", dtype=torch.float16) * k ** -0.5) for _ in range(copies)]
            out = torch.empty(m, n, device="cuda", dtype=torch.float16)
            bias = torch.randn(n, device="cuda", dtype=torch.float16)
            fns = [lambda w=w: ctx.gemm(a, w, n, out=out, residual=res, bias=bias) for w in ws]
            ctx.force_tile(code)
            try:
                fns[0]()
            except hip.FieError as e:
                print(code, n, "n/a", e)
                continue
            cells = []
            for pre in (True, True):
                ctx.epi_prefetch = pre
                cold = statistics.median(time_rot(fns, max(40, len(fns))) for _ in range(5))
                cells.append(f"{'pre' if pre else 'epi'} {cold * 1e6:6.1f} us")
            print(f"K={k} code={code} N={n}: " + "  ".join(cells), flush=True)
            ctx.force_tile(0)
