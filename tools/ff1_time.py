#!/usr/bin/env python3
"""The FF1 GEGLU projection (M 2048 x N 10240 x K 1280 and the 64x64-level M 8192 x N 5120 x K 640) on given tile codes, cold weights.  FIE_LIB_PATH selects a build.
usage: tools/ff1_time.py [codes]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fie_amd  # noqa: E402,F401
from fie_amd import hip  # noqa: E402
from tools.cold_weights import time_rot  # noqa: E402

ctx = hip.context(0)
codes = [int(c) for c in sys.argv[1].split(",")] if len(sys.argv) > 1 else [63, 64]
for m, n, k in [(2048, 10240, 1280), (8192, 5120, 640)]:
    copies = max(2, int(600e6 / (n * k * 2)) + 1)
    a = torch.randn(m, k, device="cuda", dtype=torch.float16)
    ws = [ctx.pack_linear(torch.randn(n, k, device="cuda", dtype=torch.float16) * k ** -0.5, geglu=True) for _ in range(copies)]
    out = torch.empty(m, n // 2, device="cuda", dtype=torch.float16)
    bias = torch.randn(n, device="cuda", dtype=torch.float16)
    fns = [lambda w=w: ctx.gemm(a, w, n, out=out, bias=bias, act=hip.ACT_GEGLU) for w in ws]
    for code in codes:
        ctx.force_tile(code)
        fns[0]()
        cold = statistics.median(time_rot(fns, max(40, len(fns))) for _ in range(5))
        print(f"FF1 GEGLU M={m} N={n} K={k} code={code}: {cold * 1e6:6.1f} us  {2.0 * m * n * k / cold / 1e12:5.0f} TF/s", flush=True)
    ctx.force_tile(0)
