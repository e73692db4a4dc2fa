#!/usr/bin/env python3
"""Launch ONE GEMM / conv shape a few times (for rocprofv3 --pmc runs).
usage: tools/one_kernel.py gemm M N K code [iters] [geglu] | conv B H W Cin Cout code [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fie_amd  # noqa: E402,F401
from fie_amd import hip  # noqa: E402

ctx = hip.context(0)
kind = sys.argv[1]
if kind == "gemm":
    m, n, k, code = map(int, sys.argv[2:6])
    iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
    a = torch.randn(m, k, device="cuda", dtype=torch.float16)
    w = ctx.pack_linear(torch.randn(n, k, device="cuda", dtype=torch.float16) * k ** -0.5)
    out = torch.empty(m, n, device="cuda", dtype=torch.float16)
    geglu = len(sys.argv) > 7 and sys.argv[7] == "geglu"       # the production FF1 epilogue: + bias, GEGLU
    if geglu:
        w = ctx.pack_linear(torch.randn(n, k, device="cuda", dtype=torch.float16) * k ** -0.5, geglu=True)
        bias = torch.randn(n, device="cuda", dtype=torch.float16)
        out = torch.empty(m, n // 2, device="cuda", dtype=torch.float16)
    ctx.force_tile(code)
    for _ in range(iters):
        if geglu:
            ctx.gemm(a, w, n, out=out, bias=bias, act=hip.ACT_GEGLU)
        else:
            ctx.gemm(a, w, n, out=out)
else:
    b, h, w_, cin, cout, code = map(int, sys.argv[2:8])
    iters = int(sys.argv[8]) if len(sys.argv) > 8 else 5
    x = torch.randn(b, h, w_, cin, device="cuda", dtype=torch.float16)
    wt = ctx.pack_conv3x3(torch.randn(cout, cin, 3, 3, device="cuda", dtype=torch.float16) * (9 * cin) ** -0.5)
    ctx.force_tile(code)
    for _ in range(iters):
        ctx.conv3x3(x, wt, cout)
torch.cuda.synchronize()
