#!/usr/bin/env python3
"""Where does a GEMM / conv kernel's time go?  Times one shape with the timing-only probes of fie_debug_gemm_probe:
normal | DMA loads dropped (instruction stream, waits and barriers unchanged: the issue / sync floor) | every tile loading tile
(0,0)'s operands (all L2 hits: the cost of the misses).  usage: tools/gemm_probe.py M N K code[,code...] | conv B H W Cin Cout code[,...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fie_amd  # noqa: E402,F401
from fie_amd import hip  # noqa: E402
from tools.microbench import timeit  # noqa: E402

ctx = hip.context(0)
args = sys.argv[1:]
if args[0] == "conv":
    b, h, w_, cin, cout = map(int, args[1:6])
    codes = [int(c) for c in args[6].split(",")]
    x = torch.randn(b, h, w_, cin, device="cuda", dtype=torch.float16)
    wt = ctx.pack_conv3x3(torch.randn(cout, cin, 3, 3, device="cuda", dtype=torch.float16) * (9 * cin) ** -0.5)
    fl = 2.0 * b * h * w_ * 9 * cin * cout
    fn = lambda: ctx.conv3x3(x, wt, cout)
    name = f"conv B={b} {h}x{w_} {cin}->{cout}"
else:
    m, n, k = map(int, args[0:3])
    codes = [int(c) for c in args[3].split(",")]
    a = torch.randn(m, k, device="cuda", dtype=torch.float16)
    w = ctx.pack_linear(torch.randn(n, k, device="cuda", dtype=torch.float16) * k ** -0.5)
    out = torch.empty(m, n, device="cuda", dtype=torch.float16)
    fl = 2.0 * m * n * k
    fn = lambda: ctx.gemm(a, w, n, out=out)
    name = f"gemm M={m} N={n} K={k}"
for code in codes:
    ctx.force_tile(code)
    row = []
    for probe, tag in ((0, "normal"), (2, "all-L2-hit"), (1, "loads-dropped"), (3, "no-DMA-issue"), (4, "no-epilogue")):
        ctx.gemm_probe(probe)
        dt = min(timeit(fn, iters=20) for _ in range(3))
        row.append(f"{tag}: {dt * 1e6:8.1f} us {fl / dt / 1e12:7.1f} TF")
    ctx.gemm_probe(0)
    print(f"{name} code {code}: " + "   ".join(row), flush=True)
ctx.force_tile(0)
