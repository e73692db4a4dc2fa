#!/usr/bin/env python3
"""Warm versus cold weights: the per-op microbenchmarks re-run one layer, so its weights sit in the 256 MB Infinity Cache; inside
the UNet every layer's weights come from HBM (2.6 GB per evaluation).  This times the same launch rotating over enough copies of
the weight matrix (> 600 MB) that every launch reads it cold, next to the usual warm number.
usage: tools/cold_weights.py [codes]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fie_amd  # noqa: E402,F401
from fie_amd import hip  # noqa: E402

ctx = hip.context(0)
DEV = "cuda"


def time_rot(fns, iters):
    for f in fns:
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fns[i % len(fns)]()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def main():
    codes = [int(c) for c in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 42, 62, 96, 51, 95]
    cases = []
    for m, n, k in [(2048, 1280, 1280), (2048, 10240, 1280), (2048, 1280, 5120), (8192, 5120, 640), (8192, 640, 2560), (8192, 640, 640)]:
        copies = max(2, int(600e6 / (n * k * 2)) + 1)
        a = torch.randn(m, k, device=DEV, dtype=torch.float16)
        ws = [ctx.pack_linear(torch.randn(n, k, device=DEV, dtype=torch.float16) * k ** -0.5) for _ in range(copies)]
        out = torch.empty(m, n, device=DEV, dtype=torch.float16)
        cases.append((f"gemm M={m} N={n} K={k} ({copies} copies)", [lambda w=w, a=a, n=n, out=out: ctx.gemm(a, w, n, out=out) for w in ws], 2.0 * m * n * k))
    for b, h, cin, cout in [(2, 32, 640, 1280), (2, 32, 1280, 1280), (2, 32, 1920, 1280), (2, 32, 2560, 1280), (2, 64, 640, 640), (2, 64, 1280, 640), (2, 128, 320, 320)]:
        copies = max(2, int(600e6 / (cout * cin * 18)) + 1)
        x = torch.randn(b, h, h, cin, device=DEV, dtype=torch.float16)
        ws = [ctx.pack_conv3x3(torch.randn(cout, cin, 3, 3, device=DEV, dtype=torch.float16) * (9 * cin) ** -0.5) for _ in range(copies)]
        out = torch.empty(b, h, h, cout, device=DEV, dtype=torch.float16)
        cases.append((f"conv B={b} {h}x{h} {cin}->{cout} ({copies} copies)", [lambda w=w, x=x, cout=cout, out=out: ctx.conv3x3(x, w, cout, out=out) for w in ws], 2.0 * b * h * h * cout * cin * 9))
    for label, fns, flops in cases:
        cells = []
        for c in codes:
            ctx.force_tile(c)
            try:
                fns[0]()
            except hip.FieError:                       # the code does not serve this view (e.g. 63: GEMM only)
                cells.append(f"{c}: n/a")
                continue
            warm = statistics.median(time_rot(fns[:1], 20) for _ in range(3))
            cold = statistics.median(time_rot(fns, max(20, len(fns))) for _ in range(3))
            cells.append(f"{c}: warm {warm * 1e6:6.1f} cold {cold * 1e6:6.1f} us ({(cold / warm - 1) * 100:+.0f} %)")
        ctx.force_tile(0)
        print(f"{label:46s} " + "  ".join(cells), flush=True)


if __name__ == "__main__":
    main()
