#!/usr/bin/env python3
"""Interleaved per-shape A/B of the shipped tile codes on the hot-path shapes (cdna_hip_programming.md rule 24: N variants x M
rounds in ONE process, report the median): the evidence behind the launch table of csrc/gemm_conv.hip.
usage: tools/tile_table.py [gemm|conv|all] [codes] [rounds]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fie_amd  # noqa: E402,F401
from fie_amd import hip  # noqa: E402
from tools.microbench import timeit  # noqa: E402

ctx = hip.context(0)
DEV = "cuda"
# (M, N, K): CFG batch 2 at 32x32 / 64x64 / 128x128 latents, then the same at 8 images per job (batch 16)
GEMMS = [(2048, 1280, 1280), (2048, 3840, 1280), (2048, 10240, 1280), (2048, 1280, 5120), (2048, 1280, 2560),
         (8192, 640, 640), (8192, 1920, 640), (8192, 5120, 640), (8192, 640, 2560), (32768, 320, 320), (32768, 320, 640), (16384, 1536, 512),
         (16384, 1280, 1280), (16384, 3840, 1280), (16384, 10240, 1280), (16384, 1280, 5120), (65536, 640, 640), (65536, 1920, 640),
         (65536, 5120, 640), (65536, 640, 2560)]
CONVS = [(2, 128, 128, 320, 320, 1, 0), (2, 128, 128, 640, 320, 1, 0), (2, 128, 128, 960, 320, 1, 0), (2, 64, 64, 640, 640, 1, 0), (2, 64, 64, 1280, 640, 1, 0),
         (2, 64, 64, 1920, 640, 1, 0), (2, 32, 32, 1280, 1280, 1, 0), (2, 32, 32, 2560, 1280, 1, 0), (2, 32, 32, 1280, 1280, 1, 1), (2, 64, 64, 640, 640, 1, 1),
         (2, 128, 128, 320, 320, 2, 0), (2, 64, 64, 640, 640, 2, 0),
         (1, 1024, 1024, 128, 128, 1, 0), (1, 512, 512, 256, 256, 1, 0), (1, 512, 512, 128, 256, 1, 0), (1, 256, 256, 512, 512, 1, 0), (1, 256, 256, 256, 512, 1, 0),
         (1, 128, 128, 512, 512, 1, 0), (1, 512, 512, 256, 256, 1, 1), (1, 256, 256, 512, 512, 1, 1), (1, 128, 128, 512, 512, 1, 1),
         (1, 1024, 1024, 128, 128, 2, 0), (1, 512, 512, 256, 256, 2, 0),
         (16, 32, 32, 1280, 1280, 1, 0), (16, 64, 64, 640, 640, 1, 0), (16, 128, 128, 320, 320, 1, 0)]


def table(name, cases, codes, rounds):
    for label, fn, flops in cases:
        t = {c: [] for c in codes}
        for _ in range(rounds):
            for c in codes:
                ctx.force_tile(c)
                t[c].append(timeit(fn, iters=10, warm=2))
        ctx.force_tile(0)
        med = {c: statistics.median(v) for c, v in t.items()}
        best = min((c for c in codes if c), key=lambda c: med[c])
        cells = "  ".join(f"{c}: {med[c] * 1e6:7.1f}us {flops / med[c] / 1e12:6.0f}TF" for c in codes)
        print(f"{name} {label:34s} {cells}   best {best} ({(med[0] / med[best] - 1) * 100:+.1f} % vs default)" if 0 in med else f"{name} {label} {cells} best {best}", flush=True)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    codes = [int(c) for c in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 42, 43, 51, 62, 81]
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    if which in ("gemm", "all"):
        cases = []
        for m, n, k in GEMMS:
            a = torch.randn(m, k, device=DEV, dtype=torch.float16)
            w = ctx.pack_linear(torch.randn(n, k, device=DEV, dtype=torch.float16) * k ** -0.5)
            out = torch.empty(m, n, device=DEV, dtype=torch.float16)
            cases.append((f"M={m} N={n} K={k}", (lambda a=a, w=w, n=n, out=out: ctx.gemm(a, w, n, out=out)), 2.0 * m * n * k))
        table("gemm", cases, codes, rounds)
    if which in ("conv", "all"):
        cases = []
        for b, h, w_, cin, cout, stride, ups in CONVS:
            x = torch.randn(b, h, w_, cin, device=DEV, dtype=torch.float16)
            wt = ctx.pack_conv3x3(torch.randn(cout, cin, 3, 3, device=DEV, dtype=torch.float16) * (9 * cin) ** -0.5)
            oh, ow = (h << ups) // stride, (w_ << ups) // stride
            cases.append((f"B={b} {h}x{w_} {cin}->{cout} s{stride} u{ups}",
                          (lambda x=x, wt=wt, cout=cout, stride=stride, ups=ups: ctx.conv3x3(x, wt, cout, stride=stride, upsample=bool(ups))),
                          2.0 * b * oh * ow * 9 * cin * cout))
        table("conv", cases, codes, rounds)


if __name__ == "__main__":
    main()
