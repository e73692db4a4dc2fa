#!/usr/bin/env python3
"""GroupNorm + SiLU + conv3x3: the two-launch sequence (GroupNorm from the producer's sums, then the conv by rule) against the fused form (coefficients +
the halo conv that normalises its input in LDS), per shape of the VAE, cold-rotated inputs.  usage: tools/gn_conv_fused_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fie_amd  # noqa: E402,F401
from fie_amd import hip  # noqa: E402

ctx = hip.context(0)
DEV = "cuda"
g = torch.Generator(device=DEV).manual_seed(0)


def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(0)
    torch.cuda.synchronize()
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for h, cin, cout, use_res in [(1024, 128, 128, False), (1024, 128, 128, True), (1024, 256, 128, False), (512, 256, 256, False), (512, 256, 256, True), (512, 512, 256, False),
                              (256, 512, 512, False), (256, 512, 512, True), (128, 512, 512, False), (128, 512, 512, True)]:
    copies = max(2, min(6, int(600e6 / (h * h * cin * 2)) + 1))
    xs = [torch.randn(1, h, h, cin, generator=g, device=DEV, dtype=torch.float16) for _ in range(copies)]
    wp = ctx.pack_conv3x3(torch.randn(cout, cin, 3, 3, generator=g, device=DEV, dtype=torch.float16) * (9 * cin) ** -0.5)
    bias = torch.randn(cout, generator=g, device=DEV, dtype=torch.float16)
    gam = (1 + 0.3 * torch.randn(cin, generator=g, device=DEV)).half()
    bet = (0.2 * torch.randn(cin, generator=g, device=DEV)).half()
    res = torch.randn(1, h, h, cout, generator=g, device=DEV, dtype=torch.float16) if use_res else None
    # give every copy a valid tag by arming + running a cheap producer once is not possible per timed call: time the pieces that differ instead --
    # unfused = groupnorm(apply from given stats) + conv ; fused = coef + conv_gn.  Sums: produced once per copy by a 1x1-like producer conv below.
    w1 = ctx.pack_conv3x3(torch.randn(cin, 64, 3, 3, generator=g, device=DEV, dtype=torch.float16) * 0.04)
    x0 = torch.randn(1, h, h, 64, generator=g, device=DEV, dtype=torch.float16)

    def unfused(i):
        x = ctx.conv3x3(x0, w1, cin, gn_groups=32)
        return ctx.conv3x3(ctx.groupnorm(x, gam, bet, 32, 1e-6, True), wp, cout, bias=bias, residual=res, gn_groups=32)

    def fused(i):
        x = ctx.conv3x3(x0, w1, cin, gn_groups=32)
        return ctx.conv3x3_gn(x, ctx.groupnorm_coef(x, gam, bet, 32, 1e-6), True, wp, cout, bias=bias, residual=res, gn_groups=32)

    def producer(i):
        return ctx.conv3x3(x0, w1, cin, gn_groups=32)

    tp = timed(producer, 6)
    tu, tf = timed(unfused, 6) - tp, timed(fused, 6) - tp
    print(f"1 {h}x{h} {cin}->{cout} [{'res,' if use_res else ''}gn]: GroupNorm + conv {tu:7.1f} us   fused {tf:7.1f} us   ({(tf / tu - 1) * 100:+.1f} %)", flush=True)
