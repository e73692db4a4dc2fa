#!/usr/bin/env python3
"""Race screen for the halo-resident conv (csrc/conv_halo.hip): the same launch repeated must give the same bits, alone and beside another
stream's kernels; every variant (bias / residual / GroupNorm sums / row bias), several shapes.  usage: tools/halo_race.py [code] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fie_amd  # noqa: E402,F401
from fie_amd import hip  # noqa: E402

ctx = hip.context(0)
DEV = "cuda"
code = int(sys.argv[1]) if len(sys.argv) > 1 else 72
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
g = torch.Generator(device=DEV).manual_seed(0)
side = torch.cuda.Stream()
bad = 0
for b, h, cin, cout, opts in [(1, 1024, 128, 128, "bias,res,gn"), (1, 1024, 128, 128, "bias"), (1, 512, 256, 256, "bias,gn"), (1, 512, 128, 256, ""), (1, 256, 512, 512, "bias,res,gn"),
                              (2, 128, 320, 320, "bias,rowbias,gn"), (2, 64, 640, 640, "bias,res,gn"), (1, 128, 512, 512, "bias,res"), (1, 1024, 256, 128, "bias,gn")]:
    x = torch.randn(b, h, h, cin, generator=g, device=DEV, dtype=torch.float16)
    wt = torch.randn(cout, cin, 3, 3, generator=g, device=DEV, dtype=torch.float16) * (9 * cin) ** -0.5
    bias = torch.randn(cout, generator=g, device=DEV, dtype=torch.float16) if "bias" in opts else None
    res = torch.randn(b, h, h, cout, generator=g, device=DEV, dtype=torch.float16) if "res" in opts else None
    rb = torch.randn(b, cout, generator=g, device=DEV, dtype=torch.float16) if "rowbias" in opts else None
    wp = ctx.pack_conv3x3(wt)
    a2 = torch.randn(4096, 1280, generator=g, device=DEV, dtype=torch.float16)
    w2 = ctx.pack_linear(torch.randn(1280, 1280, generator=g, device=DEV, dtype=torch.float16) * 0.03)
    ctx.force_tile(code)
    first = ctx.conv3x3(x, wp, cout, bias=bias, residual=res, rowbias=rb, gn_groups=32 if "gn" in opts else None).clone()
    name = hip.last_gemm_kernel(ctx)
    torch.cuda.synchronize()
    diffs = 0
    for it in range(reps):
        if it % 2:
            with torch.cuda.stream(side):                 # another stream's kernels share the chip (their own tile rule)
                ctx.force_tile(0)
                for _ in range(6):
                    ctx.gemm(a2, w2, 1280)
        ctx.force_tile(code)
        y = ctx.conv3x3(x, wp, cout, bias=bias, residual=res, rowbias=rb, gn_groups=32 if "gn" in opts else None)
        torch.cuda.synchronize()
        if not torch.equal(y, first):
            d = (y.float() - first.float()).abs()
            idx = d.view(b, h, h, cout).nonzero()
            diffs += 1
            if diffs <= 2:
                print(f"   rep {it}: {int((d > 0).sum())} elements differ, max {d.max().item():.3e}; first at {idx[0].tolist()} last at {idx[-1].tolist()}", flush=True)
    ctx.force_tile(0)
    print(f"{name}: B={b} {h}x{h} {cin}->{cout} [{opts}]: {diffs} of {reps} repeats differ", flush=True)
    bad += diffs
# the fused GroupNorm -> conv form (GNA: the halo is normalised in LDS by the lane that DMA'd it): the input's sums come from a producer conv each time
for h, cin, cout, use_res in [(1024, 128, 128, True), (512, 256, 256, False), (256, 512, 512, True), (1024, 256, 128, False)]:
    x0 = torch.randn(1, h, h, 64, generator=g, device=DEV, dtype=torch.float16)
    wp0 = ctx.pack_conv3x3(torch.randn(cin, 64, 3, 3, generator=g, device=DEV, dtype=torch.float16) * (9 * 64) ** -0.5)
    wp = ctx.pack_conv3x3(torch.randn(cout, cin, 3, 3, generator=g, device=DEV, dtype=torch.float16) * (9 * cin) ** -0.5)
    bias = torch.randn(cout, generator=g, device=DEV, dtype=torch.float16)
    gam = (1 + 0.3 * torch.randn(cin, generator=g, device=DEV)).half()
    bet = (0.2 * torch.randn(cin, generator=g, device=DEV)).half()
    res = torch.randn(1, h, h, cout, generator=g, device=DEV, dtype=torch.float16) if use_res else None
    a2 = torch.randn(4096, 1280, generator=g, device=DEV, dtype=torch.float16)
    w2 = ctx.pack_linear(torch.randn(1280, 1280, generator=g, device=DEV, dtype=torch.float16) * 0.03)

    def fused():
        x = ctx.conv3x3(x0, wp0, cin, gn_groups=32)
        coef = ctx.groupnorm_coef(x, gam, bet, 32, 1e-6)
        return ctx.conv3x3_gn(x, coef, True, wp, cout, bias=bias, residual=res, gn_groups=32)

    first = fused().clone()
    name = hip.last_gemm_kernel(ctx)
    torch.cuda.synchronize()
    diffs = 0
    for it in range(reps):
        if it % 2:
            with torch.cuda.stream(side):
                for _ in range(6):
                    ctx.gemm(a2, w2, 1280)
        y = fused()
        torch.cuda.synchronize()
        diffs += int(not torch.equal(y, first))
    print(f"{name} + GroupNorm of the input in LDS: 1 {h}x{h} {cin}->{cout} [bias,gn{',res' if use_res else ''}]: {diffs} of {reps} repeats differ", flush=True)
    bad += diffs
print("RACE SCREEN", "FAILED" if bad else "clean")
sys.exit(1 if bad else 0)
