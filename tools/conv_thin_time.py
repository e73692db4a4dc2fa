#!/usr/bin/env python3
"""Thin 3x3 convs (Cout <= 16: csrc/conv_thin.hip, tile code 77) against the im2col tile the rule gave them before (fie_debug_tune_exclude("77")),
per shape of one edit, inputs rotated over > 256 MB of copies (cold, as inside the network).
usage: tools/conv_thin_time.py [substring of the shape name] [thin-only]     (thin-only: no launches of the other tile: for rocprofv3 --pmc passes)"""
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


# (B, H, W, Cin, Cout, ldc, stride, act, what)
shapes = [(1, 1024, 1024, 128, 4, 4, 1, 0, "VAE decoder conv_out"), (1, 1024, 1024, 16, 16, 16, 1, 1, "cond. embedding 16 -> 16"),
          (1, 1024, 1024, 8, 16, 16, 1, 1, "cond. embedding 3(8) -> 16"), (2, 128, 128, 320, 4, 4, 1, 0, "UNet conv_out"),
          (1, 128, 128, 512, 8, 8, 1, 0, "VAE encoder conv_out")]
total = [0.0, 0.0]
want = sys.argv[1] if len(sys.argv) > 1 else ""
thin_only = len(sys.argv) > 2 and sys.argv[2] == "thin-only"
for b, h, w, cin, cout, ldc, stride, act, what in shapes:
    if want not in what:
        continue
    nbytes = b * h * w * cin * 2
    copies = max(2, min(24, int(600e6 / nbytes) + 1))
    xs = [torch.randn(b, h, w, cin, generator=g, device=DEV, dtype=torch.float16) for _ in range(copies)]
    wp = ctx.pack_conv3x3(torch.randn(cout, cin, 3, 3, generator=g, device=DEV, dtype=torch.float16) * (9 * cin) ** -0.5)
    bias = torch.randn(cout, generator=g, device=DEV, dtype=torch.float16)
    out = torch.empty(b, h // stride, w // stride, ldc, device=DEV, dtype=torch.float16)

    def run(i):
        return ctx.conv3x3(xs[i % copies], wp, cout, out=out, stride=stride, bias=bias, act=act, ldc=ldc)

    res = []
    for excl in (("", "") if thin_only else ("", "77", "", "77")):
        ctx.tune_exclude(excl)
        run(0)
        name = hip.last_gemm_kernel(ctx)
        res.append((timed(run, 20), name))
    ctx.tune_exclude("")
    if thin_only:
        print(f"{what}: thin {min(r[0] for r in res):.1f} us")
        continue
    t_new, t_old = min(res[0][0], res[2][0]), min(res[1][0], res[3][0])
    total[0] += t_new
    total[1] += t_old
    print(f"{what:28s} {b}x{h}x{w}x{cin} -> {cout}: thin {t_new:7.1f} us ({nbytes / t_new / 1e6:5.2f} TB/s of input)   before {t_old:7.1f} us [{res[1][1]}]", flush=True)
print(f"sum: thin {total[0]:.1f} us, before {total[1]:.1f} us")
