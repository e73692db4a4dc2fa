#!/usr/bin/env python3
"""Where a K-step of the 256x128 ring kernel spends its cycles: the stamped instances (tile codes 97 = plain, 98 = fragment
prefetch) sum s_memtime deltas per wave over the K loop; this prints the per-K-step mean of every segment over all waves.
Stamps cost ~40-60 cycles each (an s_memtime round trip), so the stamped kernel is slower than the shipped one: read ratios.
usage: tools/kstep_stamps.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fie_amd  # noqa: E402,F401
from fie_amd import hip  # noqa: E402
from tools.microbench import timeit  # noqa: E402

ctx = hip.context(0)
DEV = "cuda"
NAMES = ["drain+barrier", "DMA issue A", "reads kk0", "MFMA kk0", "DMA issue W", "reads kk1", "MFMA kk1"]


def run(label, fn, tiles, nk, codes=((97, 62, 8), (98, 96, 8))):
    for code, plain, nw in codes:
        buf = torch.zeros(tiles * nw * 12, device=DEV, dtype=torch.int32)
        ctx.gemm_stamps(buf)
        ctx.force_tile(code)
        fn()
        torch.cuda.synchronize()
        s = buf.view(tiles, nw, 12).cpu().numpy().astype("uint32")
        t = timeit(fn, iters=10, warm=2)
        ctx.force_tile(plain)
        t0 = timeit(fn, iters=10, warm=2)
        ctx.force_tile(0)
        ctx.gemm_stamps(None)
        life = (s[..., 10] - s[..., 9]).astype("float64")                      # uint32 wrap-safe
        start = (s[..., 9] - s[..., 9].min()).astype("float64")
        span = ((s[..., 10] - s[..., 9].min()).astype("float64")).max()
        d = s.astype("float64")
        head = (f"{label} [{plain}: {t0 * 1e6:.1f} us, stamped {t * 1e6:.1f} us]  span first entry -> last exit {span:.0f} cyc, wave life mean {life.mean():.0f} max {life.max():.0f}, "
                f"last wave enters at {start.max():.0f}; prologue {d[..., 7].mean():.0f}, epilogue {d[..., 8].mean():.0f}")
        if code != 98:
            per = d[..., :7].mean(axis=(0, 1))
            per[0] /= nk - 1
            per[1:] /= nk
            cells = "  ".join(f"{n} {v:.0f}" for n, v in zip(NAMES, per.tolist()))
            print(f"{head}; K-step {per.sum():.0f} = {cells}", flush=True)
        else:
            print(f"{head}; K-step {d[..., 0].mean() / (nk - 1):.0f} (prefetch)", flush=True)


def main():
    for m, n, k in [(2048, 10240, 1280), (16384, 1280, 5120), (16384, 10240, 1280)]:
        a = torch.randn(m, k, device=DEV, dtype=torch.float16)
        w = ctx.pack_linear(torch.randn(n, k, device=DEV, dtype=torch.float16) * k ** -0.5)
        out = torch.empty(m, n, device=DEV, dtype=torch.float16)
        run(f"gemm M={m} N={n} K={k}", lambda: ctx.gemm(a, w, n, out=out), ((m + 255) // 256) * ((n + 127) // 128), k // 64)
    for m, n, k in [(2048, 1280, 1280), (2048, 3840, 1280), (2048, 1280, 5120), (8192, 640, 640)]:
        a = torch.randn(m, k, device=DEV, dtype=torch.float16)
        w = ctx.pack_linear(torch.randn(n, k, device=DEV, dtype=torch.float16) * k ** -0.5)
        out = torch.empty(m, n, device=DEV, dtype=torch.float16)
        run(f"gemm M={m} N={n} K={k}", lambda: ctx.gemm(a, w, n, out=out), ((m + 127) // 128) * ((n + 63) // 64), k // 64, codes=((94, 42, 4),))
    for b, h, cin, cout in [(2, 64, 640, 640), (1, 512, 256, 256)]:
        x = torch.randn(b, h, h, cin, device=DEV, dtype=torch.float16)
        w = ctx.pack_conv3x3(torch.randn(cout, cin, 3, 3, device=DEV, dtype=torch.float16) * (9 * cin) ** -0.5)
        run(f"conv B={b} {h}x{h} {cin}->{cout}", lambda: ctx.conv3x3(x, w, cout), ((b * h * h + 255) // 256) * ((cout + 127) // 128), 9 * cin // 64)


if __name__ == "__main__":
    main()
