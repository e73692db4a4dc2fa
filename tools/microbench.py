#!/usr/bin/env python3
"""Per-shape kernel micro-benchmark on the GPU box (HIP-event timed, random data): TFLOP/s of the GEMM / conv /
attention shapes that make up the SSD-1B + ControlNet + VAE hot path, GB/s of the HBM-bound norm / element-wise kernels.
usage: tools/microbench.py [gemm|conv|attn|norm|all] [tiles]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fie_amd  # noqa: E402,F401
from fie_amd import hip  # noqa: E402

ctx = hip.context(0)
DEV = "cuda"


def timeit(fn, iters=20, warm=3):
    try:
        for _ in range(warm):
            fn()
    except hip.FieError:            # forced tile code not eligible for this shape
        return float("inf")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


GEMMS = [(8192, 1920, 640), (4096, 5120, 640), (1024, 10240, 1280), (8192, 640, 640), (8192, 5120, 640), (8192, 640, 2560),
         (2048, 3840, 1280), (2048, 1280, 1280), (2048, 10240, 1280), (2048, 1280, 5120),
         (154, 2560, 2048), (32768, 320, 320), (16384, 1536, 512), (4096, 4096, 4096),
         # 77-token shapes (CLIP-L / bigG at CFG batch 2, cross-attention K/V)
         (154, 1280, 1280), (154, 3840, 1280), (154, 5120, 1280), (154, 1280, 5120), (154, 2304, 768), (154, 3072, 768), (154, 768, 3072),
         (154, 1280, 2048), (2, 1280, 1280)]
CONVS = [  # b, h, w, cin, cout, stride, ups
    (2, 128, 128, 320, 320, 1, 0), (2, 128, 128, 640, 320, 1, 0), (2, 64, 64, 640, 640, 1, 0), (2, 64, 64, 1280, 640, 1, 0),
    (2, 32, 32, 1280, 1280, 1, 0), (2, 32, 32, 2560, 1280, 1, 0), (2, 32, 32, 1280, 1280, 1, 1),
    (1, 1024, 1024, 128, 128, 1, 0), (1, 512, 512, 256, 256, 1, 0), (1, 256, 256, 512, 512, 1, 0), (1, 128, 128, 512, 512, 1, 0),
    (1, 512, 512, 256, 256, 1, 1), (1, 1024, 1024, 8, 128, 1, 0), (1, 1024, 1024, 128, 4, 1, 0)]
ATTNS = [(2, 10, 4096, 4096, 64), (2, 20, 1024, 1024, 64), (2, 10, 4096, 77, 64), (2, 20, 1024, 77, 64), (1, 1, 16384, 16384, 512),
         (2, 12, 77, 77, 64)]


def run(which, tiles):
    if which in ("gemm", "all"):
        for m, n, k in GEMMS:
            a = torch.randn(m, k, device=DEV, dtype=torch.float16)
            w = ctx.pack_linear(torch.randn(n, k, device=DEV, dtype=torch.float16) * k ** -0.5)
            out = torch.empty(m, n, device=DEV, dtype=torch.float16)
            res, ref = [], None
            for t in tiles:
                ctx.force_tile(t)
                dt = timeit(lambda: ctx.gemm(a, w, n, out=out))
                if ref is None:
                    ref = out.float().clone()
                err = float((out.float() - ref).abs().max())
                res.append(f"t{t}: {dt * 1e6:8.1f} us {2 * m * n * k / dt / 1e12:7.1f} TF" + (f" !err {err:.3g}" if err > 1e-2 else ""))
            print(f"gemm M={m:6d} N={n:6d} K={k:6d}  " + "  ".join(res), flush=True)
            if n >= 5120:       # FF1 shapes: also with the production epilogue (bias + GEGLU)
                wg = ctx.pack_linear(torch.randn(n, k, device=DEV, dtype=torch.float16) * k ** -0.5, geglu=True)
                bias = torch.randn(n, device=DEV, dtype=torch.float16)
                outg = torch.empty(m, n // 2, device=DEV, dtype=torch.float16)
                res = []
                for t in tiles:
                    ctx.force_tile(t)
                    dt = timeit(lambda: ctx.gemm(a, wg, n, out=outg, bias=bias, act=hip.ACT_GEGLU))
                    res.append(f"t{t}: {dt * 1e6:8.1f} us {2 * m * n * k / dt / 1e12:7.1f} TF")
                print(f"  +bias+GEGLU epilogue            " + "  ".join(res), flush=True)
    if which in ("conv", "all"):
        for b, h, w_, cin, cout, stride, ups in CONVS:
            x = torch.randn(b, h, w_, cin, device=DEV, dtype=torch.float16)
            wt = ctx.pack_conv3x3(torch.randn(cout, cin, 3, 3, device=DEV, dtype=torch.float16) * (9 * cin) ** -0.5)
            oh, ow = (h << ups) // stride, (w_ << ups) // stride
            res = []
            for t in tiles:
                ctx.force_tile(t)
                dt = timeit(lambda: ctx.conv3x3(x, wt, (cout + 3) // 4 * 4, stride=stride, upsample=bool(ups)), iters=10)
                res.append(f"t{t}: {dt * 1e6:8.1f} us {2 * b * oh * ow * 9 * cin * cout / dt / 1e12:7.1f} TF")
            print(f"conv B={b} {h}x{w_} {cin}->{cout} s{stride} u{ups}  " + "  ".join(res), flush=True)
    ctx.force_tile(0)
    if which in ("attn", "all"):
        for b, hn, tq, tk, d in ATTNS:
            c = hn * d
            q = torch.randn(b * tq, c, device=DEV, dtype=torch.float16)
            k = torch.randn(b * tk, c, device=DEV, dtype=torch.float16)
            v = torch.randn(b * tk, c, device=DEV, dtype=torch.float16)
            res = []
            for var in (1, 0):
                hip.lib().fie_debug_attn_variant(ctx.h, var)
                dt = timeit(lambda: ctx.attention(q, k, v, hn, d, tq, tk, b), iters=10)
                res.append(f"v{2 - var}: {dt * 1e6:8.1f} us {4 * b * hn * tq * tk * d / dt / 1e12:7.1f} TF")
            print(f"attn B={b} H={hn} Tq={tq} Tk={tk} D={d}: " + "  ".join(res), flush=True)


def run_norm():
    """HBM-bound kernels: algorithmic bytes (reads + writes of the tensor) / time against the ~8 TB/s HBM3E peak."""
    for b, rows, c, note in [(1, 1048576, 128, "VAE 1024^2"), (1, 262144, 256, "VAE 512^2"), (1, 65536, 512, "VAE 256^2"),
                             (2, 16384, 320, "UNet 128^2"), (2, 4096, 640, "UNet 64^2"), (2, 1024, 1280, "UNet 32^2 (single pass)")]:
        x = torch.randn(b, rows, c, device=DEV, dtype=torch.float16)
        g, bt = torch.randn(c, device=DEV, dtype=torch.float16), torch.randn(c, device=DEV, dtype=torch.float16)
        out = torch.empty_like(x)
        cg = c // 32
        v = 8 if cg % 8 == 0 else (4 if cg % 4 == 0 else 0)
        nv = -(-rows // (1024 // (cg // v))) if v else 0
        single = (v == 8 and nv <= 10) or (v == 4 and nv <= 20)          # csrc/norm.hip gn_onepass()
        for onepass in ((1, 0) if single else (0,)):
            hip.lib().fie_debug_gn_onepass(ctx.h, onepass)
            dt = timeit(lambda: ctx.groupnorm(x, g, bt, 32, 1e-5, True, out=out))
            passes = 2 if onepass else 3
            print(f"groupnorm+SiLU {note:24s} B={b} rows={rows:8d} C={c:5d} {'single-pass' if onepass else '3-kernel   '}: {dt * 1e6:7.1f} us "
                  f"{passes * x.numel() * 2 / dt / 1e9:7.0f} GB/s ({passes} passes over the tensor)", flush=True)
        hip.lib().fie_debug_gn_onepass(ctx.h, 1)
    for rows, c in [(2048, 1280), (8192, 640), (154, 1280)]:
        x = torch.randn(rows, c, device=DEV, dtype=torch.float16)
        g, bt = torch.randn(c, device=DEV, dtype=torch.float16), torch.randn(c, device=DEV, dtype=torch.float16)
        out = torch.empty_like(x)
        dt = timeit(lambda: ctx.layernorm(x, g, bt, out=out))
        print(f"layernorm rows={rows:6d} C={c:5d}: {dt * 1e6:7.1f} us {2 * x.numel() * 2 / dt / 1e9:7.0f} GB/s", flush=True)
    img = torch.randint(0, 256, (1024, 1024, 3), device=DEV, dtype=torch.uint8)
    dt = timeit(lambda: ctx.pixels_in(img, True))
    print(f"pixels_in 1024^2: {dt * 1e6:7.1f} us {(img.numel() + 1024 * 1024 * 8 * 2) / dt / 1e9:7.0f} GB/s", flush=True)
    small = torch.randint(0, 256, (512, 512, 3), device=DEV, dtype=torch.uint8)
    dt = timeit(lambda: ctx.resize_lanczos(small, 1024, 1024))
    print(f"LANCZOS 512^2 -> 1024^2 (two passes): {dt * 1e6:7.1f} us", flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    tiles = [int(t) for t in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2, 3]
    if which in ("norm", "all"):
        run_norm()
    if which != "norm":
        run(which, tiles)
