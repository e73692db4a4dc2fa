#!/usr/bin/env python3
"""Does pulling the NEXT layers' weights into the Infinity Cache on a side stream pay?  A chain of GEMM / conv layers with distinct
weights (> 600 MB in total, so every layer's weights are cold when its turn comes, as inside the UNet) is run (a) plainly, (b) with
fie_prefetch of layer k + D's weights on a side stream released when layer k starts.  Both with the tiles the cold-timed autotune picks
and with the warm-timed winners (two-stage tiles).  usage: tools/prefetch_chain.py"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fie_amd  # noqa: E402,F401
from fie_amd import hip  # noqa: E402

ctx = hip.context(0)
DEV = "cuda"
# one transformer-ish block at 32x32 latents + one at 64x64 + convs, repeated with fresh weights
SHAPES = [("g", 2048, 3840, 1280), ("g", 2048, 1280, 1280), ("g", 2048, 1280, 1280), ("g", 2048, 10240, 1280), ("g", 2048, 1280, 5120),
          ("c", 2, 32, 1280, 1280), ("c", 2, 32, 1280, 1280),
          ("g", 8192, 1920, 640), ("g", 8192, 640, 640), ("g", 8192, 5120, 640), ("g", 8192, 640, 2560), ("c", 2, 64, 640, 640)]
WARM_BEST = {("g", 2048, 3840, 1280): 52, ("g", 2048, 1280, 1280): 46, ("g", 2048, 10240, 1280): 54, ("g", 2048, 1280, 5120): 42,
             ("c", 2, 32, 1280, 1280): 95, ("g", 8192, 1920, 640): 52, ("g", 8192, 640, 640): 44, ("g", 8192, 5120, 640): 54,
             ("g", 8192, 640, 2560): 52, ("c", 2, 64, 640, 640): 96}


def build(reps):
    layers, total = [], 0
    for _ in range(reps):
        for sh in SHAPES:
            if sh[0] == "g":
                _, m, n, k = sh
                a = torch.randn(m, k, device=DEV, dtype=torch.float16)
                w = ctx.pack_linear(torch.randn(n, k, device=DEV, dtype=torch.float16) * k ** -0.5)
                out = torch.empty(m, n, device=DEV, dtype=torch.float16)
                fn = (lambda a=a, w=w, n=n, out=out: ctx.gemm(a, w, n, out=out))
            else:
                _, b, h, cin, cout = sh
                x = torch.randn(b, h, h, cin, device=DEV, dtype=torch.float16)
                w = ctx.pack_conv3x3(torch.randn(cout, cin, 3, 3, device=DEV, dtype=torch.float16) * (9 * cin) ** -0.5)
                out = torch.empty(b, h, h, cout, device=DEV, dtype=torch.float16)
                fn = (lambda x=x, w=w, cout=cout, out=out: ctx.conv3x3(x, w, cout, out=out))
            layers.append((sh, fn, w))
            total += w.numel() * w.element_size()
    return layers, total


def run(layers, dist, codes, side, blocks):
    main = torch.cuda.current_stream()
    evs = [torch.cuda.Event() for _ in layers]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k, (sh, fn, _) in enumerate(layers):
        if dist and k + dist < len(layers):
            evs[k].record(main)                      # layer k is about to start: release the prefetch of layer k + dist
            side.wait_event(evs[k])
            ctx.prefetch(layers[k + dist][2], stream=side, blocks=blocks)
        ctx.force_tile(codes.get(sh, 0) if codes else 0)
        fn()
    ctx.force_tile(0)
    main.wait_stream(side)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


def main():
    layers, total = build(6)
    print(f"{len(layers)} layers, {total / 1e6:.0f} MB of weights")
    side = torch.cuda.Stream()
    ctx.autotune(1)
    for _, fn, _ in layers[:len(SHAPES)]:
        fn()                                          # cold-timed tile per shape
    ctx.autotune(2)
    print(ctx.autotune_report()[1])
    for tag, codes in (("cold-tuned tiles", None), ("warm-best tiles", WARM_BEST)):
        for dist, blocks in ((0, 0), (1, 16), (2, 16), (3, 16), (2, 64), (2, 4)):
            t = statistics.median(run(layers, dist, codes, side, blocks) for _ in range(5))
            print(f"{tag:18s} prefetch distance {dist} blocks {blocks:3d}: {t:8.1f} us per chain, {t / len(layers):6.1f} us per layer", flush=True)


if __name__ == "__main__":
    main()
