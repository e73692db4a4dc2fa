"""How much do cold (HBM) weights cost a GEMM, and does touching the NEXT weight matrix on a side stream while the current GEMM
runs (a software prefetch into the Infinity Cache) win it back?  Modes per shape, HIP-event timed over a rotation of weight copies
that exceeds the 256 MiB Infinity Cache:
  warm      one weight copy re-used (what tools/microbench.py measures)
  cold      rotate over the copies (every launch reads its weights from HBM, as inside the UNet)
  prefetch  cold + a side-stream kernel that reads copy i+1 while the GEMM on copy i runs"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fie_amd  # noqa: E402,F401
from fie_amd import hip  # noqa: E402

ctx = hip.context(0)
side = torch.cuda.Stream()
for m, n, k in [(2048, 1280, 1280), (2048, 1280, 5120), (2048, 3840, 1280), (2048, 10240, 1280), (8192, 640, 640)]:
    wbytes = n * k * 2
    copies = max(2, int(600e6 // wbytes) + 1)
    a = torch.randn(m, k, device="cuda", dtype=torch.float16)
    ws = [ctx.pack_linear(torch.randn(n, k, device="cuda", dtype=torch.float16) * k ** -0.5) for _ in range(copies)]
    res = torch.randn(m, n, device="cuda", dtype=torch.float16)
    out = torch.empty(m, n, device="cuda", dtype=torch.float16)
    iters = copies * 2

    def run(mode):
        main = torch.cuda.current_stream()
        for it in range(iters + copies):
            if it == copies:
                torch.cuda.synchronize()
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
            w = ws[0] if mode == "warm" else ws[it % copies]
            if mode == "prefetch":
                side.wait_stream(main)                       # starts when the previous GEMM is done = runs beside this GEMM
                with torch.cuda.stream(side):
                    ws[(it + 1) % copies].view(-1)[: wbytes // 2].float().sum()      # reads the next copy once
            ctx.gemm(a, w, n, out=out, residual=res)
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3

    t = {mode: min(run(mode) for _ in range(2)) for mode in ("warm", "cold", "prefetch")}
    print(f"M={m} N={n} K={k} ({copies} copies of {wbytes / 1e6:.1f} MB): " + "  ".join(f"{k_} {v:6.1f} us" for k_, v in t.items()), flush=True)
