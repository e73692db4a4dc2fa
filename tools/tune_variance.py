"""How repeatable is the per-shape autotune?  Re-tunes three GEMM problems six times (slightly different M each time, so each is a new
problem with the same tile counts) with FIE_TUNE_VERBOSE=1, which makes the library print every candidate's cold-weight time.
usage: FIE_TUNE_VERBOSE=1 python tools/tune_variance.py 2>&1 | grep "fie tune" """
import os, sys
sys.path.insert(0, "/root/repo")
import fie_amd, torch
from fie_amd import hip
ctx = hip.context(0)
for rep in range(6):
    for (m, n, k) in [(2048, 10240, 1280), (8192, 5120, 640), (2048, 1280, 1280)]:
        a = torch.randn(m, k, device="cuda", dtype=torch.float16)
        w = ctx.pack_linear(torch.randn(n, k, device="cuda", dtype=torch.float16) * k ** -0.5, geglu=(n == 10240))
        bias = torch.randn(n, device="cuda", dtype=torch.float16)
        ctx.autotune(0); ctx.autotune(1)
        # a fresh shape key each time: vary M by a multiple of the tile that keeps the tile counts
        ctx.gemm(a[: m - 16 * rep] if rep else a, w, n, bias=bias, act=hip.ACT_GEGLU if n == 10240 else hip.ACT_NONE)
        torch.cuda.synchronize()
