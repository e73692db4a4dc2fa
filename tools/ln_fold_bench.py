"""Cost of folding a LayerNorm into the consuming GEMM (fie_gemm_ln_f16) against the plain GEMM and the LayerNorm kernel it replaces."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fie_amd
from fie_amd import hip
from fie_amd.nn import Linear
from tools.microbench import timeit
ctx = hip.context(0)
g = torch.Generator().manual_seed(0)
for m, c, n, geglu in [(2048, 1280, 1280, False), (2048, 1280, 3840, False), (2048, 1280, 10240, True), (8192, 640, 640, False), (8192, 640, 5120, True)]:
    x = torch.randn(m, c, generator=g).half().cuda()
    w = (torch.randn(n, c, generator=g) * c ** -0.5).half()
    b = torch.randn(n, generator=g).half()
    gamma, beta = torch.ones(c).half(), torch.zeros(c).half()
    res = torch.randn(m, n, generator=g).half().cuda() if not geglu else None
    plain = Linear(ctx, None, None, w=w, b=b, geglu=geglu)
    fold = Linear(ctx, None, None, w=w, b=b, geglu=geglu, ln=(gamma, beta, 1e-5))
    y = torch.empty_like(x)
    gd, bd = gamma.cuda(), beta.cuda()
    rows = []
    for rnd in range(3):
        t_plain = timeit(lambda: plain(ctx, x, residual=res))
        t_fold = timeit(lambda: fold(ctx, x, residual=res))
        t_ln = timeit(lambda: ctx.layernorm(x, gd, bd, out=y))
        rows.append((t_plain, t_fold, t_ln))
    r = [min(v[i] for v in rows) * 1e6 for i in range(3)]
    print(f"M={m} C={c} N={n} geglu={geglu}: plain GEMM {r[0]:.1f} us, folded-LN GEMM {r[1]:.1f}, LN kernel {r[2]:.1f}", flush=True)
