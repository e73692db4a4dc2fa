#!/usr/bin/env python3
"""Where a LayerNorm-folded GEMM launch is wrong, and what kind of wrong (profiles/r04_ln_fold_tile48_anomaly.md): M 8192 x N 640 x K 640 on a forced tile,
every wrong 16-row x 1-column spot printed with the error against mean * rstd (a missing `- mean * colsum` term shows as err / (mean * rstd) = colsum) and
against rstd alone (a wrong accumulator or bias shows no such pattern).  usage: tools/ln_fold_anomaly.py [tile=48] [launches=3]"""
import math
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fie_amd  # noqa: E402,F401
from fie_amd import hip  # noqa: E402

tile = int(sys.argv[1]) if len(sys.argv) > 1 else 48
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = hip.context(0)


def rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g).half()


m, n, k = 8192, 640, 640
x = rnd(m, k, seed=1) * 2 + rnd(m, 1, seed=7) * 6
w, b = rnd(n, k, seed=2) / math.sqrt(k), rnd(n, seed=3) * 0.1
g, bta = 1 + 0.2 * rnd(k, seed=4), 0.1 * rnd(k, seed=5)
xf = x.float()
mu, rstd = xf.mean(1), (xf.var(1, unbiased=False) + 1e-5).rsqrt()
ref = F.layer_norm(xf, (k,), g.float(), bta.float(), 1e-5) @ w.float().t() + b.float()
wf = (w.float() * g.float()[None, :]).half().float()
acc = xf @ wf.t()                                   # what the MFMAs sum
wp, tab = ctx.fold_layernorm(w, b, g, bta)
tabc = tab.cpu()
xd = x.cuda()
bn = {48: 80, 42: 64, 96: 128}[tile]
for rep in range(launches):
    ctx.force_tile(tile)
    out = ctx.gemm_ln(xd, wp, n, tab).float().cpu()
    ctx.force_tile(0)
    err = out - ref
    bad = (err.abs() > 0.05).nonzero()
    print(f"launch {rep} [{hip.last_gemm_kernel(ctx)}]: {len(bad)} wrong elements, {len(set(r // 16 for r, _ in bad.tolist()))} row fragments")
    seen = set()
    for r, c in bad.tolist():
        key = (r // 16, c)
        if key in seen or len(seen) >= 5:
            continue
        seen.add(key)
        r0 = r // 16 * 16
        e, rs = err[r0:r0 + 16, c], rstd[r0:r0 + 16]
        print(f"  rows {r0}..{r0 + 15} (fragment {(r0 // 16) & 1} of wave {(r0 % 128) // 32}, block row {r0 // 128}) col {c} (tile col {c % bn}: fragment {c % bn // 16}, fq {c % 16 // 4}, q {c % 4}): "
              f"colsum {tabc[c, 0]:.3f}, bias' {tabc[c, 1]:.3f}")
        print("    err / (mean * rstd) ", [round(v, 3) for v in (e / (mu[r0:r0 + 16] * rs)).tolist()])
        print("    err / rstd          ", [round(v, 3) for v in (e / rs).tolist()])
        print("    (out - bias') / rstd vs acc - mean * colsum", [round(v, 2) for v in ((out[r0:r0 + 16, c] - tabc[c, 1]) / rs).tolist()][:6],
              [round(v, 2) for v in (acc[r0:r0 + 16, c] - mu[r0:r0 + 16] * tabc[c, 0]).tolist()][:6], "acc", [round(v, 2) for v in acc[r0:r0 + 16, c].tolist()][:6])
        print("    the wave's other row fragment, same column: max |err|", round(err[(r0 ^ 16):(r0 ^ 16) + 16, c].abs().max().item(), 5),
              " same rows, columns c-3..c+3:", [round(v, 4) for v in err[r0:r0 + 16, max(c - 3, 0):c + 4].abs().max(0).values.tolist()])
