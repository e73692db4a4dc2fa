#!/usr/bin/env python3
"""LayerNorm folded into its consumer GEMM (fie_gemm_ln_f16) against the two launches it replaces (fie_layernorm_f16 + fie_gemm_f16), per shape of the
transformer blocks, weights cold (rotated over > 256 MB of copies): the three LN consumers of a BasicTransformerBlock at the 32x32-latent level
(M 2048 x K 1280) and the 64x64-latent level (M 8192 x K 640).  usage: tools/ln_fold_time.py"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fie_amd  # noqa: E402,F401
from fie_amd import hip  # noqa: E402
from tools.cold_weights import time_rot  # noqa: E402

ctx = hip.context(0)
ctx.autotune(True)
tot_two = tot_fold = 0.0
for name, m, n, k, geglu, per_edit in [      # per_edit: launches per edit (SSD-1B + ControlNet, 2 evaluations); FF1 is not folded by default (FIE_LN_FOLD_FF1)
                                       ("qkv", 2048, 3840, 1280, False, 112), ("to_q (cross)", 2048, 1280, 1280, False, 112), ("FF1 GEGLU", 2048, 10240, 1280, True, 112),
                                       ("qkv", 8192, 1920, 640, False, 24), ("to_q (cross)", 8192, 640, 640, False, 24), ("FF1 GEGLU", 8192, 5120, 640, True, 24)]:
    copies = max(2, int(600e6 / (n * k * 2)) + 1)
    x = torch.randn(m, k, device="cuda", dtype=torch.float16) * 2 + 1
    g, b = torch.randn(k, device="cuda", dtype=torch.float16) * 0.2 + 1, torch.randn(k, device="cuda", dtype=torch.float16) * 0.1
    bias = torch.randn(n, device="cuda", dtype=torch.float16) * 0.1
    act = hip.ACT_GEGLU if geglu else hip.ACT_NONE
    wsrc = [torch.randn(n, k, device="cuda", dtype=torch.float16) * k ** -0.5 for _ in range(copies)]
    plain = [ctx.pack_linear(w, geglu=geglu) for w in wsrc]
    folded = [ctx.fold_layernorm(w, bias, g, b, geglu=geglu) for w in wsrc]
    del wsrc
    bias_p = torch.stack([bias[: n // 2], bias[n // 2:]], 1).reshape(-1).contiguous() if geglu else bias
    out = torch.empty(m, n // 2 if geglu else n, device="cuda", dtype=torch.float16)
    y = torch.empty_like(x)
    two = [lambda w=w: ctx.gemm(ctx.layernorm(x, g, b, out=y), w, n, out=out, bias=bias_p, act=act) for w in plain]
    gemm_only = [lambda w=w: ctx.gemm(y, w, n, out=out, bias=bias_p, act=act) for w in plain]
    fold = [lambda w=w, t=t: ctx.gemm_ln(x, w, n, t, act=act, out=out) for w, t in folded]
    two[0]()              # tunes the plain shape
    fold[0]()
    kern = hip.last_gemm_kernel(ctx)
    reps = max(40, copies)
    t_two = statistics.median(time_rot(two, reps) for _ in range(5))
    t_gemm = statistics.median(time_rot(gemm_only, reps) for _ in range(5))
    t_fold = statistics.median(time_rot(fold, reps) for _ in range(5))
    tot_two += t_two * per_edit
    tot_fold += t_fold * per_edit
    print(f"{name:13s} M={m} N={n} K={k}: LayerNorm + GEMM {t_two * 1e6:6.1f} us (GEMM alone {t_gemm * 1e6:6.1f}), folded {t_fold * 1e6:6.1f} us  [{kern}]  "
          f"x {per_edit} per edit = {(t_two - t_fold) * per_edit * 1e3:+.2f} ms", flush=True)
print(f"per edit (SSD-1B + ControlNet, 2 evaluations): {tot_two * 1e3:.2f} ms -> {tot_fold * 1e3:.2f} ms")
