#!/usr/bin/env python3
"""Cross-attention over the 77 text tokens: one 96-key tile (default) against two 64-key tiles (fie_debug_attn_variant 3), 50 launches per hipGraph replay."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fie_amd  # noqa: E402,F401
from fie_amd import hip  # noqa: E402
from bench import _graph_ms  # noqa: E402

ctx = hip.context(0)
for b, hn, tq in [(2, 20, 1024), (2, 10, 4096)]:
    c = hn * 64
    q = torch.randn(b * tq, c, device="cuda", dtype=torch.float16)
    kv = torch.randn(b * 77, 2 * c, device="cuda", dtype=torch.float16)
    outs = {}
    for v in (0, 3, 0, 3):
        hip.lib().fie_debug_attn_variant(ctx.h, v)

        def many():
            for _ in range(50):
                o = ctx.attention(q, kv[:, :c], kv[:, c:], hn, 64, tq, 77, b)
            return o
        ms, o, _ = _graph_ms(many)
        outs[v] = o.clone()
        print(f"cross-attention B={b} H={hn} Tq={tq}: variant {v}: {ms * 1e3 / 50:5.2f} us", flush=True)
    ref = torch.nn.functional.scaled_dot_product_attention(q.view(b, tq, hn, 64).transpose(1, 2).float(), kv[:, :c].reshape(b, 77, hn, 64).transpose(1, 2).float(),
                                                           kv[:, c:].reshape(b, 77, hn, 64).transpose(1, 2).float()).transpose(1, 2).reshape(b * tq, c)
    print("   max |one tile - two tiles|", (outs[0].float() - outs[3].float()).abs().max().item(), " vs torch", (outs[0].float() - ref).abs().max().item())
hip.lib().fie_debug_attn_variant(ctx.h, 0)
