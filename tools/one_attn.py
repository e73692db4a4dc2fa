#!/usr/bin/env python3
"""Launch one attention shape a few times (for rocprofv3 --pmc). usage: tools/one_attn.py B H Tq Tk D [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fie_amd  # noqa: E402,F401
from fie_amd import hip  # noqa: E402

ctx = hip.context(0)
b, hn, tq, tk, d = map(int, sys.argv[1:6])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 3
c = hn * d
q = torch.randn(b * tq, c, device="cuda", dtype=torch.float16)
k = torch.randn(b * tk, c, device="cuda", dtype=torch.float16)
v = torch.randn(b * tk, c, device="cuda", dtype=torch.float16)
for _ in range(iters):
    ctx.attention(q, k, v, hn, d, tq, tk, b)
torch.cuda.synchronize()
